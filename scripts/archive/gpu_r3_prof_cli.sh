#!/bin/bash
# rocprofv3 traces of the command line (device unpack path): kernels, memory copies, HIP runtime calls
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3prof
rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/msamtools_amd/bin/msamtools
$B synth --groups 6000000 --refs 1000000 -b > /tmp/in6.bam
export MSX_CLEAN_EXIT=1
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d $OUT/cli -o cli -- $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in6.bam > /tmp/f.bam 2> $OUT/cli.err
ls -R $OUT | head -30
for f in $(find $OUT -name "*stats*.csv"); do echo "== $f"; head -14 $f | cut -c1-160; done
