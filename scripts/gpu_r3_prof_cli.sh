#!/bin/bash
# rocprofv3 kernel stats of the command line (device unpack path) on a 4 M-group file
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3prof
rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/msamtools_amd/bin/msamtools
$B synth --groups 6000000 --refs 1000000 -b > /tmp/in6.bam
export MSX_CLEAN_EXIT=1
rocprofv3 --kernel-trace --stats -d $OUT/cli -o cli -- $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in6.bam > /tmp/f.bam 2> $OUT/cli.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "head -32 {}"; ls -R $OUT | head -20
