#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the command line on a 20 M-group file: where a batch's time goes on the device
# beyond the inflate kernel (round 5). Output: gpurun_out/r5trace/{timeline.txt,cli_timing.txt,per_batch.txt}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5trace
rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/msamtools_amd/bin/msamtools
msamtools_amd/bin/msamtools-dev synth --groups ${1:-20000000} --refs 1000000 -b > /tmp/in.bam
ls -l /tmp/in.bam
for rep in 1 2; do
  rm -f /tmp/f.bam
  MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> $OUT/plain_timing_$rep.txt
done
grep -E "filter pipeline|device stage|batches:" $OUT/plain_timing_2.txt | cut -c1-700
rm -f /tmp/f.bam
export MSX_TIMING=1 MSX_CLEAN_EXIT=1
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/cli -o cli -- $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> $OUT/cli_timing.txt
grep -E "filter pipeline|device stage" $OUT/cli_timing.txt | cut -c1-700
python3 scripts/archive/r5_cli_trace.py $OUT
rm -rf $OUT/cli
