#!/bin/bash
# debug: one-context command lines on uncompressed / compressed input, traced, under a short timeout
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dbgone
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 200000 --refs 800 -u > /tmp/in_u.bam
$D synth --groups 200000 --refs 800 -b > /tmp/in_b.bam
export MSX_THREADS=16 MSX_BATCH_BYTES=1500000 MSX_BATCH_RECORDS=110000 MSX_INFLATE_BLOCKS=24 MSX_TIMING=1 MSX_TRACE=1
for rep in 1 2 3; do
  for in in u b; do
    for flag in -bu -b; do
      log=gpurun_out/dbgone/err_${rep}_${in}_${flag}.log
      timeout 20 $B filter -l 80 -p 95 -z 80 --besthit $flag /tmp/in_$in.bam > /tmp/f.bam 2> $log
      rc=$?
      echo "rep $rep in $in $flag rc $rc $(grep -c trace $log) trace lines $($D digest /tmp/f.bam 2>/dev/null | head -1)"
      if [ $rc -ne 0 ]; then tail -30 $log | cut -c1-200; exit 0; fi
      rm -f $log
    done
  done
done
