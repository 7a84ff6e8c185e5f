#!/bin/bash
# stress: the command line with 1-4 contexts on one GPU, -bu and -b, traced (MSX_TRACE), each run under timeout -- a hang shows in
# the last trace lines; REPS and DEVS choose how many and which (three deadlocks of round 4 were found with this)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dbgmulti
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 600000 --refs 3000 -b > /tmp/in.bam
export MSX_THREADS=16 MSX_BATCH_BYTES=2500000 MSX_BATCH_RECORDS=160000 MSX_INFLATE_BLOCKS=24 MSX_TIMING=1 MSX_TRACE=1
ref=""
for rep in $(seq 1 ${REPS:-6}); do
  for flag in -bu -b; do
    for dev in ${DEVS:-0 0,0 0,0,0 0,0,0,0}; do
      log=gpurun_out/dbgmulti/err_${rep}_${flag}_${dev}.log
      MSX_DEVICES=$dev timeout 30 $B filter -l 80 -p 95 -z 80 --besthit $flag --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2> $log
      rc=$?
      dg=$($D digest /tmp/f.bam 2>/dev/null | head -1)
      echo "rep $rep $flag devices $dev rc $rc $(grep -c trace $log) trace lines digest $dg"
      if [ $rc -ne 0 ]; then tail -40 $log; exit 0; fi
      rm -f $log
    done
  done
done
