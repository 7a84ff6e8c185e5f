#!/bin/bash
# soak: the first command of tests/test_cli_scale.py (uncompressed 1 M-record input, -bu, small batches), again and again under a
# short timeout -- a hang or a wrong digest stops it with the stderr of that run
cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 200000 --refs 800 -u > /tmp/in_u.bam
$D synth --groups 200000 --refs 800 -b > /tmp/in_b.bam
export MSX_THREADS=16 MSX_BATCH_BYTES=1500000 MSX_BATCH_RECORDS=110000 MSX_INFLATE_BLOCKS=24 MSX_TIMING=1
ref=""; n=0
for rep in $(seq 1 ${REPS:-100}); do
  for in in u b; do
    for flag in -bu -b; do
      n=$((n+1))
      timeout 30 $B filter -l 80 -p 95 -z 80 --besthit $flag /tmp/in_$in.bam > /tmp/f.bam 2> /tmp/err.log
      rc=$?
      d=$($D digest /tmp/f.bam 2>/dev/null | head -1)
      [ -z "$ref" ] && ref="$d"
      if [ $rc -ne 0 ] || [ "$d" != "$ref" ]; then echo "run $n (rep $rep, in $in, $flag): rc $rc digest $d (want $ref)"; tail -20 /tmp/err.log; exit 1; fi
    done
  done
done
echo "runs=$n all rc 0, digest $ref"
