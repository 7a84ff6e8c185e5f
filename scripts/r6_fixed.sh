#!/bin/bash
# round 6: what a command spends outside its pipeline's steady state (MSX_TRACE: who waits for what over the first batches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r6_fixed; rm -rf $OUT; mkdir -p $OUT
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
for rep in 1 2 3; do
  rm -f /tmp/f.bam; sleep 0.5
  a=$EPOCHREALTIME
  MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/lean100.bam > /tmp/f.bam 2> /tmp/err.log
  b=$EPOCHREALTIME
  echo "== total $(python3 -c "print(round($b-$a,3))") s"
  grep "batches done\|filter pipeline" /tmp/err.log | cut -c1-330
  python3 - /tmp/err.log > $OUT/trace$rep.log <<'PY'
import sys,re
t0=None
for l in open(sys.argv[1]):
    m=re.match(r"# trace (\d+\.\d+): (.*)",l)
    if not m:
        print(l.rstrip()[:300]); continue
    t=float(m.group(1))
    if t0 is None: t0=t
    print(f"{(t-t0)*1e3:8.1f} {m.group(2)[:200]}")
PY
done
