#!/bin/bash
# round 6: the one-process command's thread hand-overs (MSX_TRACE=1) over the first 250 ms of a 100 M-record file
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_cmd_hosttrace}; rm -rf $OUT; mkdir -p $OUT
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/big.bam
$B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/big.bam 2>/dev/null > /tmp/f.bam
MSX_TRACE=1 MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/big.bam 2> $OUT/trace.log > /tmp/f.bam
grep -c trace $OUT/trace.log
python3 - $OUT/trace.log <<'PY' > $OUT/trace_rel.log
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
head -c 60000000 $OUT/trace_rel.log > $OUT/t.log; mv $OUT/t.log $OUT/trace_rel.log; rm $OUT/trace.log
