#!/bin/bash
# round 6: bgzip'd SAM text through filter -S, members inflated side by side on the reader's pool (msamtools) against one zlib
# stream (msamtools-prev, the commit before), and plain text for scale; 20 M SEQ/QUAL records
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev
$D synth --groups 4000000 --refs 100000 --seq -h > /tmp/s.sam
python3 - <<'PY'
import sys
sys.path.insert(0, 'tests')
from test_cli_scale import bgzf_blocks
open('/tmp/s.bgz.sam.gz', 'wb').write(bgzf_blocks(open('/tmp/s.sam', 'rb').read()))
PY
ls -l /tmp/s.sam /tmp/s.bgz.sam.gz
run() { # name exe file
  sleep 0.5
  local a=$EPOCHREALTIME
  MSX_TIMING=1 $2 filter -S -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S $3 > /tmp/f.bam 2> /tmp/err.log
  local b=$EPOCHREALTIME
  echo "[$1] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-120) $($D digest /tmp/f.bam)"
}
for rep in 1 2 3; do
  run plain msamtools_amd/bin/msamtools /tmp/s.sam
  run bgz-pool msamtools_amd/bin/msamtools /tmp/s.bgz.sam.gz
  [ -x msamtools_amd/bin/msamtools-prev ] && run bgz-one-stream msamtools_amd/bin/msamtools-prev /tmp/s.bgz.sam.gz
done
