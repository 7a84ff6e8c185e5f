#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
timeout 900 python -m pytest tests/test_cli_scale.py tests/test_host_cli.py -x -q -m gpu -k "pipe or stdin or stream" 2>&1 | tail -2
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
for cfg in MSX_PIPE_SPIN=0 MSX_PIPE_SPIN=200 MSX_PIPE_SPIN=1000 MSX_PIPE_SPIN=0 MSX_PIPE_SPIN=200; do echo "== $cfg"; for i in 1 2 3; do { time ( env $cfg $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/in.bam 2>/dev/null | env $cfg $B profile --label S -o /tmp/p2.gz - 2>/dev/null ); } 2>&1 | tail -1; done; done
zcat /tmp/p2.gz | grep -v "^# Command\|^#.*ommand" | md5sum
( $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/in.bam 2>/dev/null; echo done >&2 ) | $B digest - | tail -1
