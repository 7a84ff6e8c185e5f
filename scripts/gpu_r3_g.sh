#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_cli_scale.py -m gpu -q -x -k "sam_text" 2>&1 | tail -5
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
run() { echo "== $*"; ( time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam ) 2>&1 | grep "filter pipeline\|^real\|process:"; sleep 1; }
run X=1
run MSX_THREADS=24
run MSX_THREADS=32
run MSX_THREADS=12
run MSX_CLEAN_EXIT=1
