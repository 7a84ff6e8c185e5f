#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
timeout 1500 python -m pytest tests/test_cli_scale.py tests/test_host_cli.py tests/test_gpu_unpack.py tests/test_gpu_inflate.py tests/test_gpu_chains.py -x -q -m gpu 2>&1 | tail -4
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
run() { for i in 1 2 3; do rm -f /tmp/f.bam; { time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/in.bam > $OUTF 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "batches\|filter pipeline\|on the host\|process:" /tmp/err.txt | sed 's/; decode/ decode/; s/99992794 records.*//; s/, [0-9.]* s of CPU.*//; s/; BGZF blocks inflated on the device//' | tr '\n' ' '; cat /tmp/time.txt; done; }
OUTF=/tmp/f.bam
for cfg in X=1 MSX_PIN_THREADS=1 MSX_PIN_THREADS=4 MSX_HOST_INFLATE=1; do echo "== $cfg"; run $cfg; done
$B digest /tmp/f.bam | tail -1
