#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
$B synth --groups 20000000 --refs 1000000 -b > /tmp/in.bam
TIMEFORMAT="%R s real"
k=0
run() { for i in 1 2 3; do k=$((k+1)); { time env MSX_TIMING=1 "$@" $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p$k.gz --label S /tmp/in.bam > /tmp/f$k.bam 2> /tmp/err.txt; } 2> /tmp/time.txt; grep "batches:\|filter pipeline" /tmp/err.txt | sed 's/; decode/ decode/; s/99992794 records.*//' | tr '\n' ' '; cat /tmp/time.txt; rm -f /tmp/f$k.bam; sleep 2; done; }
for cfg in X=1 MSX_PIN_THREADS=2 MSX_PIN_THREADS=4 X=2 MSX_PIN_THREADS=4; do echo "== $cfg"; run $cfg; done
