#!/bin/bash
# where a 50 M-record command-line run spends the time the stage timers do not see
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools
T=/tmp/msx_st; mkdir -p $T
$B synth --groups 10000000 --refs 100000 -u > $T/u.bam
F="filter -l 80 -p 95 -z 80 --besthit -bu"
cat /sys/kernel/mm/transparent_hugepage/enabled
run() { echo "== $*"; for i in 1 2 3; do /usr/bin/env bash -c "time (MSX_TIMING=1 $* $B $F $T/u.bam > $T/f.bam)" 2>&1 | grep -E "process:|real" | tr '\n' ' '; echo; done; }
run
run MSX_NO_MMAP=1
run MSX_NO_PIN=1
echo "== to /dev/null"; for i in 1 2; do /usr/bin/env bash -c "time (MSX_TIMING=1 $B $F $T/u.bam > /dev/null)" 2>&1 | grep -E "process:|real" | tr '\n' ' '; echo; done
rm -rf $T
