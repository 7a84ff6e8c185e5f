#!/bin/bash
# End-to-end host-pipeline timing of the C command line on the GPU box
# (BAM file -> filtered BAM -> profile.txt.gz).  Reported separately from
# bench.py's device-resident `value`.
set -e
B=msamtools_amd/bin/msamtools
make -C msamtools_amd/csrc/host >/dev/null 2>&1 || true
NGRP=${1:-2000000}
REFS=${2:-10000}
T=/tmp/msx_e2e
mkdir -p $T
echo "host: $(nproc) cores; groups=$NGRP refs=$REFS"
time $B synth --groups $NGRP --refs $REFS -u > $T/in_u.bam
time $B synth --groups $NGRP --refs $REFS -b > $T/in_b.bam
N=$($B recode $T/in_u.bam | wc -l)
ls -la $T/in_u.bam $T/in_b.bam
echo "records=$N"
for IN in in_u in_b; do
  for TH in 1 8 32; do
    S=$(date +%s.%N)
    MSX_THREADS=$TH $B filter -bu -l 80 -p 95 -z 80 --besthit $T/$IN.bam > $T/f.bam
    E=$(date +%s.%N)
    echo "filter $IN threads=$TH: $(echo "$E - $S" | bc) s  -> $(echo "$N / ($E - $S) / 1000000" | bc -l | cut -c1-6) M aln/s"
  done
done
S=$(date +%s.%N)
$B filter -bu -l 80 -p 95 -z 80 --besthit $T/in_b.bam | $B profile --label S -o $T/p.gz - 2> $T/p.err
E=$(date +%s.%N)
echo "filter|profile in_b: $(echo "$E - $S" | bc) s -> $(echo "$N / ($E - $S) / 1000000" | bc -l | cut -c1-6) M aln/s"
tail -3 $T/p.err
zcat $T/p.gz | head -12
rm -rf $T
