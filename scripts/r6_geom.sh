#!/bin/bash
# round 6, VERDICT 1(a): per-command batch geometry -- filter -b / -bu on the 100 M-record SEQ/QUAL and lean files with the
# default batches (2048 blocks, one sent ahead) against larger ones (3072-4096 blocks, two ahead), alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_geom}; rm -rf $OUT; mkdir -p $OUT
EXE=msamtools_amd/bin/msamtools; DEV=msamtools_amd/bin/msamtools-dev
T=/tmp/msx_geom; mkdir -p $T
$DEV synth --groups 20000000 --refs 1000000 --seq -b > $T/seq.bam
$DEV synth --groups 20000000 --refs 1000000 -b > $T/lean.bam
ls -l $T/*.bam | tee $OUT/log.txt
F="filter -l 80 -p 95 -z 80 --besthit"
run() { # name envs... -- cmd
  local a=$EPOCHREALTIME; env "${@:2}" bash -c "$CMD" 2> $T/err.txt; local b=$EPOCHREALTIME
  python3 -c "print('%-34s %-8s %.3f s' % ('$NAME', '$1', $b - $a))" | tee -a $OUT/log.txt
  grep "filter pipeline\|profile pipeline" $T/err.txt | cut -c1-230 >> $OUT/detail.txt
}
for f in seq lean; do
  for flag in -b -bu; do
    NAME="$f $flag one process"; CMD="$EXE $F $flag --profile-out $T/p.gz --label S $T/$f.bam > $T/f.bam"
    for rep in 1 2 3; do
      sleep 0.7; run default MSX_TIMING=1
      sleep 0.7; run big3072 MSX_TIMING=1 MSX_COMP_BLOCKS=3072 MSX_COMP_BYTES=67108864 MSX_INFLATE_AHEAD=2
      sleep 0.7; run big4096w14 MSX_TIMING=1 MSX_COMP_BLOCKS=4096 MSX_COMP_BYTES=100663296 MSX_INFLATE_AHEAD=2 MSX_INFLATE_WAVES=14
      sleep 0.7; run w14 MSX_TIMING=1 MSX_INFLATE_WAVES=14
    done
  done
  NAME="$f profile"; CMD="$EXE profile --label S -o $T/p1.gz $T/$f.bam"
  for rep in 1 2; do
    sleep 0.7; run default MSX_TIMING=1
    sleep 0.7; run big3072 MSX_TIMING=1 MSX_COMP_BLOCKS=3072 MSX_COMP_BYTES=67108864 MSX_INFLATE_AHEAD=2
  done
done
rm -rf $T
