#!/bin/bash
# round 6: the SAM-text leg by itself, with the decode stage's own split (MSX_TIMING=1)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_sam}; rm -rf $OUT; mkdir -p $OUT
G=${2:-4000000}
EXE=msamtools_amd/bin/msamtools; DEV=msamtools_amd/bin/msamtools-dev
T=/tmp/msx_sam; mkdir -p $T
$DEV synth --groups $G --refs 100000 --seq -h > $T/in.sam
ls -l $T/in.sam | tee $OUT/log.txt
cat $T/in.sam > /dev/null
F="filter -S -l 80 -p 95 -z 80 --besthit -bu"
for rep in 1 2 3; do
  a=$EPOCHREALTIME
  cat $T/in.sam | MSX_TIMING=1 $EXE $F --profile-out $T/p.gz --label S - 2> $OUT/one_$rep.err > $T/f.bam
  b=$EPOCHREALTIME
  python3 -c "print('one process: %.3f s' % ($b - $a))" | tee -a $OUT/log.txt
  grep "decode stage\|filter pipeline" $OUT/one_$rep.err | cut -c1-400 | tee -a $OUT/log.txt
done
for bb in 16777216 33554432; do
  a=$EPOCHREALTIME
  cat $T/in.sam | MSX_BATCH_BYTES=$bb MSX_TIMING=1 $EXE $F --profile-out $T/p.gz --label S - 2> $OUT/bb_$bb.err > $T/f.bam
  b=$EPOCHREALTIME
  python3 -c "print('MSX_BATCH_BYTES=$bb: %.3f s' % ($b - $a))" | tee -a $OUT/log.txt
  grep "filter pipeline\|# process" $OUT/bb_$bb.err | cut -c1-300 | tee -a $OUT/log.txt
done
a=$EPOCHREALTIME; MSX_TIMING=1 $EXE $F --profile-out $T/p.gz --label S $T/in.sam 2> $OUT/file.err > $T/f.bam; b=$EPOCHREALTIME
python3 -c "print('from the file, no pipe: %.3f s' % ($b - $a))" | tee -a $OUT/log.txt
grep "decode stage\|filter pipeline" $OUT/file.err | cut -c1-400 | tee -a $OUT/log.txt
a=$EPOCHREALTIME; cat $T/in.sam | cat > /dev/null; b=$EPOCHREALTIME
python3 -c "print('cat | cat: %.3f s' % ($b - $a))" | tee -a $OUT/log.txt
$DEV digest $T/f.bam | tee -a $OUT/log.txt
rm -rf $T
