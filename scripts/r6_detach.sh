#!/bin/bash
# round 6, VERDICT item 9: what the driver holds after _exit, hidden behind a child (MSX_DETACH=1)?  Single commands and
# back-to-back pairs, alternating, wall time as the shell sees it.  gpurun -- 'bash scripts/r6_detach.sh NAME [groups]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=${1:-r6_detach}; G=${2:-20000000}
OUT=gpurun_out/$NAME; rm -rf $OUT; mkdir -p $OUT
EXE=msamtools_amd/bin/msamtools; DEV=msamtools_amd/bin/msamtools-dev
T=/tmp/msx_detach; mkdir -p $T
$DEV synth --groups $G --refs 1000000 -b > $T/in.bam
ls -l $T/in.bam | tee $OUT/log.txt
F="filter -l 80 -p 95 -z 80 --besthit"
t() { local a=$EPOCHREALTIME; "$@"; local b=$EPOCHREALTIME; python3 -c "print($b - $a)"; }
one_bu()  { $EXE $F -bu --profile-out $T/p.gz --label S $T/in.bam > $T/f.bam; }
one_b()   { $EXE $F -b --profile-out $T/p.gz --label S $T/in.bam > $T/fb.bam; }
prof()    { $EXE profile --label S -o $T/p1.gz $T/in.bam; }
pair()    { $EXE $F -b $T/in.bam > $T/fb.bam; $EXE profile --label S -o $T/p2.gz $T/fb.bam; }
pipe_b()  { $EXE $F -b $T/in.bam | $EXE profile --label S -o $T/p3.gz -; }
one_bu > /dev/null 2>&1   # warm the page cache
for cmd in one_bu one_b prof pair pipe_b; do
  for rep in 1 2 3 4; do
    for d in 0 1; do
      sleep 0.7
      s=$(MSX_DETACH=$d t $cmd 2>/dev/null)
      printf "%-8s MSX_DETACH=%d  %.3f s\n" $cmd $d $s | tee -a $OUT/log.txt
    done
  done
done
# the outputs are the same either way
MSX_DETACH=0 one_b; $DEV digest $T/fb.bam | tee -a $OUT/log.txt; zcat $T/p.gz | md5sum | tee -a $OUT/log.txt
MSX_DETACH=1 one_b; $DEV digest $T/fb.bam | tee -a $OUT/log.txt; zcat $T/p.gz | md5sum | tee -a $OUT/log.txt
# a failing command's code and message come through
MSX_DETACH=1 $EXE filter -p 95 /nonexistent.bam; echo "rc of a missing input: $?" | tee -a $OUT/log.txt
printf '@HD\tVN:1.6\tSO:queryname\n@SQ\tSN:r1\tLN:1000\nq1\t0\tr1\t10\t255\t10M\t*\t0\t0\t*\t*\n' | MSX_DETACH=1 $EXE filter -S -p 95 --besthit - ; echo "rc of a record without AS: $?" | tee -a $OUT/log.txt
rm -rf $T
