#!/bin/bash
# soak: the one-process command (-bu, every third run -b) on a 30 M-record file under varying slot counts, batch sizes, thread counts,
# ahead depth;
# every run's BAM digest and profile text must equal the first run's
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=msamtools_amd/bin/msamtools; D=msamtools_amd/bin/msamtools-dev
$D synth --groups 6000000 --refs 200000 -b > /tmp/in.bam
ref_d=""; ref_p=""; n=0; bad=0
for slots in 2 3 4 6; do for blocks in 97 640 2048 3000; do for th in 3 16; do for ahead in 1 2; do
  out=-bu; [ $((n % 3)) -eq 2 ] && out=-b
  n=$((n+1))
  env MSX_SLOTS=$slots MSX_COMP_BLOCKS=$blocks MSX_THREADS=$th MSX_INFLATE_AHEAD=$ahead timeout 120 $B filter -l 80 -p 95 -z 80 --besthit $out --profile-out /tmp/p.gz --label S /tmp/in.bam > /tmp/f.bam 2>/tmp/err.txt || { echo "run failed: $slots $blocks $th $ahead"; tail -2 /tmp/err.txt; bad=$((bad+1)); continue; }
  d=$($D digest /tmp/f.bam | tail -1); p=$(zcat /tmp/p.gz | grep -v "ommand" | md5sum | cut -c1-32)
  if [ -z "$ref_d" ]; then ref_d="$d"; ref_p="$p"; fi
  if [ "$d" != "$ref_d" ] || [ "$p" != "$ref_p" ]; then echo "MISMATCH slots=$slots blocks=$blocks threads=$th ahead=$ahead: $d $p"; bad=$((bad+1)); fi
done; done; done; done
echo "runs=$n bad=$bad ref: $ref_d $ref_p"
# the same for `profile` alone and `coverage --summary`
ref_p=""; ref_c=""; n=0
for slots in 2 4 6; do for blocks in 97 2048; do for th in 3 16; do
  n=$((n+1))
  env MSX_SLOTS=$slots MSX_COMP_BLOCKS=$blocks MSX_THREADS=$th timeout 120 $B profile --label S -o /tmp/p1.gz /tmp/in.bam 2>/dev/null || { echo "profile failed"; bad=$((bad+1)); }
  env MSX_SLOTS=$slots MSX_COMP_BLOCKS=$blocks MSX_THREADS=$th timeout 120 $B coverage --summary -o /tmp/c1.gz /tmp/in.bam 2>/dev/null || { echo "coverage failed"; bad=$((bad+1)); }
  p=$(zcat /tmp/p1.gz | grep -v "ommand" | md5sum | cut -c1-32); c=$(zcat /tmp/c1.gz | md5sum | cut -c1-32)
  if [ -z "$ref_p" ]; then ref_p="$p"; ref_c="$c"; fi
  if [ "$p" != "$ref_p" ] || [ "$c" != "$ref_c" ]; then echo "MISMATCH profile/coverage slots=$slots blocks=$blocks threads=$th: $p $c"; bad=$((bad+1)); fi
done; done; done
echo "profile/coverage runs=$n bad=$bad ref: $ref_p $ref_c"
