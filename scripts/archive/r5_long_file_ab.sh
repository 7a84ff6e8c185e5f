D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 80000000 --refs 1000000 -b > /tmp/big.bam
for mode in -b -bu; do
for e in "X=1" "MSX_INFLATE_AHEAD=1 MSX_COMP_BLOCKS=2048 MSX_COMP_BYTES=41943040"; do
for rep in 1 2; do
  rm -f /tmp/f.bam; sleep 1
  t0=$(date +%s.%N)
  env $e MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit $mode --profile-out /tmp/p.gz --label S /tmp/big.bam > /tmp/f.bam 2> /tmp/err.log
  t1=$(date +%s.%N)
  echo "[$e $mode] $(python3 -c "print(round($t1-$t0,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c20-120) $(grep '^# batches' /tmp/err.log | cut -c1-60)"
done
$D digest /tmp/f.bam
done
done
for cmd in "profile --label S -o /tmp/p1.gz" "coverage --summary -o /tmp/c1.gz"; do
for e in "X=1" "MSX_INFLATE_AHEAD=1 MSX_COMP_BLOCKS=2048 MSX_COMP_BYTES=41943040"; do
  t0=$(date +%s.%N); env $e $B $cmd /tmp/big.bam 2>/dev/null; t1=$(date +%s.%N)
  echo "[$e] $cmd: $(python3 -c "print(round($t1-$t0,3))") s $(zcat /tmp/p1.gz 2>/dev/null | grep -v ommand | md5sum | cut -c1-12) $(zcat /tmp/c1.gz 2>/dev/null | md5sum | cut -c1-12)"
done
done
