# a longer file than BASELINE's: 400 M records (80 M QNAME groups, 1 M references) through the one-process command, -bu and -b
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
t0=$(date +%s.%N); $D synth --groups 80000000 --refs 1000000 -b > /tmp/big.bam; t1=$(date +%s.%N)
ls -l /tmp/big.bam | awk '{print "input bytes", $5}'; python3 -c "print('synth', round($t1-$t0,1), 's')"
for mode in -bu -b; do
for rep in 1 2; do
  rm -f /tmp/f.bam; sleep 1
  t0=$(date +%s.%N)
  MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit $mode --profile-out /tmp/p$mode.gz --label S /tmp/big.bam > /tmp/f.bam 2> /tmp/err.log
  t1=$(date +%s.%N)
  echo "[$mode] $(python3 -c "print(round($t1-$t0,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-330)"
done
$D digest /tmp/f.bam; ls -l /tmp/f.bam | awk '{print "output bytes", $5}'
done
zcat /tmp/p-bu.gz | grep -v ommand | md5sum; zcat /tmp/p-b.gz | grep -v ommand | md5sum
zcat /tmp/p-bu.gz | head -12
