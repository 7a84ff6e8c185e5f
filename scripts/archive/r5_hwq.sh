# HSA queues cost 2 x ~182 MB of resident host memory each (wave save area): created at start-up, torn down at exit
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/x.bam
for q in 0 1 2 3 4; do
for mode in -bu -b; do
for rep in 1 2 3; do
  rm -f /tmp/f.bam /tmp/p.gz; sleep 0.3
  t0=$(date +%s.%N)
  if [ $q = 0 ]; then MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit $mode --profile-out /tmp/p.gz --label S /tmp/x.bam > /tmp/f.bam 2> /tmp/err.log
  else GPU_MAX_HW_QUEUES=$q MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit $mode --profile-out /tmp/p.gz --label S /tmp/x.bam > /tmp/f.bam 2> /tmp/err.log; fi
  t1=$(date +%s.%N)
  echo "[queues $q $mode] outside $(python3 -c "print(round($t1-$t0,3))") s | $(grep -E '^# process:' /tmp/err.log | cut -c11-40) | $(grep -E 'filter pipeline' /tmp/err.log | cut -c20-32) | $(grep 'process memory' /tmp/err.log | cut -c26-90)"
done
done
done
