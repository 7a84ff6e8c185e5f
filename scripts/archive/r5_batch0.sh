D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/x.bam
for rep in 1 2 3; do
for e in "X=1" "MSX_NO_MODULE_WARMUP=1"; do
rm -f /tmp/f.bam; sleep 0.5
t0=$(date +%s.%N)
env $e MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/x.bam 2> /tmp/err.log > /tmp/f.bam
t1=$(date +%s.%N)
echo "[$e] $(python3 -c "print(round($t1-$t0,3))") s | $(grep 'batch 0' /tmp/err.log | tr '\n' ' ' | cut -c1-110) | $(grep -o 'batches done at (ms): [0-9]* [0-9]* [0-9]*' /tmp/err.log) | $(grep -o 'device: start-up [0-9.]*' /tmp/err.log) | $(grep -o 'wall [0-9.]* s' /tmp/err.log)"
done
done
$D synth --groups 2000 --refs 1000 -b > /tmp/t.bam
for rep in 1 2 3; do for e in "X=1" "MSX_NO_MODULE_WARMUP=1"; do
t0=$(date +%s.%N); env $e $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/t.bam > /tmp/f.bam 2>/dev/null; t1=$(date +%s.%N)
echo "[tiny $e] $(python3 -c "print(round($t1-$t0,3))") s"
done; done
