D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/x.bam
for q in "X=1" "MSX_SKIP_DESTROY=1" "MSX_EXIT_SLEEP_MS=10" "MSX_EXIT_SLEEP_MS=40" "MSX_EXIT_SLEEP_MS=100" "MSX_SKIP_DESTROY=1 MSX_EXIT_SLEEP_MS=40"; do
for rep in 1 2 3; do
rm -f /tmp/f.bam; sleep 0.2
t0=$(date +%s.%N)
env $q MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/x.bam > /tmp/f.bam 2> /tmp/err.log
t1=$(date +%s.%N)
in=$(grep -E '^# process:' /tmp/err.log | awk '{print $3}')
echo "[$q] outside $(python3 -c "print(round($t1-$t0,3), 'inside', $in, 'teardown+start', round($t1-$t0-$in,3))")"
done
done
