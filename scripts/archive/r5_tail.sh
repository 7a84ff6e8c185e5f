D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/x.bam
for rep in 1 2 3 4; do
rm -f /tmp/f.bam; sleep 0.5
t0=$(date +%s.%N)
MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/x.bam 2> /tmp/err.log > /tmp/f.bam
t1=$(date +%s.%N)
echo "outside $(python3 -c "print(round($t1-$t0,3))") | $(grep -o 'writer done.*' /tmp/err.log) | $(grep -E '^# process' /tmp/err.log | cut -c1-44)"
done
