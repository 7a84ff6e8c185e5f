# what makes a process that has lived 0.3 s and more slow to let go of?  (tiny input, 300 ms of sleep before _exit)
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 2000 --refs 1000 -b > /tmp/t.bam
run() {
  rm -f /tmp/f.bam; sleep 0.3
  t0=$(date +%s.%N)
  env MSX_EXIT_DELAY_MS=300 $1 MSX_TIMING=1 $B $2 /tmp/t.bam > /tmp/f.bam 2> /tmp/err.log
  t1=$(date +%s.%N)
  in=$(grep -E '^# process:' /tmp/err.log | awk '{print $3}')
  echo "[$1 | $2] after exit (minus the delay) $(python3 -c "print(round($t1-$t0-$in-0.3,3))")"
}
F="filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S"
for e in "X=1" "MSX_NO_PIN=1" "MSX_THREADS=1" "MSX_NO_WARMUP=1" "MSX_SERIAL=1" "MSX_SERIAL_IO=1" "MSX_NO_PIN=1 MSX_THREADS=1 MSX_SERIAL=1 MSX_NO_WARMUP=1"; do run "$e" "$F"; run "$e" "$F"; done
run "X=1" "filter -l 80 -p 95 -z 80 --besthit -bu"; run "X=1" "filter -l 80 -p 95 -z 80 -bu"
run "X=1" "profile --label S -o /tmp/p1.gz"; run "X=1" "summary"
