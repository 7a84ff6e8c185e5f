cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 -b > /tmp/lean100.bam
for rep in 1 2 3 4 5; do
  rm -f /tmp/f.bam; sleep 0.7
  a=$EPOCHREALTIME
  MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/lean100.bam > /tmp/f.bam 2> /tmp/err.log
  b=$EPOCHREALTIME
  echo "[now] $(python3 -c "print(round($b-$a,3))") s | $(grep 'filter pipeline' /tmp/err.log | cut -c1-250)"
done
