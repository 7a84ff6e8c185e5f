D=msamtools_amd/bin/msamtools-dev
$D synth --groups 4000000 --refs 1000 -b > /tmp/a.bam
$D synth --groups 1500000 --refs 1000 --seq -b > /tmp/s.bam
for f in /tmp/a.bam /tmp/s.bam; do
for k in 0 2 4 6 8 10; do
  echo "mix=$k $f: $(MSX_INFLATE_MIX=$k python3 scripts/bench_inflate.py $f 8192 2>&1 | tail -1)"
done
echo "vec=1 $f: $(MSX_INFLATE_VEC=1 python3 scripts/bench_inflate.py $f 8192 2>&1 | tail -1)"
done
