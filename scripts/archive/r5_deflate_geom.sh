D=msamtools_amd/bin/msamtools-dev
$D synth --groups 800000 --refs 1000 -u > /tmp/dl.bam
$D synth --groups 300000 --refs 1000 --seq -u > /tmp/ds.bam
for f in /tmp/dl.bam /tmp/ds.bam; do
  for g in 0 2 3 4; do
    echo "geom $g $f: $(MSX_DEFLATE_STATS=1 MSX_DEFLATE_GEOM=$g python3 scripts/bench_deflate.py $f 2>&1 | grep -E 'level 6|waves per' | sort -u | tr '\n' ' ')"
  done
  for w in 9 10; do
    echo "geom 3 waves $w $f: $(MSX_DEFLATE_WAVES=$w MSX_DEFLATE_GEOM=3 python3 scripts/bench_deflate.py $f 2>&1 | grep -E 'level 6' | tr '\n' ' ')"
  done
done
for g in 3 4; do MSX_DEFLATE_GEOM=$g timeout 600 python3 -m pytest -q -x -m gpu tests/test_gpu_deflate.py -k "not twins" 2>&1 | tail -2; done
