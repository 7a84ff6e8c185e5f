# the reference's two-process surface with a compressed pipe: filter -b | profile -   (against -bu)
D=msamtools_amd/bin/msamtools-dev; E=msamtools_amd/bin/msamtools
T=/tmp/pp; mkdir -p $T
$D synth --groups 20000000 --refs 1000000 -b > $T/in.bam
ls -l $T/in.bam
for mode in -bu -b; do
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    $E filter -l 80 -p 95 -z 80 --besthit $mode $T/in.bam 2>$T/f.err | $E profile --label S -o $T/p$mode.gz - 2>$T/p.err
    t1=$(date +%s.%N)
    echo "pipe $mode: $(python3 -c "print(round($t1-$t0,3))") s"
  done
done
zcat $T/p-bu.gz | md5sum; zcat $T/p-b.gz | md5sum
tail -3 $T/f.err; tail -3 $T/p.err
