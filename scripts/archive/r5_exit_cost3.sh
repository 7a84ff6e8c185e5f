hipcc --offload-arch=gfx950 -O1 -o /tmp/hip_life scripts/micro/hip_life.hip 2>/dev/null || exit 1
for what in 1 2 3 4 5 6 7; do
for ms in 0 400; do
for rep in 1 2; do
  out=$(/tmp/hip_life $ms $what); t1=$(date +%s.%N)
  echo "what=$what live +$ms ms: $(echo $out | cut -d' ' -f2-) | after the last line: $(python3 -c "print(round($t1 - $(echo $out | cut -d' ' -f1), 3))") s"
done
done
done
