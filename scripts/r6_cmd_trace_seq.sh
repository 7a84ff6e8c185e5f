#!/bin/bash
# round 6: the one-process command over a 100 M-record BAM file (-b) under a kernel + memory-copy trace: where the device's
# 'upload' stage (copy, inflate, CRC, record walk) spends its time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_cmd_trace}; rm -rf $OUT; mkdir -p $OUT
D=msamtools_amd/bin/msamtools-dev; B=msamtools_amd/bin/msamtools
$D synth --groups 20000000 --refs 1000000 --seq -b > /tmp/big.bam
ls -l /tmp/big.bam | awk '{print "input bytes", $5}' | tee $OUT/run.log
for rep in 1 2; do
MSX_TIMING=1 $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/big.bam 2>> $OUT/run.log > /tmp/f.bam
done
MSX_CLEAN_EXIT=1 MSX_TIMING=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/tr -- $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/big.bam 2>> $OUT/run.log > /tmp/f.bam
grep "filter pipeline" $OUT/run.log | cut -c1-330
k=$(find $OUT/tr -name "*kernel_stats.csv" | head -1); m=$(find $OUT/tr -name "*memory_copy_stats.csv" | head -1)
cp $k $OUT/kernel_stats.csv; cp $m $OUT/memory_copy_stats.csv 2>/dev/null
head -14 $OUT/kernel_stats.csv | cut -c1-160; cat $OUT/memory_copy_stats.csv | cut -c1-160
t=$(find $OUT/tr -name "*kernel_trace.csv" | head -1); c=$(find $OUT/tr -name "*memory_copy_trace.csv" | head -1)
python3 - "$t" "$c" <<'PY' | tee $OUT/busy.txt
import csv, sys
def iv(path, a="Start_Timestamp", b="End_Timestamp"):
    return sorted((int(r[a]), int(r[b])) for r in csv.DictReader(open(path)))
def union(v):
    busy, (ca, cb) = 0, v[0]
    for a, b in v[1:]:
        if a > cb: busy += cb - ca; ca, cb = a, b
        else: cb = max(cb, b)
    return busy + cb - ca
k, c = iv(sys.argv[1]), iv(sys.argv[2])
span = max(b for _, b in k + c) - min(a for a, _ in k + c)
print(f"span {span/1e6:.1f} ms; kernels busy (union) {union(k)/1e6:.1f} ms; copies busy (union) {union(c)/1e6:.1f} ms; either {union(sorted(k+c))/1e6:.1f} ms")
PY
cp $t $OUT/kernel_trace.csv; cp $c $OUT/memory_copy_trace.csv; rm -rf $OUT/tr
