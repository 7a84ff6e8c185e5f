#!/bin/bash
# the command line's host code under ThreadSanitizer on the GPU box (the library and the HIP runtime are not instrumented)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5tsan; rm -rf $OUT; mkdir -p $OUT
B=msamtools_amd/bin/msamtools-tsan; D=msamtools_amd/bin/msamtools-dev
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"
$D synth --groups 400000 --refs 2000 -b > /tmp/t.bam
run() { local n=$1; shift; MSX_THREADS=8 MSX_BATCH_BYTES=2500000 MSX_BATCH_RECORDS=160000 timeout 300 setarch x86_64 -R "$@" > /tmp/t.out 2> $OUT/$n.err; echo "$n rc=$? warnings=$(grep -c 'WARNING: ThreadSanitizer' $OUT/$n.err) $(grep -m1 -E 'FATAL|unexpected memory' $OUT/$n.err)"; }
run tiny $B filter -l 80 -p 95 -z 80 --besthit -S tests/golden/fixtures/besthit.sam
run filter_bu $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/t.bam
$D digest /tmp/t.out
run filter_b $B filter -l 80 -p 95 -z 80 --besthit -b /tmp/t.bam
$D digest /tmp/t.out
run profile $B profile --label S -o /tmp/p2.gz /tmp/t.bam
run coverage $B coverage --summary -o /tmp/c.gz /tmp/t.bam
MSX_DEVICES=0,0 run two_ctx $B filter -l 80 -p 95 -z 80 --besthit -bu --profile-out /tmp/p.gz --label S /tmp/t.bam
$D synth --groups 100000 --refs 2000 -u --seq > /tmp/ts.bam
run seq_b $B filter -l 80 -p 95 -z 80 --besthit -b --profile-out /tmp/p.gz --label S /tmp/ts.bam
$D recode -h /tmp/ts.bam > /tmp/ts.sam
MSX_SAM_CHUNK=1000000 run sam_in $B filter -l 80 -p 95 -z 80 --besthit -S -bu --profile-out /tmp/p.gz --label S /tmp/ts.sam
MSX_DEVICES=0,0,0 run three_ctx $B profile --label S -o /tmp/p3.gz /tmp/t.bam
MSX_HOST_INFLATE=1 run host_inflate $B filter -l 80 -p 95 -z 80 --besthit -bu /tmp/t.bam
MSX_HOST_UNPACK=1 run host_unpack $B filter -l 80 -p 95 -z 80 --besthit -b /tmp/t.bam
MSX_HOST_DEFLATE=1 run host_deflate $B filter -l 80 -p 95 -z 80 --besthit -b /tmp/t.bam
run rescore $B filter -l 80 -p 95 -z 80 --rescore --besthit -bu /tmp/t.bam
run summary $B summary /tmp/t.bam
run cov_text $B coverage -o /tmp/c2.gz /tmp/t.bam
python3 scripts/tsan_ours.py $OUT/*.err | tee $OUT/summary.txt
