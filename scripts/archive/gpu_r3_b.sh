#!/bin/bash
# round 3: the default bench line (with e2e parity, one-rank distributed step, coverage block)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_time.txt
tail -c 6000 $OUT/bench_default.json
tail -5 $OUT/bench_default.err
cat $OUT/bench_time.txt
