#!/bin/bash
# One parameterised GPU-box script (gpurun -- 'bash scripts/gpu.sh NAME step [step ...]'); outputs under gpurun_out/NAME/.
# steps:
#   tests[=PYTEST_ARGS]        python -m pytest -m gpu -x -q PYTEST_ARGS (default: tests)
#   bench[=ARGS]               python bench.py ARGS  -> bench.json (+ one-line summary)
#   write_rate                 scripts/micro/write_rate.c on /tmp
#   e2e=GROUPS[,seq]           the command line on a synthetic BAM, A/B over the env settings in $E2E_ENVS (";"-separated)
#   stats=CMD                  rocprofv3 --kernel-trace --stats -- CMD
#   pmc=COUNTERS=CMD           rocprofv3 --pmc COUNTERS --kernel-trace -- CMD  (one pass)
#   sh=CMD                     anything else
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
NAME=$1; shift
OUT=gpurun_out/$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
EXE=msamtools_amd/bin/msamtools
DEV=msamtools_amd/bin/msamtools-dev
[ -x $DEV ] || DEV=$EXE
# a box whose GPU does not answer (or one a hung command has wedged) costs GPU-minutes and nothing else: stop at once
healthy() {
  timeout 180 python3 -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float(x.sum()) == 1 << 20" > $OUT/health.log 2>&1 && return 0
  echo "GPU health check failed:"; tail -3 $OUT/health.log; return 1
}
healthy || exit 1
k=0
for step in "$@"; do
  k=$((k+1))
  if [ $k -gt 1 ]; then healthy || exit 1; fi
  kind=${step%%=*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*=}
  echo "== step $k: $step"
  case $kind in
    tests)
      eval "timeout 1200 python -m pytest --timeout=300 -m gpu -x -q --durations=5 ${arg:-tests}" > $OUT/pytest_$k.log 2>&1; echo "rc=$?" >> $OUT/pytest_$k.log
      tail -12 $OUT/pytest_$k.log ;;
    bench)
      timeout 1200 python bench.py $arg > $OUT/bench_$k.json 2> $OUT/bench_$k.err; echo "rc=$?"
      python3 - $OUT/bench_$k.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("roofline", {}).get("frac"))
    for k, v in d.get("roofline", {}).get("per_kernel", {}).items():
        print("  ", k, v.get("ms_per_step"), v.get("launches"), v.get("algorithmic_GBps"))
    for key in ("e2e", "e2e_seq", "coverage"):
        if key in d:
            print(key, json.dumps(d[key])[:1500])
except Exception as e:
    print("bench failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
      ;;
    write_rate)
      { df -h /tmp /dev/shm; nproc; free -g; cat /sys/fs/cgroup/cpu.max 2>/dev/null; uname -r
        gcc -O2 -o /tmp/write_rate scripts/micro/write_rate.c -lpthread
        for T in 4 8 16; do /tmp/write_rate /tmp 4 $T; done
        /tmp/write_rate /dev/shm 4 16; } > $OUT/write_rate.log 2>&1
      cat $OUT/write_rate.log ;;
    e2e)
      groups=${arg%%,*}; seq=""; [ "$arg" != "$groups" ] && seq="--seq"
      T=/tmp/msx_e2e_$$; mkdir -p $T
      $DEV synth --groups $groups --refs 1000000 $seq -b > $T/in.bam
      ls -l $T/in.bam
      IFS=';' read -ra ENVS <<< "${E2E_ENVS:-X=1}"
      for e in "${ENVS[@]}"; do
        for rep in 1 2; do
          rm -f $T/f.bam; sleep 1
          t0=$(date +%s.%N)
          env MSX_TIMING=1 $e $EXE filter -l 80 -p 95 -z 80 --besthit ${E2E_MODE:--bu} --profile-out $T/p.gz --label S $T/in.bam > $T/f.bam 2> $OUT/e2e_${k}_err.log
          t1=$(date +%s.%N)
          echo "[$e] $(python3 -c "print(round($t1-$t0,3))") s | $(grep 'filter pipeline' $OUT/e2e_${k}_err.log | cut -c1-400)"
        done
        $DEV digest $T/f.bam; ls -l $T/f.bam | awk '{print $5}'
      done 2>&1 | tee $OUT/e2e_$k.log
      rm -rf $T ;;
    stats)
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$k -o s -- $arg > $OUT/stats_$k.log 2>&1; echo "rc=$?"
      f=$(find $OUT/stats_$k -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/stats_${k}_kernel_stats.csv && head -25 "$f" ;;
    pmc)
      ctr=${arg%%=*}; cmd=${arg#*=}
      timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$k -o p -- $cmd > $OUT/pmc_$k.log 2>&1; echo "rc=$?"
      f=$(find $OUT/pmc_$k -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 scripts/pmc_sum.py "$f" | tee $OUT/pmc_${k}_${ctr// /_}.txt ;;
    sh)
      bash -c "$arg" 2>&1 | tee $OUT/sh_$k.log | tail -40 ;;
  esac
done
