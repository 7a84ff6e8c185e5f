#!/bin/bash
# The round's profile evidence in one GPU call: rocprofv3 kernel statistics and PMC traffic (separate passes, as
# MI355X_MICROARCH.md prescribes) of the c3 step, the coverage pass, the DEFLATE encoder and the inflater.
# usage (on the GPU box): bash scripts/gpu_profile.sh NAME     -> gpurun_out/NAME/; copy what is to be judged to profiles/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-prof}
rm -rf $OUT; mkdir -p $OUT
stats() {   # name, program args...
  local n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -- "$@" > $OUT/$n.log 2>&1
  find $OUT/$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${n}_kernel_stats.csv
  head -12 $OUT/${n}_kernel_stats.csv | cut -c1-160
}
pmc() {     # name, counters, program args...
  local n=$1 c=$2; shift; shift
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- "$@" > $OUT/$n.log 2>&1
  find $OUT/$n -name "*counter_collection.csv" | head -1 | xargs -I{} python3 scripts/pmc_sum.py {} > $OUT/${n}.json
}
STEP="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg"
STEP1="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-coverage --no-dist-leg"
stats c3 python3 $STEP
pmc c3_fetch FETCH_SIZE python3 $STEP1
pmc c3_write WRITE_SIZE python3 $STEP1
stats cov python3 scripts/bench_coverage.py 10000000 4 depths
pmc cov_fetch FETCH_SIZE python3 scripts/bench_coverage.py 10000000 2 depths
pmc cov_write WRITE_SIZE python3 scripts/bench_coverage.py 10000000 2 depths
pmc cov_c4_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" python3 scripts/bench_coverage.py 10000000 2 depths
# the encoder's input is written BEFORE the profiler starts: under rocprofv3 (with --pmc certainly) every child process has
# the GPU initialised by the preloaded tool library, and a child that execs is what this pool forbids
msamtools_amd/bin/msamtools-dev synth --groups 3200000 --refs 1000 -u > /tmp/prof_d.bam      # (15 000 blocks: several rounds of waves)
stats deflate python3 scripts/bench_deflate.py /tmp/prof_d.bam
pmc deflate_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" python3 scripts/bench_deflate.py /tmp/prof_d.bam
python3 - $OUT <<'PY'
import json, sys, hashlib, os, re
out = sys.argv[1]
def load(n):
    try:
        return json.load(open(f"{out}/{n}.json"))
    except Exception:
        return {}
def traffic(fetch, write):
    res = {}
    for k, cs in fetch.items():
        if "FETCH_SIZE" in cs and k in write and "WRITE_SIZE" in write[k]:
            f, w = cs["FETCH_SIZE"]["avg"], write[k]["WRITE_SIZE"]["avg"]
            res[re.sub(r"<.*>$", "", k)] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "launches_sampled": cs["FETCH_SIZE"]["launches"],
                                            "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    return res
src = {}
for f in sorted(os.listdir("msamtools_amd/csrc")):
    if f.endswith((".hip", ".h")):
        src[f] = hashlib.sha256(open(os.path.join("msamtools_amd/csrc", f), "rb").read()).hexdigest()[:16]
note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (scripts/gpu_profile.sh); per-launch averages. "
        "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE halving; calibrated for 16-byte "
        "streaming accesses, 8-byte gathers uncalibrated; Infinity-Cache hits are counted: fabric traffic out of the L2s, an upper bound of HBM traffic). "
        "source_sha16: the kernel sources these were collected on -- bench.py says so when they have changed since.")
json.dump({"_comment": note + " Workload: bench.py c3.", "source_sha16": src, "kernels": traffic(load("c3_fetch"), load("c3_write"))},
          open(f"{out}/pmc_traffic_c3.json", "w"), indent=1)
json.dump({"_comment": note + " Workload: scripts/bench_coverage.py (c4), msx_coverage_depths.", "source_sha16": src,
           "kernels": traffic(load("cov_fetch"), load("cov_write"))}, open(f"{out}/cov_c4_pmc.json", "w"), indent=1)
for k, v in json.load(open(f"{out}/pmc_traffic_c3.json"))["kernels"].items():
    print(k, v["hbm_bytes_per_launch"])
PY
tail -2 $OUT/cov.log $OUT/deflate.log
# the inflater: the serial kernel and the lane-parallel one (bench_inflate.py runs both on the same blocks): kernel statistics and
# SQ instruction counters per launch
msamtools_amd/bin/msamtools-dev synth --groups 1800000 --refs 1000 -b > /tmp/prof_a.bam
stats inflate python3 scripts/bench_inflate.py /tmp/prof_a.bam 8192
pmc inflate_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" python3 scripts/bench_inflate.py /tmp/prof_a.bam 8192
cat $OUT/inflate.log | tail -3
