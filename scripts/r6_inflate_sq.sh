#!/bin/bash
# round 6: SQ instruction counters of the three inflate kernels on the same 8192 lean blocks (one pass per counter group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r6_inflate_sq}; rm -rf $OUT; mkdir -p $OUT
msamtools_amd/bin/msamtools-dev synth --groups 1800000 --refs 1000 ${2:-} -b > /tmp/sq.bam
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$n -- python3 scripts/bench_inflate.py /tmp/sq.bam 8192 > $OUT/$n.log 2>&1
  find $OUT/$n -name "*counter_collection.csv" | head -1 | xargs -I{} python3 scripts/pmc_sum.py {} > $OUT/$n.json
done
python3 - $OUT <<'PY'
import json, sys, glob
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/SQ_*.json"):
    for k, cs in json.load(open(f)).items():
        if "bgzf_inflate" in k:
            res.setdefault(k.split("(")[0], {}).update({c: int(v["avg"]) for c, v in cs.items()})
json.dump(res, open(out + "/inflate_sq_all.json", "w"), indent=1)
for k, v in res.items():
    print(k, {c: round(x / 8192) for c, x in v.items()}, "(per block)")
PY
