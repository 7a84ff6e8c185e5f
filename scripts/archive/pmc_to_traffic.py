#!/usr/bin/env python3
"""gpurun_out/pmc_l2/summary.json (scripts/pmc_l2.sh) -> profiles/round1/pmc_traffic_c3.json:
HBM bytes per launch of every kernel, (2*FETCH_SIZE + WRITE_SIZE) * 1024 as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE reports half the bytes)."""
import json
import re
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_l2/summary.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/round1/pmc_traffic_c3.json"
note = sys.argv[3] if len(sys.argv) > 3 else ""
res = json.load(open(src))
out = {"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/pmc_l2.sh), "
                   "bench.py workload c3; per-launch averages. hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                   "per MI355X_MICROARCH.md (gfx950 FETCH_SIZE halving; the guide calibrates this for 16-byte streaming "
                   "accesses only -- 8-byte gathers are uncalibrated -- and Infinity-Cache hits are counted, so this is fabric "
                   "traffic out of the L2s, an upper bound of HBM traffic). " + note,
       "kernels": {}}
for name, cs in sorted(res.items()):
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    k = re.sub(r"<.*>$", "", name)
    f, w = cs["FETCH_SIZE"]["avg"], cs["WRITE_SIZE"]["avg"]
    e = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "launches_sampled": cs["FETCH_SIZE"]["launches"],
         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    for extra in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "SQ_INSTS_VALU", "SQ_WAVES"):
        if extra in cs:
            e[extra] = round(cs[extra]["avg"], 1)
    if k in out["kernels"]:      # several instantiations of one template: keep the one with more launches... both
        k = name
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, len(out["kernels"]), "kernels")
