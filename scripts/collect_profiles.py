#!/usr/bin/env python3
"""gpurun_out/<NAME>/ (scripts/gpu_profile.sh) -> profiles/round<R>/: the files the docs cite, in the shapes they have there.
usage: collect_profiles.py [NAME] [ROUND]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r4final")
dst = os.path.join(ROOT, "profiles", "round" + (sys.argv[2] if len(sys.argv) > 2 else "5"))
os.makedirs(dst, exist_ok=True)
for seed in ("cov_c4_sq.json", "inflate_vec.json"):      # (the two files that are updated in place start from the round before's)
    if not os.path.exists(os.path.join(dst, seed)):
        shutil.copy(os.path.join(ROOT, "profiles", "round4", seed), os.path.join(dst, seed))
for a, b in (("c3_kernel_stats.csv", "c3_kernel_stats.csv"), ("pmc_traffic_c3.json", "pmc_traffic_c3.json"),
             ("cov_kernel_stats.csv", "cov_c4_kernel_stats.csv"), ("cov_c4_pmc.json", "cov_c4_pmc.json"),
             ("deflate_kernel_stats.csv", "deflate_kernel_stats.csv"), ("deflate_sq.json", "deflate_sq_counters.json")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
d = json.load(open(os.path.join(src, "cov_c4_sq.json")))
old = json.load(open(os.path.join(dst, "cov_c4_sq.json")))
old["kernels"] = {k: {c: v["avg"] for c, v in cs.items()} for k, cs in d.items() if k.startswith(("k_cov", "k_rs"))}
json.dump(old, open(os.path.join(dst, "cov_c4_sq.json"), "w"), indent=1)
iv = json.load(open(os.path.join(dst, "inflate_vec.json")))
for v in (0, 1):
    dd = json.load(open(os.path.join(src, f"inflate_sq_vec{v}.json")))
    iv[f"vec{v}"]["rate"] = open(os.path.join(src, f"inflate_vec{v}.txt")).read().strip()
    iv[f"vec{v}"]["counters"] = {k: {c: int(x["avg"]) for c, x in cs.items()} for k, cs in dd.items() if k.startswith("k_bgzf_inflate")}
json.dump(iv, open(os.path.join(dst, "inflate_vec.json"), "w"), indent=1)
print("copied; coverage kernels:", {k[:24]: round(v.get("SQ_INSTS_VALU", 0) / 1e6, 1) for k, v in old["kernels"].items() if "emit3" in k or "depths3" in k})
