#!/usr/bin/env python3
"""End-to-end host-pipeline timing of the C command line on the GPU box
(BAM file -> filtered BAM -> profile.txt.gz).  Reported separately from
bench.py's device-resident `value`."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools")
subprocess.call(["make", "-C", os.path.join(ROOT, "msamtools_amd", "csrc", "host")], stdout=subprocess.DEVNULL)
ngrp = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
refs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
T = "/tmp/msx_e2e"
os.makedirs(T, exist_ok=True)
res = {"cores": os.cpu_count(), "groups": ngrp, "refs": refs, "runs": []}


def timed(cmd, **kw):
    t = time.perf_counter()
    subprocess.check_call(cmd, shell=True, **kw)
    return time.perf_counter() - t


for flag in ("u", "b"):
    dt = timed(f"{B} synth --groups {ngrp} --refs {refs} -{flag} > {T}/in_{flag}.bam")
    res[f"synth_{flag}_s"] = round(dt, 2)
    res[f"size_{flag}_MB"] = round(os.path.getsize(f"{T}/in_{flag}.bam") / 1e6, 1)
n = int(subprocess.check_output(f"{B} recode {T}/in_u.bam | wc -l", shell=True))
res["records"] = n
filt = "filter -l 80 -p 95 -z 80 --besthit"
for inp in ("in_u", "in_b"):
    for th in (1, 8, 32):
        for outflag in ("-bu", "-b"):
            if outflag == "-b" and th != 32:
                continue
            dt = timed(f"MSX_THREADS={th} {B} {filt} {outflag} {T}/{inp}.bam > {T}/f.bam")
            res["runs"].append({"cmd": f"filter {outflag} {inp}", "threads": th, "s": round(dt, 3),
                                "M_aln_per_s": round(n / dt / 1e6, 3)})
dt = timed(f"{B} {filt} -bu {T}/in_b.bam | {B} profile --label S -o {T}/p.gz - 2> {T}/p.err")
res["runs"].append({"cmd": "filter -bu in_b | profile", "threads": "default", "s": round(dt, 3),
                    "M_aln_per_s": round(n / dt / 1e6, 3)})
dt = timed(f"{B} profile --label S -o {T}/p2.gz {T}/in_b.bam 2> {T}/p2.err")
res["runs"].append({"cmd": "profile in_b", "threads": "default", "s": round(dt, 3), "M_aln_per_s": round(n / dt / 1e6, 3)})
print(json.dumps(res, indent=1))
print(open(f"{T}/p.err").read()[-300:])
subprocess.call(f"rm -rf {T}", shell=True)
