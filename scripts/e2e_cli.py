#!/usr/bin/env python3
"""End-to-end host-pipeline timing of the C command line on the GPU box
(BAM file -> filtered BAM -> profile.txt.gz; SURVEY.md 8d metric (ii), BASELINE.md section 2 stage
split).  Reported separately from bench.py's device-resident `value`.  Also used by bench.py (--e2e).

usage: e2e_cli.py [groups] [refs] [--json-only]
"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools")
D = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
FILT = "filter -l 80 -p 95 -z 80 --besthit"


def timed(cmd, env=None):
    # (a leftover output of the previous run would be truncated by the shell inside the timed region: freeing
    #  1.7 GB of page cache costs 0.1-0.2 s)
    for tok in cmd.split():
        if tok.endswith("/f.bam") and os.path.exists(tok):
            os.remove(tok)
    t = time.perf_counter()
    r = subprocess.run(cmd, shell=True, env=dict(os.environ, **(env or {})), stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    if r.returncode != 0:
        raise RuntimeError(f"{cmd}: rc={r.returncode}\n{r.stderr.decode()[-2000:]}")
    return dt, r.stderr.decode()


def stage_times(err):
    """numbers of the '# filter pipeline:' / '# profile pipeline:' lines printed under MSX_TIMING"""
    out = {}
    for line in err.split("\n"):
        m = re.match(r"# (filter|profile) pipeline: (.*)", line)
        if not m:
            continue
        kind, rest = m.group(1), m.group(2)
        d = {}
        for key, pat in (("wall_s", r"wall ([0-9.]+) s"), ("decode_s", r"decode ([0-9.]+) s"),
                         ("hip_startup_s", r"start-up ([0-9.]+)"), ("upload_s", r"upload ([0-9.]+)"),
                         ("gpu_s", r"kernels ([0-9.]+)"), ("fetch_s", r"fetch ([0-9.]+)"),
                         ("upload_accumulate_s", r"upload\+accumulate ([0-9.]+)"),
                         ("encode_s", r"encode\+write ([0-9.]+) s"), ("threads", r"(\d+) threads")):
            mm = re.search(pat, rest)
            if mm:
                d[key] = float(mm.group(1)) if key != "threads" else int(mm.group(1))
        out[kind] = d
    for line in err.split("\n"):
        m = re.match(r"# decode stage: (.*)", line)
        if m:
            out.setdefault("decode_detail", []).append(m.group(1))
    return out


def run(ngrp=10_000_000, refs=100_000, tmp="/tmp/msx_e2e", levels=("u", "b"), verbose=True):
    subprocess.call(["make", "-C", os.path.join(ROOT, "msamtools_amd", "csrc", "host")], stdout=subprocess.DEVNULL)
    os.makedirs(tmp, exist_ok=True)
    res = {"host_cores": os.cpu_count(), "groups": ngrp, "refs": refs, "runs": []}
    try:
        for flag in levels:
            dt, _ = timed(f"{D} synth --groups {ngrp} --refs {refs} -{flag} > {tmp}/in_{flag}.bam")
            res[f"synth_{flag}_s"] = round(dt, 2)
            res[f"size_{flag}_MB"] = round(os.path.getsize(f"{tmp}/in_{flag}.bam") / 1e6, 1)
        n = int(subprocess.check_output(f"{D} recode {tmp}/in_{levels[0]}.bam | wc -l", shell=True))
        res["records"] = n
        res["bytes_per_record"] = round(res[f"size_{levels[0]}_MB"] * 1e6 / n, 1) if levels[0] == "u" else None
        env = {"MSX_TIMING": "1"}
        for inp in levels:
            for outflag in ("-bu", "-b"):
                dt, err = timed(f"{B} {FILT} {outflag} {tmp}/in_{inp}.bam > {tmp}/f.bam", env)
                res["runs"].append({"cmd": f"filter --besthit {outflag} in_{inp}.bam > f.bam", "s": round(dt, 3),
                                    "M_alignments_per_s": round(n / dt / 1e6, 2), "stages": stage_times(err)})
            dt, err = timed(f"{B} profile --label S -o {tmp}/p2.gz {tmp}/in_{inp}.bam", env)
            res["runs"].append({"cmd": f"profile in_{inp}.bam", "s": round(dt, 3), "M_alignments_per_s": round(n / dt / 1e6, 2),
                                "stages": stage_times(err)})
            # the reference's workflow: two processes, uncompressed BAM through the pipe
            if os.path.exists(f"{tmp}/f.bam"):
                os.remove(f"{tmp}/f.bam")
            t = time.perf_counter()
            p = subprocess.run(f"{B} {FILT} -bu {tmp}/in_{inp}.bam | {B} profile --label S -o {tmp}/p.gz -", shell=True,
                               env=dict(os.environ, **env), stderr=subprocess.PIPE)
            dt = time.perf_counter() - t
            if p.returncode != 0:
                raise RuntimeError(p.stderr.decode()[-2000:])
            st = stage_times(p.stderr.decode())
            res["runs"].append({"cmd": f"filter --besthit -bu in_{inp}.bam | profile -", "s": round(dt, 3),
                                "M_alignments_per_s": round(n / dt / 1e6, 2), "stages": st})
        # thread scaling of the decode-bound command
        for th in (8, 32):
            dt, err = timed(f"{B} {FILT} -bu {tmp}/in_{levels[-1]}.bam > {tmp}/f.bam", dict(env, MSX_THREADS=str(th)))
            res["runs"].append({"cmd": f"filter --besthit -bu in_{levels[-1]}.bam (MSX_THREADS={th})", "s": round(dt, 3),
                                "M_alignments_per_s": round(n / dt / 1e6, 2)})
    finally:
        subprocess.call(f"rm -rf {tmp}", shell=True)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ngrp = int(args[0]) if len(args) > 0 else 10_000_000
    refs = int(args[1]) if len(args) > 1 else 100_000
    print(json.dumps(run(ngrp, refs), indent=1))
