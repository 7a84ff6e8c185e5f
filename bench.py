#!/usr/bin/env python3
"""bench.py -- throughput of the msamtools hot path on MI355X.

A step = one pass of `filter -l 80 -p 95 -z 80 --besthit | profile
--multi=proportional` over one synthetic record batch that is already resident
in HBM (the metric of BASELINE.json).  Workload (default "c3",
BASELINE.json configs[2]): 20 M QNAME groups ~ 100 M alignments, 1 M
references, generated on the device by the library's deterministic generator.
With --gpus N each rank holds its own shard of that shape (weak scaling): the
filter/best-hit kernels need no communication; the per-reference count vector
and, per proportional-sharing iteration, the increment vector are all-reduced
over RCCL inside the library (msx_profile_finalize_dist_enqueue; torch.distributed is not
used: the launcher's environment -- RANK / WORLD_SIZE / MASTER_* -- only carries the
128-byte communicator id from rank 0 to the others).

Prints ONE JSON line on rank 0 (see the driver contract in the task prompt),
including `roofline` for the dominant kernel (HIP-event timings taken on the
library's own stream) and `cpu_baseline` (the CPU oracle, 1 thread, on a
bounded sample of the same stream; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: (n_groups per GPU, n_refs, description)
    "c3": (20_000_000, 1_000_000, "synthetic 100 M-alignment / 1 M-ref batch (BASELINE configs[2])"),
    "c2": (2_000_000, 10_000, "synthetic 10 M-alignment / 10 k-ref batch (BASELINE configs[1])"),
    "tiny": (20_000, 1_000, "smoke-size batch"),
}
FILTER_OPTS = dict(l=80, p=95, z=80, besthit=True)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
# HBM bytes per launch from rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE),
# filled from profiles/ when measured for the default workload; None = not measured.
TRAFFIC = {}
_PMC = next((p for p in (os.path.join(ROOT, "profiles", r, "pmc_traffic_c3.json") for r in ("round6", "round5", "round4", "round3")) if os.path.exists(p)),
            os.path.join(ROOT, "profiles", "round5", "pmc_traffic_c3.json"))
TRAFFIC_STALE = []          # kernel sources that have changed since the PMC passes were collected


def load_traffic(workload, ng, nrefs):
    """PMC-measured HBM bytes per launch; only valid for the workload they were collected on (c3 defaults).  The file
    records a hash of every kernel source it was collected on: when one has changed since, the line says so
    (roofline.traffic_stale_sources) and a warning goes to stderr -- the figure describes an earlier kernel."""
    if workload == "c3" and (ng, nrefs) == WORKLOADS["c3"][:2] and os.path.exists(_PMC):
        try:
            doc = json.load(open(_PMC))
            for k, v in doc["kernels"].items():
                TRAFFIC[k] = int(v["hbm_bytes_per_launch"])
            import hashlib
            for f, h in doc.get("source_sha16", {}).items():
                q = os.path.join(ROOT, "msamtools_amd", "csrc", f)
                if not os.path.exists(q) or hashlib.sha256(open(q, "rb").read()).hexdigest()[:16] != h:
                    TRAFFIC_STALE.append(f)
            if "source_sha16" not in doc:
                TRAFFIC_STALE.append("(no source hashes in the file: collected before round 4)")
            if TRAFFIC_STALE:
                print(f"bench.py: warning: {os.path.relpath(_PMC, ROOT)} was collected on other versions of {', '.join(TRAFFIC_STALE)}",
                      file=sys.stderr)
        except Exception:
            pass
SEED = 13579


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--groups", type=int, default=0, help="override QNAME groups per GPU")
    ap.add_argument("--refs", type=int, default=0, help="override number of references")
    ap.add_argument("--cpu-sample-groups", type=int, default=0,
                    help="QNAME groups the CPU oracle is timed on (0 = the whole batch: ~7 s per run on c3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the command-line end-to-end timing (BAM file in, BAM + profile out)")
    ap.add_argument("--e2e-groups", type=int, default=20_000_000,
                    help="QNAME groups of the end-to-end BAM (~5 records each; the default is the size of the c3 batch)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-GPU step (msx_profile_finalize_dist_enqueue over a one-rank RCCL communicator) "
                         "even with one rank")
    ap.add_argument("--print-checksum", action="store_true", help="add a checksum of the abundance vector to the JSON")
    ap.add_argument("--no-coverage", action="store_true", help="skip the coverage block (BASELINE configs[3])")
    ap.add_argument("--no-dist-parity", action="store_true",
                    help="N ranks: skip comparing the distributed result with one context doing all N shards")
    ap.add_argument("--no-dist-leg", action="store_true",
                    help="skip timing the multi-GPU step over a one-rank communicator (dist_one_rank_ms_per_step)")
    ap.add_argument("--e2e-seq-groups", type=int, default=20_000_000,
                    help="QNAME groups of the end-to-end BAM with SEQ/QUAL (~221 B per record); the default is BASELINE configs[2]'s "
                         "size (100 M records, 22 GB of BAM inflated) as for the lean file; 0 = skip")
    ap.add_argument("--e2e-sam-groups", type=int, default=4_000_000,
                    help="QNAME groups of the SAM-text leg (SEQ/QUAL records, ~20 M of them, ~7 GB of text: the reference's documented "
                         "workflow is an aligner piping SAM into `filter -S`); 0 = skip")
    ap.add_argument("--dry-launch", action="store_true",
                    help="ranks print their launch environment as JSON and exit (tests of the launcher; no GPU)")
    return ap.parse_args()


def granted_cpus():
    """CPUs this process may really use: its affinity mask and the cgroup's CPU quota (the GPU box grants the job
    fewer than it shows online)."""
    import math
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                q, per = parse(open(path).read())
                if q != "max":
                    n = min(n, max(1, math.ceil(int(q) / int(per))))
            else:
                q = int(open(path).read())
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0 and per > 0:
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except Exception:
            continue
    return max(1, n)


def launch_ranks(args):
    """`bench.py --gpus N` started plainly (no launcher, WORLD_SIZE unset): this process starts the N ranks itself,
    one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, relays rank 0's JSON
    line and exits non-zero if any rank does.  It never touches the GPU (no HIP call, no torch.cuda call): the
    children are fresh processes, nothing is re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MSX_BENCH_RANK_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"[bench] ranks failed: {bad}", file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def read_profile_gz(path):
    """(header lines, feature names, values[Unknown, features...]) of a profile written by the command line"""
    import gzip
    import numpy as np
    head, names, vals = [], [], []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                head.append(line.rstrip("\n"))
                continue
            k, _, v = line.rstrip("\n").partition("\t")
            if k == "ID" or not v:
                continue
            names.append(k)
            vals.append(float(v))
    return head, names, np.array(vals)


def profile_parity(path, stats, abundance, refs, ref_len):
    """The profile file against the oracle: header counts equal, every value within 1e-6 relative (+ the 5e-8 of
    the eight digits %.8g prints), a checksum of the values at six decimals (msam_profile.c:858-983)."""
    import hashlib
    import numpy as np
    import oracle_lib as orc
    head, names, got = read_profile_gz(path)

    def field(key):
        return next(l for l in head if l.startswith(key)).split(":")[1].split("(")[0].strip()
    vals, purged, eff = orc.profile_finish(abundance, np.full(refs, ref_len, np.uint32), stats, unit="rel")
    counts = (int(field("# Mapped inserts")), int(field("#   - Multiple mapped")), int(field("#   - Uniquely mapped")))
    want_counts = (int(stats.insert_count), int(stats.multi_mapper_count), int(stats.uniq_mapper_count))
    rel = float((np.abs(got - vals) / np.maximum(np.abs(vals), 1e-300)).max()) if got.shape == vals.shape else float("inf")
    return {"header_counts_equal": counts == want_counts, "mapped_multiple_unique": list(counts),
            "rows": int(got.size), "zero_pattern_equal": bool(got.shape == vals.shape and np.array_equal(got == 0, vals == 0)),
            "max_rel_err": rel, "tolerance": 1e-6 + 1e-7,
            "sha1_6dp": hashlib.sha1(np.round(got, 6).tobytes()).hexdigest(),
            "sha1_6dp_oracle": hashlib.sha1(np.round(vals, 6).tobytes()).hexdigest(),
            "ok": bool(counts == want_counts and rel <= 1e-6 + 1e-7)}


def inflate_probe(m, ctx, path, n_blocks=8192, skip_header=True):
    """msx_bgzf_inflate (msx_inflate.hip) on n_blocks blocks of a BAM file: rate, the device's own CRC verdicts, and EVERY block
    against zlib.  skip_header: the blocks behind the header's (a million @SQ lines are 700 blocks of text whose long tokens the
    lanes do not synchronise on: they are handed to the serial kernel -- measured beside it as `with_header_blocks`)."""
    import ctypes as C
    import struct
    import zlib
    import numpy as np
    from msamtools_amd import _lib as L
    skip_bytes = 0
    if skip_header:
        # the header's inflated size: magic 4, l_text 4, text, n_ref 4, then per reference l_name 4 + name + l_ref 4 (the synthetic
        # files' names all have one length)
        import gzip
        with gzip.open(path, "rb") as g:
            l_text = struct.unpack("<4si", g.read(8))[1]
            g.read(l_text)
            n_ref = struct.unpack("<i", g.read(4))[0]
            l_name = struct.unpack("<i", g.read(4))[0] if n_ref else 0
        skip_bytes = 12 + l_text + n_ref * (8 + l_name)
    with open(path, "rb") as fh:
        raw = fh.read((n_blocks + skip_bytes // 60000 + 64) * 70000)
    blocks, pos, inflated = [], 0, 0
    while pos + 18 <= len(raw) and len(blocks) < n_blocks:
        xlen = struct.unpack_from("<H", raw, pos + 10)[0]
        bsize = struct.unpack_from("<H", raw, pos + 16)[0] + 1
        if pos + bsize > len(raw):
            break
        crc, isize = struct.unpack_from("<II", raw, pos + bsize - 8)
        if inflated >= skip_bytes:
            blocks.append((pos + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
        inflated += isize
        pos += bsize
    n = len(blocks)
    arr = (L.BgzfBlock * n)()
    uo = 0
    for i, (io, il, ol, crc) in enumerate(blocks):
        arr[i].in_off, arr[i].in_len, arr[i].out_off, arr[i].out_len, arr[i].crc32 = io, il, uo, ol, crc
        uo += ol
    d_comp, d_blk, d_out, d_st = ctx.alloc(pos + 64), ctx.alloc(32 * n), ctx.alloc(uo + 64), ctx.alloc(4 * n)
    try:
        ctx.to_dev(d_comp, np.frombuffer(raw[:pos], np.uint8))
        ctx.to_dev(d_blk, np.frombuffer(bytes(arr), np.uint8))
        ref = C.c_int64()

        def timed():
            ts = []
            for _ in range(5):
                ctx.sync()
                t0 = time.perf_counter()
                ctx.check(ctx.lib.msx_bgzf_inflate(ctx.h, C.c_void_p(d_comp), pos, C.c_void_p(d_blk), n, C.c_void_p(d_out), C.c_void_p(d_st),
                                                   C.byref(ref)))
                ts.append(time.perf_counter() - t0)
            return min(ts[1:]), int(ref.value), ctx.to_host(d_out, uo, np.uint8).tobytes()
        best, refused, out = timed()
        # EVERY block against zlib (round 5 compared a sample of 256)
        t0 = time.perf_counter()
        want = b"".join(zlib.decompress(raw[io:io + il], -15) for io, il, ol, crc in blocks)
        zs = time.perf_counter() - t0
        ok = out == want
        # the serial kernel of rounds 3-5 (MSX_INFLATE_SERIAL=1: one wave per block, one symbol after the other) on the same blocks
        ctx.to_dev(d_out, np.zeros(uo, np.uint8))
        os.environ["MSX_INFLATE_SERIAL"] = "1"
        try:
            best_l, refused_l, out_l = timed()
        finally:
            del os.environ["MSX_INFLATE_SERIAL"]
        return {"blocks": n, "compressed_MB": round(pos / 1e6, 1), "inflated_MB": round(uo / 1e6, 1), "ms": round(best * 1e3, 3),
                "GBps_inflated": round(uo / best / 1e9, 1), "blocks_refused": refused,
                "every_block_equals_zlib": bool(ok), "blocks_compared": n,
                "zlib_one_core_GBps": round(uo / zs / 1e9, 2),
                "frac_of_hbm_peak": round((pos + uo) / best / 8e12, 4),
                "bound": "not memory: the 64 lanes of a wave decode one deflate block's symbols at once (self-synchronising Huffman "
                         "decoding, 5-6 walks per segment), vector-instruction issue; DESIGN.md section 3, profiles/round6/inflate_lanes.md",
                "serial_kernel": {"ms": round(best_l * 1e3, 3), "GBps_inflated": round(uo / best_l / 1e9, 1), "blocks_refused": refused_l,
                                  "every_block_equals_zlib": bool(out_l == want),
                                  "note": "MSX_INFLATE_SERIAL=1: rounds 3-5's kernel, now the fallback for blocks the lanes hand back"}}
    finally:
        for q in (d_comp, d_blk, d_out, d_st):
            ctx.free(q)


def coverage_cli(groups=10_000_000, refs=50_000):
    """`msamtools coverage` end to end on a c4-like BAM file (BASELINE configs[3]: 50 k references, ~50 M alignments;
    BGZF level 6): --summary and the per-position text, each also through the serial record-at-a-time reader
    (MSX_SERIAL_IO=1) as a cross-check of the outputs (the pile-up itself is checked against the oracle in the
    `coverage` block above and, through the command line, in tests/test_cli_scale.py)."""
    import gzip
    import hashlib
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools")
    dev = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")      # synth / digest: generator and self-tests, not in the product binary
    if not os.path.exists(exe) or not os.path.exists(dev):
        return {"error": "msamtools_amd/bin/msamtools or msamtools-dev not built"}
    tmp = tempfile.mkdtemp(prefix="msx_cov_", dir="/tmp")

    def md5(path):
        h = hashlib.md5()
        with gzip.open(path, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
        return h.hexdigest()

    def run(cmd, **env):
        time.sleep(1.0)
        t = time.perf_counter()
        r = subprocess.run(cmd, shell=True, env=dict(os.environ, **env), stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-300:])
        return time.perf_counter() - t
    try:
        subprocess.check_call(f"{dev} synth --groups {groups} --refs {refs} -b > {tmp}/in.bam", shell=True)
        n = int(subprocess.check_output([dev, "digest", f"{tmp}/in.bam"]).decode().split()[0].split("=")[1])
        dt_s = run(f"{exe} coverage --summary -o {tmp}/s.gz {tmp}/in.bam")
        dt_t = run(f"{exe} coverage -o {tmp}/t.gz {tmp}/in.bam")
        dt_s1 = run(f"{exe} coverage --summary -o {tmp}/s1.gz {tmp}/in.bam", MSX_SERIAL_IO="1")
        res = {"records": n, "references": refs, "positions": refs * 4496,
               "summary": {"seconds": round(dt_s, 3), "M_alignments_per_s": round(n / dt_s / 1e6, 1)},
               "per_position_text": {"seconds": round(dt_t, 3), "M_alignments_per_s": round(n / dt_t / 1e6, 1),
                                     "gz_MB": round(os.path.getsize(f"{tmp}/t.gz") / 1e6, 1)},
               "summary_serial_reader": {"seconds": round(dt_s1, 3)},
               "outputs_equal_serial_reader": md5(f"{tmp}/s.gz") == md5(f"{tmp}/s1.gz")}
        return res
    except Exception as exc:
        return {"error": str(exc)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def e2e_cli(groups, refs, expect=None, seq=False, probe=None, marginal_groups=0, cpu_records=0, out_flag="-bu"):
    """The command line end to end on this box: a synthetic BAM of `groups` QNAME groups (BGZF level 6; records
    without SEQ/QUAL, or ~250 B records with them: seq) through `msamtools filter -l 80 -p 95 -z 80 --besthit -bu |
    msamtools profile -` -- the reference's own two-process workflow -- through either command alone, and through
    the one-process form `filter --profile-out`.  Host-bound (BGZF inflate, record walk, deflate); reported next to
    `value`, never as it (SURVEY.md 8d metric (ii)).  The outputs are kept and compared with
    the oracle's for the same stream (expect: emit digest, stats and abundances of the oracle's pipe and of the
    oracle's plain profile) -- which records, in which order; header counts; every value."""
    import re
    import shutil
    import subprocess
    import tempfile

    import numpy as np
    exe = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools")
    dev = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
    if not os.path.exists(exe) or not os.path.exists(dev):
        return {"error": "msamtools_amd/bin/msamtools or msamtools-dev not built"}
    tmp = tempfile.mkdtemp(prefix="msx_e2e_", dir="/tmp")
    filt = f"filter -l 80 -p 95 -z 80 --besthit {out_flag}"
    env = dict(os.environ, MSX_TIMING="1")
    ref_len = 4496              # msh_dev.c: synth_main

    def stages(err, kind):
        d = {}
        for line in err.split("\n"):
            if line.startswith(f"# {kind} pipeline:"):
                for key, pat in (("wall_s", r"wall ([0-9.]+) s"), ("decode_s", r"decode ([0-9.]+) s"), ("hip_startup_s", r"start-up ([0-9.]+)"),
                                 ("upload_s", r"upload ([0-9.]+)"), ("gpu_s", r"kernels ([0-9.]+)"), ("fetch_s", r"fetch ([0-9.]+)"),
                                 ("upload_accumulate_s", r"upload\+accumulate ([0-9.]+)"), ("encode_s", r"encode\+write ([0-9.]+) s"),
                                 ("threads", r"(\d+) threads")):
                    mm = re.search(pat, line)
                    if mm:
                        d[key] = float(mm.group(1)) if key != "threads" else int(mm.group(1))
        return d

    def run(cmd, reps=2):
        # the better of two runs (the first run of a command on a fresh box pays for cold page-cache pages of its output file
        # and whatever else the box was doing: +-10 % between runs; every e2e figure of this block is timed this way)
        best = None
        for _ in range(reps):
            time.sleep(1.0)     # (the device is still releasing the previous process's memory right after it exits)
            t = time.perf_counter()
            r = subprocess.run(cmd, shell=True, env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                raise RuntimeError(r.stderr.decode()[-500:])
            if best is None or dt < best[0]:
                best = (dt, r.stderr.decode())
        return best

    def digest(path):
        out = subprocess.check_output([dev, "digest", path]).decode().split()
        return int(out[0].split("=")[1]), out[1].split("=")[1]
    try:
        t0 = time.perf_counter()
        subprocess.check_call(f"{dev} synth --groups {groups} --refs {refs} {'--seq' if seq else ''} -b > {tmp}/in.bam", shell=True)
        synth_s = time.perf_counter() - t0
        n, _ = digest(f"{tmp}/in.bam")
        # bytes per record: a small uncompressed sample with a short header (the records do not depend on it)
        subprocess.check_call(f"{dev} synth --groups {min(groups, 100000)} --refs 1000 {'--seq' if seq else ''} -u > {tmp}/s.bam", shell=True)
        n_s, _ = digest(f"{tmp}/s.bam")
        size_u = os.path.getsize(f"{tmp}/s.bam")
        dt_f, err_f = run(f"{exe} {filt} {tmp}/in.bam > {tmp}/f.bam")
        dt_p, err_p = run(f"{exe} profile --label S -o {tmp}/p1.gz {tmp}/in.bam")
        dt_fp, err_fp = run(f"{exe} {filt} {tmp}/in.bam | {exe} profile --label S -o {tmp}/p.gz -")
        tee = None
        try:
            dt_t, err_t = run(f"{exe} {filt} --profile-out {tmp}/pt.gz --label S {tmp}/in.bam > {tmp}/ft.bam")
            tee = {"M_alignments_per_s": round(n / dt_t / 1e6, 2), "seconds": round(dt_t, 3),
                   "command": f"msamtools {filt} --profile-out p.gz --label S in.bam > f.bam", **stages(err_t, "filter")}
            # the pipe never stores the alignments either: the same command with its BAM output discarded
            dt_n, err_n = run(f"{exe} {filt} --profile-out {tmp}/pn.gz --label S {tmp}/in.bam > /dev/null")
            tee["alignments_discarded"] = {"M_alignments_per_s": round(n / dt_n / 1e6, 2), "seconds": round(dt_n, 3),
                                           "note": "stdout to /dev/null (filter's BAM is formed and framed, not stored)"}
            # A/B: the same command with BGZF inflate on the host cores instead of the device (MSX_HOST_INFLATE=1)
            env["MSX_HOST_INFLATE"] = "1"
            try:
                dt_h, err_h = run(f"{exe} {filt} --profile-out {tmp}/ph.gz --label S {tmp}/in.bam > {tmp}/fh.bam", reps=1)
            finally:
                del env["MSX_HOST_INFLATE"]
            tee["host_inflate"] = {"M_alignments_per_s": round(n / dt_h / 1e6, 2), "seconds": round(dt_h, 3), **stages(err_h, "filter")}
            os.remove(f"{tmp}/fh.bam")
        except RuntimeError as exc:
            tee = {"error": str(exc)[:200]}
        sf, sp = stages(err_fp, "filter"), stages(err_fp, "profile")
        res = {
            "M_alignments_per_s": round(n / dt_fp / 1e6, 2),
            "command": f"msamtools {filt} in.bam | msamtools profile --label S -o p.gz -",
            "seconds": round(dt_fp, 3), "records": n, "references_in_header": refs,
            "decode_s": sf.get("decode_s"), "upload_s": sf.get("upload_s"), "gpu_s": sf.get("gpu_s"), "encode_s": sf.get("encode_s"),
            "profile_decode_s": sp.get("decode_s"), "profile_upload_accumulate_s": sp.get("upload_accumulate_s"),
            "threads": sf.get("threads"), "host_cpus_online": os.cpu_count(), "host_cpus_granted": granted_cpus(),
            "bgzf_level": {"input": 6, "pipe": 0}, "encoding": "SEQ/QUAL present" if seq else "lean (l_seq = 0)",
            "bytes_per_record": round(size_u / max(1, n_s), 1),
            "input_MB": round(os.path.getsize(f"{tmp}/in.bam") / 1e6, 1),
            "filter_alone": {"M_alignments_per_s": round(n / dt_f / 1e6, 2), "seconds": round(dt_f, 3), **stages(err_f, "filter")},
            "profile_alone": {"M_alignments_per_s": round(n / dt_p / 1e6, 2), "seconds": round(dt_p, 3), **stages(err_p, "profile")},
            "one_process_tee": tee,
            "synth_s": round(synth_s, 1),
            "note": "stage times are busy times of overlapping pipeline stages (decode | device | encode), not a sum; every command of this block is timed as the better of two runs",
        }
        # ---- compressed output (the reference's -b: htslib deflate at level 6, msam_filter.c:464-470) ----
        # device DEFLATE (msx_deflate.hip) against zlib level 6 on the granted cores (MSX_HOST_DEFLATE=1, round 3's path): time,
        # size, the same records in the same order, every block readable by the host inflater + zlib (digest) and by the
        # device inflater (profile of the compressed output: its counts are those of the one-process form)
        if out_flag == "-bu":
            try:
                fb = filt.replace("-bu", "-b")
                dt_b, err_b = run(f"{exe} {fb} --profile-out {tmp}/pb.gz --label S {tmp}/in.bam > {tmp}/fb.bam", reps=1)
                dt_b = min(dt_b, run(f"{exe} {fb} --profile-out {tmp}/pb.gz --label S {tmp}/in.bam > {tmp}/fb.bam", reps=1)[0])
                nb_, dgb = digest(f"{tmp}/fb.bam")
                size_dev = os.path.getsize(f"{tmp}/fb.bam")
                env["MSX_HOST_DEFLATE"] = "1"
                try:
                    dt_z, err_z = run(f"{exe} {fb} --profile-out {tmp}/pz.gz --label S {tmp}/in.bam > {tmp}/fz.bam", reps=1)
                finally:
                    del env["MSX_HOST_DEFLATE"]
                nz_, dgz = digest(f"{tmp}/fz.bam")
                size_z = os.path.getsize(f"{tmp}/fz.bam")
                dt_pb, err_pb = run(f"{exe} profile --label S -o {tmp}/pbb.gz {tmp}/fb.bam")
                comp = {"command": f"msamtools {fb} --profile-out p.gz --label S in.bam > f.bam",
                        "M_alignments_per_s": round(n / dt_b / 1e6, 2), "seconds": round(dt_b, 3), **stages(err_b, "filter"),
                        "output_MB": round(size_dev / 1e6, 1),
                        "zlib_level6_on_host": {"M_alignments_per_s": round(n / dt_z / 1e6, 2), "seconds": round(dt_z, 3),
                                                "output_MB": round(size_z / 1e6, 1), **stages(err_z, "filter")},
                        "size_ratio_to_zlib6": round(size_dev / max(size_z, 1), 4),
                        "records_and_order_equal_to_zlib_output": bool((nb_, dgb) == (nz_, dgz)),
                        "read_back_through_device_inflater": "BGZF blocks inflated on the device" in err_pb,
                        "_digest": (nb_, dgb)}
                res["compressed_out"] = comp
                for f in ("fz.bam", "pz.gz"):
                    os.remove(f"{tmp}/{f}")
                # the reference's two processes with a COMPRESSED pipe (-b in the place of -bu: htslib's "wb", msam_filter.c:464-470):
                # a ninth of the bytes through the 1 MB pipe, deflated on the device on one side, inflated on the device on the other
                dt_cp, err_cp = run(f"{exe} {fb} {tmp}/in.bam | {exe} profile --label S -o {tmp}/pcp.gz -", reps=1)
                dt_cp = min(dt_cp, run(f"{exe} {fb} {tmp}/in.bam | {exe} profile --label S -o {tmp}/pcp.gz -", reps=1)[0])
                res["pipe_compressed"] = {"command": f"msamtools {fb} in.bam | msamtools profile --label S -o p.gz -",
                                          "M_alignments_per_s": round(n / dt_cp / 1e6, 2), "seconds": round(dt_cp, 3),
                                          "inflated_on_the_device_by_profile": "BGZF blocks inflated on the device" in err_cp}
            except Exception as exc:
                res["compressed_out"] = {"error": str(exc)[:300]}
        # ---- the pipeline's own rate: a second, smaller file, and what the additional records cost ----
        # (a whole-process figure carries ~0.35 s of start-up -- HIP, page-locking, process start and exit -- whatever the
        #  file's size; the marginal rate between two sizes is what a longer file would run at)
        if marginal_groups and tee and "error" not in tee:
            try:
                subprocess.check_call(f"{dev} synth --groups {marginal_groups} --refs {refs} {'--seq' if seq else ''} -b > {tmp}/in2.bam", shell=True)
                n2, _ = digest(f"{tmp}/in2.bam")
                cmd2 = f"{exe} {filt} --profile-out {tmp}/pt2.gz --label S {tmp}/in2.bam > {tmp}/ft2.bam"
                dt2 = min(run(cmd2, reps=1)[0], run(cmd2, reps=1)[0])
                dt1 = min(dt_t, run(f"{exe} {filt} --profile-out {tmp}/pt.gz --label S {tmp}/in.bam > {tmp}/ft.bam", reps=1)[0])
                tee["marginal"] = {"records_small": n2, "seconds_small": round(dt2, 3), "records_large": n, "seconds_large": round(dt1, 3),
                                   "marginal_M_alignments_per_s": round((n - n2) / max(dt1 - dt2, 1e-9) / 1e6, 1),
                                   "note": "delta records / delta seconds of the one-process form between the two files (best of two runs each)"}
                os.remove(f"{tmp}/in2.bam")
            except Exception as exc:
                tee["marginal"] = {"error": str(exc)[:200]}
        # ---- the reference's execution model on this box's host cores, same file (CPU-1 / CPU-N) ----
        cpu = os.path.join(ROOT, "oracle", "msx_cpu_e2e")
        if cpu_records and os.path.exists(cpu):
            try:
                def cpu_run(threads, level):
                    out = subprocess.check_output([cpu, f"{tmp}/in.bam", str(cpu_records), str(threads), f"{tmp}/cpu.bam", str(level)])
                    d = json.loads(out.decode())
                    os.remove(f"{tmp}/cpu.bam")
                    return d
                g = granted_cpus()
                res["cpu_baseline_e2e"] = {
                    "kind": "port", "what": "oracle/msx_cpu_e2e.c: zlib inflate + CRC, record walk, one aux scan per record (decode); orc_filter "
                    "--besthit + orc_profile proportional (compute); gather + BGZF blocks + write (encode) on a prefix of the same file, "
                    "1 M-reference header included; one thread = the reference's execution model",
                    "sample": f"the first {cpu_records} records of the e2e BAM (whole QNAME groups)",
                    "cpu_1_thread_bu": cpu_run(1, 0), "cpu_1_thread_b_level6": cpu_run(1, 6),
                    f"cpu_{g}_threads_bu": cpu_run(g, 0), f"cpu_{g}_threads_b_level6": cpu_run(g, 6),
                }
            except Exception as exc:
                res["cpu_baseline_e2e"] = {"error": str(exc)[:300]}
        if probe is not None:
            try:
                res["inflate"] = probe(f"{tmp}/in.bam")
                wh = probe(f"{tmp}/in.bam", skip_header=False)
                res["inflate"]["with_header_blocks"] = {k: wh.get(k) for k in ("blocks", "ms", "GBps_inflated", "blocks_refused", "every_block_equals_zlib")}
                res["inflate"]["with_header_blocks"]["note"] = ("the file's first blocks, its header's among them: text with 25-bit tokens "
                                                              "that the lanes hand back to the serial kernel")
            except Exception as exc:
                res["inflate"] = {"error": str(exc)[:200]}
        if expect is not None:
            # ---- the outputs against the oracle's for the same stream ----
            par = {}
            n_out, dg = digest(f"{tmp}/f.bam")
            par["filter_records_out"] = n_out
            par["filter_records_oracle"] = int(expect["n_emit"])
            if "emit_digest" in expect:
                par["filter_digest"] = dg
                par["filter_digest_oracle"] = f"{expect['emit_digest']:016x}"
                par["filter_ok"] = bool(n_out == expect["n_emit"] and dg == par["filter_digest_oracle"])
            else:
                par["filter_ok"] = bool(n_out == expect["n_emit"])
            full_want = None
            if "emit" in expect:
                # every byte of every record: the output's whole-record digest against the input's records at the oracle's emit
                # list, in its order (msamtools-dev digest --full [--select]; tests/test_digest_cpu.py)
                np.asarray(expect["emit"]).astype("<u4").tofile(f"{tmp}/emit.u32")
                full_want = subprocess.check_output([dev, "digest", "--full", "--select", f"{tmp}/emit.u32", f"{tmp}/in.bam"]).decode().strip()
                full = lambda path: subprocess.check_output([dev, "digest", "--full", path]).decode().strip()
                par["filter_bytes_equal_selected_input"] = bool(full(f"{tmp}/f.bam") == full_want)
                par["filter_ok"] = bool(par["filter_ok"] and par["filter_bytes_equal_selected_input"])
            if "pipe" in expect:
                par["pipe_profile"] = profile_parity(f"{tmp}/p.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, ref_len)
            if "plain" in expect:
                par["plain_profile"] = profile_parity(f"{tmp}/p1.gz", expect["plain"]["stats"], expect["plain"]["abundance"], refs, ref_len)
            if tee and "error" not in tee:
                nt, dgt = digest(f"{tmp}/ft.bam")
                par["tee_filter_ok"] = bool((nt, dgt) == (n_out, dg) and (full_want is None or full(f"{tmp}/ft.bam") == full_want))
                if "pipe" in expect:
                    par["tee_profile"] = profile_parity(f"{tmp}/pt.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, ref_len)
            co = res.get("compressed_out")
            if isinstance(co, dict) and "_digest" in co:
                nb_, dgb = co.pop("_digest")
                par["compressed_filter_ok"] = bool((nb_, dgb) == (n_out, dg) and co["records_and_order_equal_to_zlib_output"]
                                                   and (full_want is None or full(f"{tmp}/fb.bam") == full_want))
                if full_want is not None:
                    co["bytes_equal_selected_input"] = bool(full(f"{tmp}/fb.bam") == full_want)
                if "pipe" in expect:
                    par["compressed_tee_profile"] = profile_parity(f"{tmp}/pb.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, ref_len)
                    par["profile_of_compressed_output"] = profile_parity(f"{tmp}/pbb.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, ref_len)
                    if isinstance(res.get("pipe_compressed"), dict) and os.path.exists(f"{tmp}/pcp.gz"):
                        par["compressed_pipe_profile"] = profile_parity(f"{tmp}/pcp.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, ref_len)
                        res["pipe_compressed"]["parity_ok"] = bool(par["compressed_pipe_profile"].get("ok"))
                co["parity_ok"] = bool(par["compressed_filter_ok"] and par.get("compressed_tee_profile", {}).get("ok", True)
                                       and par.get("profile_of_compressed_output", {}).get("ok", True) and co["read_back_through_device_inflater"])
            par["parity_ok"] = bool(par["filter_ok"] and par.get("compressed_filter_ok", True)
                                    and all(v.get("ok", True) for v in par.values() if isinstance(v, dict))
                                    and par.get("tee_filter_ok", True))
            res["parity"] = par
            res["parity_ok"] = par["parity_ok"]
        if isinstance(res.get("compressed_out"), dict):
            res["compressed_out"].pop("_digest", None)
        return res
    except Exception as exc:
        return {"error": str(exc)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def headline(out):
    """The figures a reader of the line looks for first, in one small object at its very end: the step, the dominant kernel's
    fraction of the roof, and -- in M alignments/s -- the reference's two-process pipe (-bu and -b), the one-process form (-bu and
    -b), the same with SEQ/QUAL records (configs[2]'s size), the inflater's rate and coverage's time."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    e, q = out.get("e2e"), out.get("e2e_seq")
    h = {"ms_step": out.get("ms_per_step"), "frac": g(out, "roofline", "frac"),
         "pipe_bu": g(e, "M_alignments_per_s"), "pipe_b": g(e, "pipe_compressed", "M_alignments_per_s"),
         "one_bu": g(e, "one_process_tee", "M_alignments_per_s"), "one_b": g(e, "compressed_out", "M_alignments_per_s"),
         "marginal_bu": g(e, "one_process_tee", "marginal", "marginal_M_alignments_per_s"),
         "seq_records": g(q, "records"), "seq_pipe_bu": g(q, "M_alignments_per_s"), "seq_one_bu": g(q, "one_process_tee", "M_alignments_per_s"),
         "seq_one_b": g(q, "compressed_out", "M_alignments_per_s"),
         "inflate_GBps": g(e, "inflate", "GBps_inflated"), "seq_inflate_GBps": g(q, "inflate", "GBps_inflated"),
         "sam_pipe": g(out, "e2e_sam", "pipe", "M_alignments_per_s"), "sam_one": g(out, "e2e_sam", "one_process", "M_alignments_per_s"),
         "cov_ms": g(out, "coverage", "ms"), "e2e_parity": g(e, "parity_ok"), "seq_parity": g(q, "parity_ok"),
         "unit": "M alignments/s unless named"}
    return {k: v for k, v in h.items() if v is not None}


def e2e_sam(groups, refs, expect=None):
    """The reference's documented workflow is SAM TEXT on stdin from the aligner (`bwa-mem2 mem ... | msamtools filter -S -bu ...
    --besthit - | msamtools profile ... -`, the reference's README.md:133-134, :197-199; input mode "r", msam.h:105): `groups`
    QNAME groups of SEQ/QUAL records as SAM text, written by `cat` into the pipe, through (a) that two-process pipe and (b) the
    one-process form; the text is parsed into BAM records on the host cores (msh_sam.c, all threads), everything behind that is
    the pipeline of the BAM legs.  Parity: filter's records against the oracle's emit list for the same stream (count and the
    order-sensitive digest of QNAME / FLAG / tid / pos), the profile's header counts against the oracle's."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools")
    dev = os.path.join(ROOT, "msamtools_amd", "bin", "msamtools-dev")
    tmp = tempfile.mkdtemp(prefix="msx_sam_", dir="/tmp")
    filt = "filter -S -l 80 -p 95 -z 80 --besthit -bu"
    env = dict(os.environ, MSX_TIMING="1")

    def run(cmd, reps=2):
        best = None                                   # (the better of two runs, as in e2e_cli)
        for _ in range(reps):
            time.sleep(1.0)
            t = time.perf_counter()
            r = subprocess.run(cmd, shell=True, env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t
            if r.returncode != 0:
                raise RuntimeError(r.stderr.decode()[-500:])
            if best is None or dt < best[0]:
                best = (dt, r.stderr.decode())
        return best

    def stage(err, kind, key, pat):
        for line in err.split("\n"):
            if line.startswith(f"# {kind} pipeline:"):
                mm = re.search(pat, line)
                if mm:
                    return float(mm.group(1))
        return None
    try:
        t0 = time.perf_counter()
        subprocess.check_call(f"{dev} synth --groups {groups} --refs {refs} --seq -h > {tmp}/in.sam", shell=True)
        synth_s = time.perf_counter() - t0
        size = os.path.getsize(f"{tmp}/in.sam")
        subprocess.check_call(f"cat {tmp}/in.sam > /dev/null", shell=True)
        dt_p, err_p = run(f"cat {tmp}/in.sam | {exe} {filt} - | {exe} profile --label S -o {tmp}/p.gz -", reps=1)
        dt_p2, err_p2 = run(f"cat {tmp}/in.sam | {exe} {filt} - | {exe} profile --label S -o {tmp}/p.gz -", reps=1)
        if dt_p2 < dt_p:
            dt_p, err_p = dt_p2, err_p2
        dt_t, err_t = run(f"cat {tmp}/in.sam | {exe} {filt} --profile-out {tmp}/pt.gz --label S - > {tmp}/ft.bam", reps=1)
        dt_t2, err_t2 = run(f"cat {tmp}/in.sam | {exe} {filt} --profile-out {tmp}/pt.gz --label S - > {tmp}/ft.bam", reps=1)
        if dt_t2 < dt_t:
            dt_t, err_t = dt_t2, err_t2
        out = subprocess.check_output([dev, "digest", f"{tmp}/ft.bam"]).decode().split()
        n_out, dg = int(out[0].split("=")[1]), out[1].split("=")[1]
        n = int(subprocess.check_output(f"grep -vc '^@' {tmp}/in.sam", shell=True).decode())
        res = {"records": n, "text_MB": round(size / 1e6, 1), "bytes_per_record": round(size / max(n, 1), 1), "synth_s": round(synth_s, 1),
               "pipe": {"command": f"cat in.sam | msamtools {filt} - | msamtools profile --label S -o p.gz -",
                        "M_alignments_per_s": round(n / dt_p / 1e6, 2), "seconds": round(dt_p, 3),
                        "filter_decode_s": stage(err_p, "filter", "decode_s", r"decode ([0-9.]+) s"),
                        "filter_wall_s": stage(err_p, "filter", "wall_s", r"wall ([0-9.]+) s")},
               "one_process": {"command": f"cat in.sam | msamtools {filt} --profile-out p.gz --label S - > f.bam",
                               "M_alignments_per_s": round(n / dt_t / 1e6, 2), "seconds": round(dt_t, 3),
                               "decode_s": stage(err_t, "filter", "decode_s", r"decode ([0-9.]+) s"),
                               "wall_s": stage(err_t, "filter", "wall_s", r"wall ([0-9.]+) s"),
                               "text_GBps": round(size / dt_t / 1e9, 2)},
               "threads": granted_cpus(),
               "bound": "the text is parsed into BAM records on the granted host cores (decode_s is that stage's busy time)"}
        if expect is not None:
            par = {"filter_records_out": n_out, "filter_records_oracle": int(expect["n_emit"])}
            par["filter_ok"] = bool(n_out == expect["n_emit"] and ("emit_digest" not in expect or dg == f"{expect['emit_digest']:016x}"))
            if "pipe" in expect:
                par["pipe_profile"] = profile_parity(f"{tmp}/p.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, 4496)
                par["one_process_profile"] = profile_parity(f"{tmp}/pt.gz", expect["pipe"]["stats"], expect["pipe"]["abundance"], refs, 4496)
            par["parity_ok"] = bool(par["filter_ok"] and all(v.get("ok", True) for v in par.values() if isinstance(v, dict)))
            res["parity"] = par
            res["parity_ok"] = par["parity_ok"]
        return res
    except Exception as exc:
        return {"error": str(exc)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def exchange_comm_id(m, rank, world):
    """The RCCL communicator id, made on rank 0, to every rank: through the launcher's key-value store
    (torchrun hosts one on MASTER_ADDR:MASTER_PORT) or, failing that, the library's own TCP rendezvous
    (msx_dist_init_env).  Returns the id bytes, or None for "use msx_dist_init_env"."""
    if world == 1:
        return m.dist_unique_id()
    try:
        from datetime import timedelta
        from torch.distributed import TCPStore
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")),
                         world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=300),
                         wait_for_workers=False)
        key = "msx_comm_id/" + os.environ.get("TORCHELASTIC_RUN_ID", "0")
        if rank == 0:
            uid = m.dist_unique_id()
            store.set(key, uid)
        else:
            uid = bytes(store.get(key))
        return uid
    except Exception as exc:      # no store reachable: the C rendezvous on MSX_DIST_PORT / MASTER_PORT + 17
        print(f"[bench rank {rank}] launcher store unavailable ({exc}); using the library's TCP rendezvous",
              file=sys.stderr, flush=True)
        return None


def algorithmic_bytes_per_step(name, w):
    """Bytes the launches of `name` in ONE step must move once, or None
    when the kernel is not priced.  w: n (records), ng (pools), n_cig, n_md, n_emit, uniq, L0/E0 (multi-mapper
    lists / entries as accumulated), L/E (after identical sets were merged), nf (features), iters (sharing
    iterations that ran), lib (bytes the library attached to its own launches: scans and radix passes, each
    priced by its own length)."""
    n, ng, E, L, E0, L0, nf, it = w["n"], w["ng"], w["E"], w["L"], w["E0"], w["L0"], w["nf"], w["iters"]
    if name == "k_aln_stats_flat":
        # reads flag 2, rflags 1, cigar_off 4, cigar, md_off 4, md ; writes the pool byte
        return 2 * n + n + 4 * (n + 1) + 4 * w["n_cig"] + 4 * (n + 1) + w["n_md"] + n
    if name == "k_besthit_select":
        # reads group_off, pool byte, AS ; writes keep, per-pool count ; fused insert accounting: reads tid of
        # the kept records ; writes the per-pool word 4 and the distinct features of multi-mapped pools
        return 4 * (ng + 1) + n + 4 * n + n + 4 * ng + 4 * w["n_emit"] + 4 * ng + 4 * E0
    if name == "k_emit_order":
        return 4 * (ng + 1) + n + 4 * (ng + 1) + 4 * w["n_emit"]
    if name == "k_insert_count":
        # counting the unique-insert keys by partition: histogram reads the keys, scatter reads them again and
        # writes the kept ones, the count kernel reads those and updates ui once per feature
        return 8 * ng + 8 * w["uniq"] + 8 * nf
    if name == "k_multi_compact":
        # on the lists as accumulated: per-pool word 4, group_off 4 ; scratch features in, CSR out
        return 8 * ng + 4 * L0 + 8 * E0
    if name == "k_list_order":
        # k_list_key (m_off, m_fid in; key 4 + signature 8 out), k_dup_mark (signatures in; head, length out),
        # k_uniq_gather (head, index, offset, signature in; CSR, entry key 4 + value 8, head position out),
        # k_entry_weight (offsets, positions, first value in; entry keys read and written)
        return 36 * L0 + 4 * E0 + 40 * L + 24 * E
    if name == "k_share_reduce":
        # per launch: entry key 4 + value 8, a[] read once, share[] written once
        return it * (12 * E + 16 * nf)
    if name == "k_prop_apply":
        # per iteration U, share, a in; a, share out (k_prop_apply) ; once: k_prop_begin (ui in; U, a, share
        # out) and k_prop_purged (offsets, positions, features in, a[] gathered)
        return it * 40 * nf + 28 * nf + 8 * L + 12 * E
    if name in ("scan", "k_rs_hist", "k_rs_scatter"):
        return w["lib"].get(name) or None
    return None        # k_general_recip (a few thousand lists), k_partial_reduce: not priced


def coverage_block(m, ctx, with_oracle):
    """BASELINE configs[3]: per-base depth over 50 k references x 5 kb from ~50 M alignments resident in HBM
    (msam_coverage.c:33-87): msx_coverage_depths (the batch is the sample; round 3 timed zero-less accumulate + finish,
    kept beside it as `streamed_form_ms`), wall time around the call; priced against the HBM roof on 4 B per depth cell +
    the record bytes read (tid, pos, cigar_off, CIGAR words)."""
    import ctypes as C
    import numpy as np
    refs, tl, ngrp = 50_000, 5_000, 10_000_000
    db = m.DeviceBatch.synth(ctx, SEED, ngrp, refs, 4)
    try:
        off = np.arange(refs + 1, dtype=np.int64) * tl
        total = int(off[-1])
        d_off, d_cov = ctx.alloc(off.nbytes), ctx.alloc(4 * total + 8)
        ctx.to_dev(d_off, off)

        def one(batch, streamed=False):
            if streamed:          # round 2/3's form: the caller's zeroed difference array, marks, in-place prefix sum (what the command line streams batches into)
                ctx.zero(d_cov, 4 * total + 8)
                ctx.sync()
                t0 = time.perf_counter()
                ctx.check(ctx.lib.msx_coverage_accumulate(ctx.h, C.byref(batch.b), C.c_void_p(d_off), refs, total,
                                                          C.c_void_p(d_cov), None))
                ctx.check(ctx.lib.msx_coverage_finish(ctx.h, C.c_void_p(d_cov), total))
                ctx.sync()
                return time.perf_counter() - t0
            ctx.sync()
            t0 = time.perf_counter()              # the whole-sample form: nothing to zero, depths written once
            ctx.check(ctx.lib.msx_coverage_depths(ctx.h, C.byref(batch.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
            return time.perf_counter() - t0
        ts_s = [one(db, streamed=True) for _ in range(4)]
        cov_s = ctx.to_host(d_cov, total, np.int32)
        # the first whole-sample form (a +1 and a -1 mark per run, two 8-bit passes over them): what samples beyond
        # 255 * 2^20 cells take, and the figure the second form is held against
        os.environ["MSX_COV_MARKS"] = "1"
        try:
            ts_m = [one(db) for _ in range(4)]
            cov_m = ctx.to_host(d_cov, total, np.int32)
        finally:
            del os.environ["MSX_COV_MARKS"]
        # the command's form (round 5): the same stream in batches of the command line's size, resident in HBM; every batch's
        # pieces kept on the device (msx_coverage_collect: one emit kernel and one 76-byte read of its overflow flag per
        # batch), sorted and summed once (msx_coverage_collect_finish)
        per = 370_000
        parts = [m.DeviceBatch.synth(ctx, SEED, min(per, ngrp - g0), refs, 4, first_group=g0) for g0 in range(0, ngrp, per)]

        def collected():
            ctx.sync()
            t0 = time.perf_counter()
            for part in parts:
                ctx.check(ctx.lib.msx_coverage_collect(ctx.h, C.byref(part.b), C.c_void_p(d_off), refs, total, C.c_void_p(d_cov), None))
            n_str = C.c_int64(-1)
            ctx.check(ctx.lib.msx_coverage_collect_finish(ctx.h, C.c_void_p(d_cov), total, C.byref(n_str)))
            ctx.sync()
            return time.perf_counter() - t0, int(n_str.value)
        ts_c = [collected() for _ in range(4)]
        cov_c = ctx.to_host(d_cov, total, np.int32)
        for part in parts:
            part.free()
        ts = [one(db) for _ in range(6)]
        best = min(ts[1:])
        cov = ctx.to_host(d_cov, total, np.int32)
        sz = db.sizes
        alg = 4 * total + 12 * db.n_records + 4 * int(sz.n_cigar)
        # size-independent property at full size: the depths sum to the M/=/X bases of the records with a reference
        cig = db.fetch("cigar", int(sz.n_cigar), np.uint32)
        n_cig = np.diff(db.fetch("cigar_off", db.n_records + 1, np.uint32).astype(np.int64))
        has_ref = np.repeat(db.fetch("tid", db.n_records, np.int32) >= 0, n_cig)
        op, w = cig & 0xF, (cig >> 4).astype(np.int64)
        want_sum = int(w[has_ref & ((op == 0) | (op == 7) | (op == 8))].sum())
        blk = {
            "workload": f"c4: coverage of {db.n_records} alignments on {refs} references x {tl} bp "
                        f"({4 * total / 1e9:.2f} GB of int32 depths), inputs resident in HBM",
            "ms": round(best * 1e3, 3), "G_alignments_per_s": round(db.n_records / best / 1e9, 2),
            "algorithmic_bytes": alg, "algorithmic_GBps": round(alg / best / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
            "frac": round(alg / best / 1e9 / HBM_PEAK_GBS, 4),
            "depth_sum": int(cov.astype(np.int64).sum()), "depth_sum_expected": want_sum,
            "depth_sum_ok": bool(int(cov.astype(np.int64).sum()) == want_sum and (cov >= 0).all()),
            "form": "one word per run piece: 4 K-cell tiles, the tile's top byte dropped by the first of two radix passes",
            "streamed_form_ms": round(min(ts_s[1:]) * 1e3, 3), "equal_to_streamed_form": bool(np.array_equal(cov, cov_s)),
            "marks_form_ms": round(min(ts_m[1:]) * 1e3, 3), "equal_to_marks_form": bool(np.array_equal(cov, cov_m)),
            "collected_form": {
                "what": "msx_coverage_collect per batch + msx_coverage_collect_finish: what `msamtools coverage` calls (msh_coverage.c) -- "
                        "the kernels of `ms` with the emit kernel run per batch and the host reading each batch's overflow flag",
                "batches": len(parts), "ms": round(min(t for t, _ in ts_c[1:]) * 1e3, 3), "batches_streamed": ts_c[-1][1],
                "equal": bool(np.array_equal(cov, cov_c))},
        }
        del cov, cov_s, cov_m, cov_c, cig, n_cig, has_ref, op, w
        if with_oracle:
            import oracle_lib as orc
            pg = 200_000                                   # every depth of a prefix of the stream against the oracle
            small = m.DeviceBatch.synth(ctx, SEED, pg, refs, 4)
            hs = m.HostSynth(SEED, pg, refs, 4)
            # (the large-batch path: the binned pile-up is what the c4 batch takes)
            one(small)
            got = ctx.to_host(d_cov, total, np.int32)
            t0 = time.perf_counter()
            want = np.concatenate(orc.coverage(hs, [tl] * refs))
            cpu_s = time.perf_counter() - t0
            blk["parity_prefix"] = {"alignments": hs.n_records, "every_depth_equal": bool(np.array_equal(got, want))}
            blk["cpu_oracle_M_alignments_per_s"] = round(hs.n_records / cpu_s / 1e6, 2)
            blk["parity_ok"] = bool(blk["depth_sum_ok"] and blk["parity_prefix"]["every_depth_equal"] and blk["equal_to_streamed_form"] and
                                     blk["equal_to_marks_form"] and blk["collected_form"]["equal"])
            small.free()
        ctx.free(d_off)
        ctx.free(d_cov)
        return blk
    finally:
        db.free()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)            # (does not return)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.dry_launch:
        print(json.dumps({"rank": rank, "world": world, "local_rank": local_rank,
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "child": os.environ.get("MSX_BENCH_RANK_CHILD") == "1"}), flush=True)
        return

    # stdout carries ONE line, the JSON: whatever the libraries print through C stdio or Python (RCCL's version banner, warnings)
    # goes to stderr -- file descriptor 1 is pointed there for the run, the line is written to the real one at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import msamtools_amd as m

    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist

    ng, nrefs, desc = WORKLOADS[args.workload]
    if args.groups:
        ng = args.groups
    if args.refs:
        nrefs = args.refs

    ctx = m.Context(local_rank)
    # each rank owns the groups [rank*ng, (rank+1)*ng) of one deterministic stream
    db = m.DeviceBatch.synth(ctx, SEED, ng, nrefs, 4, first_group=rank * ng)
    n = db.n_records
    run = m.FilterRun(ctx, db, **FILTER_OPTS)
    prof = m.Profile(ctx, nrefs, "proportional")
    if use_dist:
        # one communicator per rank, owned by the library; every collective of a step is enqueued by
        # the C entry points on the library's own stream
        uid = exchange_comm_id(m, rank, world)
        if uid is None:
            ctx.dist_init_env()
        else:
            ctx.dist_init(uid, rank, world)

    state = {"iters": 0, "n_emit": 0}

    def step():
        prof.reset()
        run.enqueue_with_profile(prof)     # filter | profile: aln_stats_flat, besthit_select with the
                                           # insert accounting inside, scans, emit_order, list compaction
        if not use_dist:
            prof.finalize_enqueue()        # <= 19 proportional iterations, no host round trip
        else:
            # counts all-reduced once, `share` all-reduced inside each of the 19 iterations, purged at the
            # end: RCCL on the library's stream between its kernels, no host synchronisation
            prof.finalize_dist_enqueue()
        st = run.finish()                  # one sync per step; raises on data errors
        state["n_emit"] = int(st.n_emit)

    def barrier():
        ctx.barrier()                      # stream sync, and over RCCL a one-element all-reduce + sync
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ctx.max_over_ranks(elapsed)          # the slowest rank's time
    total_records = ctx.sum_over_ranks(n)
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    try:
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        hbm_used_gib = round((total_b - free_b) / 2**30, 2)      # batch + outputs + every workspace, this rank
    except Exception:
        hbm_used_gib = None
    value = total_records * args.steps / elapsed / 1e6

    ab, pst = prof.fetch()
    out = {
        "metric": "M alignments/s through filter --besthit | profile",
        "value": round(value, 3),
        "unit": "M alignments/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32/f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {desc}; filter -l 80 -p 95 -z 80 --besthit | profile --multi=proportional; "
                        "inputs resident in HBM",
            "alignments_per_gpu": n, "qname_groups_per_gpu": ng, "references": nrefs,
            "alignments_kept_rank0": state["n_emit"],
            "prop_iterations": int(pst.iterations),
            "hbm_used_gib_rank0": hbm_used_gib,
            "parallelism": f"shard{world}" if world > 1 else "single",
        },
    }

    if args.print_checksum:
        import hashlib
        out["checksum"] = {"abundance_sum": float(ab.sum()), "abundance_sha1_6dp": hashlib.sha1(
            np.round(ab, 6).tobytes()).hexdigest(), "inserts": int(pst.insert_count), "uniq": int(pst.uniq_mapper_count),
            "multi": int(pst.multi_mapper_count), "purged": int(pst.purged_insert_count),
            "iterations": int(pst.iterations)}

    # ---- roofline of the dominant kernel (HIP events on the library's stream) ----
    if rank == 0 and not args.no_roofline:
        ctx.timing(True)
        ctx.timing_reset()
        reps = 3
        for _ in range(reps):
            prof.reset()
            run.enqueue_with_profile(prof)
            prof.finalize_enqueue()
            run.finish()
        names = ["k_aln_stats_flat", "k_besthit_select", "k_emit_order", "k_insert_count", "k_multi_compact",
                 "k_list_order", "k_rs_hist", "k_rs_scatter", "k_general_recip", "k_share_reduce", "k_partial_reduce",
                 "k_prop_apply", "scan"]
        tms, lib_bytes = {}, {}
        for k in names:
            ms, cnt = ctx.timing_get(k)
            tms[k] = (ms / reps, cnt / reps)          # ms per step, launches per step
            lib_bytes[k] = ctx.timing_bytes(k) / reps
        ctx.timing(False)
        ab1, st1 = prof.fetch()
        sz = db.sizes
        n_lists0, n_entries0 = prof.multi_size()       # multi-mapped inserts as accumulated
        n_lists, n_entries = prof.shared_size()        # after identical feature sets were merged
        w = dict(n=n, ng=ng, n_cig=int(sz.n_cigar), n_md=int(sz.n_md), n_emit=state["n_emit"],
                 uniq=int(st1.uniq_mapper_count), L=n_lists, E=n_entries, L0=n_lists0, E0=n_entries0, nf=nrefs,
                 iters=int(st1.iterations), lib=lib_bytes)

        def price(k):
            """(bytes per step, GB/s over the kernel's whole time in a step, average launch ms)"""
            ms_step, launches = tms[k]
            b = algorithmic_bytes_per_step(k, w)
            if ms_step <= 0 or launches <= 0:
                return b, None, 0.0
            return b, (b / (ms_step * 1e-3) / 1e9 if b else None), ms_step / launches

        load_traffic(args.workload, ng, nrefs)
        # the dominant kernel: the largest share of the step among the kernels that are priced
        dom = max((k for k in names if algorithmic_bytes_per_step(k, w) and k not in ("scan", "k_rs_hist", "k_rs_scatter")),
                  key=lambda k: tms[k][0])
        dom_bytes, achieved, avg_ms = price(dom)
        per_kernel = {}
        for k, v in tms.items():
            b, g, _ = price(k)
            per_kernel[k] = {"ms_per_step": round(v[0], 4), "launches": v[1],
                             "algorithmic_bytes_per_step": int(b) if b else None,
                             "algorithmic_GBps": round(g, 1) if g else None}
            # a figure above the roof: a small workload served from the 256 MB Infinity Cache, or a wrong formula --
            # flagged, not fatal (a missing JSON line is a failed run to the driver)
            if g is not None and g > HBM_PEAK_GBS:
                per_kernel[k]["over_roof"] = True
                print(f"[bench] warning: {k} prices at {g:.0f} GB/s algorithmic, above the {HBM_PEAK_GBS:.0f} GB/s roof",
                      file=sys.stderr)
        out["roofline"] = {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": TRAFFIC.get(dom),
            "traffic_source": (os.path.relpath(_PMC, ROOT) + " (rocprofv3 --pmc passes of an earlier run of this command; "
                               "not collected inside this run)") if TRAFFIC.get(dom) else None,
            "traffic_stale_sources": TRAFFIC_STALE or None,
            "algorithmic_bytes_per_launch": int(dom_bytes / tms[dom][1]),
            "avg_launch_ms": round(avg_ms, 5), "launches_per_step": tms[dom][1],
            "per_kernel": per_kernel,
            "multi_mapper_lists": n_lists0, "multi_mapper_entries": n_entries0,
            "merged_lists": n_lists, "merged_entries": n_entries,
        }

    # ---- CPU baseline: the oracle (scalar C port of the reference path), 1 thread ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as orc

        class HostCopy:
            """The first sg QNAME groups of the device batch, copied back to the host (the very records the GPU
            was timed on; generating them a second time on the host would take longer than timing them)."""

            def __init__(self, db, sg):
                h = db.to_host()
                goff = h["group_off"]
                nrec = int(goff[sg])
                for k in ("flag", "rflags", "tid", "pos", "nm", "as_"):
                    setattr(self, k, h[k][:nrec])
                self.cigar_off, self.md_off = h["cigar_off"][:nrec + 1], h["md_off"][:nrec + 1]
                self.cigar, self.md = h["cigar"], h["md"]
                self.group_off = goff[:sg + 1]
                self.name_id = np.repeat(np.arange(sg, dtype=np.int32), np.diff(self.group_off.astype(np.int64)))
                self.qname_off = self.qname = None
                self.n_records, self.n_groups = nrec, sg

        sg = ng if args.cpu_sample_groups <= 0 else min(args.cpu_sample_groups, ng)
        hs = HostCopy(db, sg)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            f = orc.run_filter(hs, **FILTER_OPTS)
            p = orc.run_profile(hs, nrefs, multi="proportional", sel=f["emit"])
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)

        # ---- parity of the timed workload itself: the GPU results of this very batch against the oracle
        # run that was just paid for (bit-exact selection / order / counts, <= 1e-6 relative on the profile)
        res = run.result()
        ne = len(f["emit"])
        par = {"checked": "whole batch" if sg == ng else f"first {sg} pools (filter output only)",
               "emit_equal": bool(ne <= res.n_emit and np.array_equal(res.emit[:ne], f["emit"])
                                  and (sg < ng or res.n_emit == ne)),
               "n_emit_gpu": int(res.n_emit), "n_emit_oracle": int(ne)}
        if sg == ng:
            s = p["stats"]
            ui_gpu = prof.ui()
            par["ui_equal"] = bool(np.array_equal(ui_gpu, p["ui"]))
            par["counters_equal"] = bool(
                (int(pst.insert_count), int(pst.uniq_mapper_count), int(pst.multi_mapper_count),
                 int(pst.purged_insert_count)) ==
                (int(s.insert_count), int(s.uniq_mapper_count), int(s.multi_mapper_count), int(s.purged_insert_count)))
            par["iterations_equal"] = bool((int(pst.iterations), int(pst.converged)) == (int(s.iterations), int(s.converged)))
            want = p["abundance"]
            par["zero_pattern_equal"] = bool(np.array_equal(ab == 0, want == 0))
            par["max_rel_err"] = float((np.abs(ab - want) / np.maximum(np.abs(want), 1e-300)).max())
            par["tolerance"] = 1e-6
            par["ok"] = bool(par["emit_equal"] and par["ui_equal"] and par["counters_equal"] and
                             par["iterations_equal"] and par["zero_pattern_equal"] and par["max_rel_err"] <= 1e-6)
        else:
            par["ok"] = par["emit_equal"]
        out["parity"] = par
        del res
        if sg == ng and not args.no_e2e and args.e2e_groups == ng:
            # the end-to-end BAM is this very stream (same generator, seed, groups and references): what the oracle
            # just computed is what the command line must write
            import digest as dg
            counts = np.diff(hs.group_off.astype(np.int64))
            gidx = np.repeat(np.arange(counts.size, dtype=np.int64), counts)
            em = f["emit"]
            h, cnt = dg.stream_digest(hs.flag[em], hs.tid[em], hs.pos[em], dg.fnv_sim_names(gidx[em]))
            del gidx
            e2e_expect = {"n_emit": cnt, "emit_digest": h, "emit": em,
                          "pipe": {"stats": p["stats"], "abundance": p["abundance"]},
                          "plain": orc.run_profile(hs, nrefs, multi="proportional")}
            if args.e2e_seq_groups and args.e2e_seq_groups < ng:
                goff_s = int(hs.group_off[args.e2e_seq_groups])
                e2e_seq_expect = {"n_emit": int(np.searchsorted(em, goff_s))}
                e2e_seq_expect["emit"] = em[:e2e_seq_expect["n_emit"]]       # (pools are independent: the prefix's emit list is the list's prefix)
            elif args.e2e_seq_groups == ng:
                e2e_seq_expect = e2e_expect               # the same stream with SEQ and QUAL: same records kept, same profile
            if args.e2e_sam_groups and args.e2e_sam_groups <= ng:
                # the SAM-text leg reads a prefix of the same stream: filter's records are the emit list's prefix (count and digest)
                n_s = int(np.searchsorted(em, int(hs.group_off[args.e2e_sam_groups])))
                counts_s = np.diff(hs.group_off[:args.e2e_sam_groups + 1].astype(np.int64))
                gidx_s = np.repeat(np.arange(counts_s.size, dtype=np.int64), counts_s)
                ems = em[:n_s]
                h_s, cnt_s = dg.stream_digest(hs.flag[ems], hs.tid[ems], hs.pos[ems], dg.fnv_sim_names(gidx_s[ems]))
                e2e_sam_expect = {"n_emit": cnt_s, "emit_digest": h_s}
        out["cpu_baseline"] = {
            "value": round(hs.n_records / best / 1e6, 3), "unit": "M alignments/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model(), "host_cpus_online": os.cpu_count(), "host_cpus_granted": granted_cpus(),
            "sample": f"first {sg} QNAME groups ({hs.n_records} alignments) of the same synthetic stream, "
                      f"{nrefs} references, copied back from the device batch, resident in RAM; best of 2 runs of oracle filter+profile "
                      f"({best:.2f} s each).  COMPUTE ONLY: pre-parsed SoA arrays in, no BGZF inflate, no record walk, no output written "
                      f"-- comparable with `value` (kernel-only), not with the command line; the end-to-end CPU figure with its "
                      f"decode / compute / encode split is e2e.cpu_baseline_e2e",
        }

        # the same oracle with every host core: filter (the per-record walk + best hit, the bulk of the
        # time) on pool-aligned shards in parallel (ctypes releases the GIL), then one profile pass over
        # the concatenated selection -- the "best CPU" figure
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncores = min(granted_cpus(), 64)         # what the cgroup / affinity mask grants, not what is online
            goff = hs.group_off.astype(np.int64)
            cuts = [int(goff[int(sg * i / ncores)]) for i in range(ncores + 1)]
            shards = [orc.make_records_slice(hs, hs.name_id, cuts[i], cuts[i + 1]) for i in range(ncores)
                      if cuts[i + 1] > cuts[i]]

            starts = [cuts[i] for i in range(ncores) if cuts[i + 1] > cuts[i]]

            def one(sh):
                return orc.run_filter(sh, name_id=sh.name_id, **FILTER_OPTS)["emit"]

            bestn = None
            with ThreadPoolExecutor(max_workers=len(shards)) as ex:
                for _ in range(2):
                    t0 = time.perf_counter()
                    parts = list(ex.map(one, shards))
                    sel = np.concatenate([e.astype(np.int64) + st for e, st in zip(parts, starts)]).astype(np.int32)
                    orc.run_profile(hs, nrefs, multi="proportional", sel=sel)
                    dt = time.perf_counter() - t0
                    bestn = dt if bestn is None else min(bestn, dt)
            out["cpu_baseline_all_cores"] = {
                "value": round(hs.n_records / bestn / 1e6, 2), "unit": "M alignments/s", "cores": len(shards),
                "kind": "port", "sample": f"same sample, compute only; ONLY filter runs on {len(shards)} pool-aligned shards in parallel "
                                          f"threads -- the profile (insert counting + the proportional iterations) is one serial pass "
                                          f"behind them ({bestn:.2f} s in all)"}
        except Exception as exc:      # never let the extra figure break the bench line
            out["cpu_baseline_all_cores"] = {"error": str(exc)[:200]}

    # ---- N ranks: the distributed result against ONE context doing all N shards (rank 0, after the timed region) ----
    if use_dist and not args.no_dist_parity:
        par = None
        if rank == 0:
            try:
                prof1 = m.Profile(ctx, nrefs, "proportional")
                recs = 0
                for r in range(world):
                    db_r = db if r == 0 else m.DeviceBatch.synth(ctx, SEED, ng, nrefs, 4, first_group=r * ng)
                    run_r = run if r == 0 else m.FilterRun(ctx, db_r, **FILTER_OPTS)
                    run_r.enqueue_with_profile(prof1)
                    run_r.finish()
                    recs += db_r.n_records
                    if r:
                        run_r.free()
                        db_r.free()
                prof1.finalize_enqueue()              # no collective: everything is on this device
                ab1, st1 = prof1.fetch()
                same = lambda a, b: all(int(getattr(a, k)) == int(getattr(b, k)) for k in
                                        ("insert_count", "uniq_mapper_count", "multi_mapper_count", "purged_insert_count",
                                         "iterations", "converged"))
                rel = float((np.abs(ab - ab1) / np.maximum(np.abs(ab1), 1e-300)).max())
                par = {"checked": f"{world} ranks x {n} alignments against one context accumulating all {world} shards",
                       "records_equal": bool(recs == total_records), "counters_and_iterations_equal": bool(same(pst, st1)),
                       "zero_pattern_equal": bool(np.array_equal(ab == 0, ab1 == 0)), "max_rel_diff": rel,
                       "tolerance": 1e-9, "ok": bool(recs == total_records and same(pst, st1) and rel <= 1e-9)}
                prof1.close()
            except Exception as exc:
                par = {"error": str(exc)[:300]}
            out["dist_parity"] = par
        barrier()                                      # the other ranks keep their communicator until rank 0 is done

    # ---- the multi-GPU step on the one GPU there is: msx_profile_finalize_dist_enqueue over a one-rank RCCL
    # communicator -- the code path the ranks of --gpus N run, collectives included -- next to ms_per_step
    if rank == 0 and world == 1 and not use_dist and not args.no_dist_leg and not args.no_roofline:
        try:
            ctx.dist_init(m.dist_unique_id(), 0, 1)

            def dstep():
                prof.reset()
                run.enqueue_with_profile(prof)
                prof.finalize_dist_enqueue()
                run.finish()
            def timed(poll):
                os.environ["MSX_DIST_POLL"] = str(poll)
                for _ in range(2):
                    dstep()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    dstep()
                barrier()
                dt = 1e3 * (time.perf_counter() - t0) / max(args.steps, 1)
                a_, s_ = prof.fetch()
                return dt, a_, s_
            dms0, ab_0, st_0 = timed(0)           # all 19 iterations enqueued, nothing waits for the host (rounds 2-3)
            dms, ab_d, st_d = timed(8)            # the default: the convergence flag looked at every 8th iteration
            os.environ["MSX_DIST_SLICES"] = "2"   # the local half in two slices, slice 0's all-reduce on a side stream
            dms2, ab_2, st_2 = timed(8)
            os.environ.pop("MSX_DIST_SLICES", None)
            os.environ.pop("MSX_DIST_POLL", None)
            out["dist_one_rank_ms_per_step"] = round(dms, 4)
            out["dist_one_rank"] = {
                "ms_per_step": round(dms, 4), "ratio_to_plain_step": round(dms / ms_per_step, 4),
                "iterations": int(st_d.iterations),
                "max_rel_diff_to_plain": float((np.abs(ab_d - ab) / np.maximum(np.abs(ab), 1e-300)).max()),
                "without_convergence_poll": {"ms_per_step": round(dms0, 4), "iterations": int(st_0.iterations),
                                             "max_rel_diff_to_plain": float((np.abs(ab_0 - ab) / np.maximum(np.abs(ab), 1e-300)).max())},
                "slices_2": {"ms_per_step": round(dms2, 4), "iterations": int(st_2.iterations),
                             "abundances_bit_equal_to_one_slice": bool(np.array_equal(ab_2, ab_d))},
                "note": "one-rank RCCL communicator: every collective of the N-rank step is enqueued and runs; MSX_DIST_POLL=8 (default) "
                        "stops enqueueing all-reduces once the convergence flag is seen, =0 enqueues all 19"}
        except Exception as exc:
            out["dist_one_rank"] = {"error": str(exc)[:300]}

    # ---- N ranks: the same K steps with MSX_DIST_SLICES=2 (the local half in two slices of the feature range, slice 0's
    # all-reduce under slice 1's kernels) beside the default, which `value` is -- the A/B a node with more than one GPU decides.
    # The last thing the run does, and under a watchdog: the form has only ever run on a one-rank communicator -- should it hang
    # on a real one, rank 0 still prints the line (everything else is in it by now) and every rank leaves.
    if use_dist and not args.no_dist_parity:
        import threading

        def give_up():
            if rank == 0:
                out["dist_slices_2"] = {"error": "no answer within 120 s: abandoned"}
                out["headline"] = headline(out)
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os._exit(0)
        dog = threading.Timer(120.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            os.environ["MSX_DIST_SLICES"] = "2"
            for _ in range(max(1, min(args.warmup, 2))):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el2 = ctx.max_over_ranks(time.perf_counter() - t0)
            ab2, pst2 = prof.fetch()
            out["dist_slices_2"] = {"ms_per_step": round(1e3 * el2 / max(args.steps, 1), 4),
                                    "value": round(total_records * args.steps / el2 / 1e6, 3),
                                    "ratio_to_default": round(el2 / elapsed, 4),
                                    "abundances_bit_equal_to_default": bool(np.array_equal(ab2, ab)),
                                    "iterations": int(pst2.iterations),
                                    "note": "MSX_DIST_SLICES=2; default (1) is what `value` reports"}
        except Exception as exc:
            out["dist_slices_2"] = {"error": str(exc)[:300]}
        finally:
            os.environ.pop("MSX_DIST_SLICES", None)
            dog.cancel()

    prof.close()
    run.free()
    db.free()

    # ---- BASELINE configs[3]: per-base coverage, 50 k references x 5 kb, ~50 M alignments ----
    if rank == 0 and world == 1 and not args.no_coverage and args.workload == "c3":
        try:
            out["coverage"] = coverage_block(m, ctx, not args.no_cpu_baseline)
        except Exception as exc:
            out["coverage"] = {"error": str(exc)[:300]}
    ctx.close()
    if rank == 0 and world == 1 and not args.no_e2e and not args.no_cpu_baseline:
        e2e_refs = nrefs if args.e2e_groups == ng else 100_000
        def probe(path, skip_header=True):          # (a context of its own, after the command lines have had the device to themselves)
            c = m.Context(0)
            try:
                return inflate_probe(m, c, path, skip_header=skip_header)
            finally:
                c.close()
        out["e2e"] = e2e_cli(args.e2e_groups, e2e_refs, locals().get("e2e_expect"), probe=probe, marginal_groups=args.e2e_groups // 4,
                             cpu_records=2_000_000)
        if args.e2e_seq_groups:
            out["e2e_seq"] = e2e_cli(args.e2e_seq_groups, e2e_refs, locals().get("e2e_seq_expect"), seq=True, probe=probe,
                                     marginal_groups=args.e2e_seq_groups // 4)
        if args.e2e_sam_groups:
            out["e2e_sam"] = e2e_sam(args.e2e_sam_groups, e2e_refs, locals().get("e2e_sam_expect"))
        if isinstance(out.get("coverage"), dict) and "error" not in out["coverage"]:
            out["coverage"]["cli"] = coverage_cli()
    if rank == 0:
        out["headline"] = headline(out)         # (LAST key, <= 600 bytes: a record that keeps only the line's tail keeps these)
        # RCCL prints a version banner through C stdio; flush it first so the JSON line is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
