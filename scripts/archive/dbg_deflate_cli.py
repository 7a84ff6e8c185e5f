#!/usr/bin/env python3
"""debug: `filter -b` output block by block against zlib, the -bu output (same blocks, stored) and the host twin"""
import os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msamtools_amd as m
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seq = "--seq" if len(sys.argv) > 2 and sys.argv[2] == "seq" else ""
exe, dev = os.path.join(ROOT, "msamtools_amd/bin/msamtools"), os.path.join(ROOT, "msamtools_amd/bin/msamtools-dev")
T = "/tmp/dbgdef"; os.makedirs(T, exist_ok=True)
subprocess.check_call(f"{dev} synth --groups {groups} --refs 1000000 {seq} -b > {T}/in.bam", shell=True)
F = "filter -l 80 -p 95 -z 80 --besthit"
subprocess.check_call(f"{exe} {F} -b {T}/in.bam > {T}/b.bam", shell=True)
subprocess.check_call(f"{exe} {F} -bu {T}/in.bam > {T}/u.bam", shell=True)
B = m.bgzf_split(open(f"{T}/b.bam", "rb").read())
U = m.bgzf_split(open(f"{T}/u.bam", "rb").read())
print("blocks", len(B), len(U))
subprocess.check_call(["gcc", "-O2", "-o", f"{T}/twin", os.path.join(ROOT, "tests/c/deflate_twin.c"), "-lz"])
bad = 0
for i, (pl, isize, crc) in enumerate(B):
    ok = True
    try:
        d = zlib.decompress(pl, -15)
        ok = len(d) == isize and zlib.crc32(d) == crc
    except Exception as e:
        ok = False
    if not ok:
        bad += 1
        if bad > 3:
            continue
        raw = zlib.decompress(U[i][0], -15) if i < len(U) and U[i][1] == isize else None
        print("block", i, "isize", isize, "comp", len(pl), "btype", (pl[0] >> 1) & 3, "u-block matches" , raw is not None)
        if raw is not None:
            open(f"{T}/raw.bin", "wb").write(raw)
            out = os.path.join(ROOT, "gpurun_out", "dbgdef"); os.makedirs(out, exist_ok=True)
            open(os.path.join(out, f"raw_{i}.bin"), "wb").write(raw)
            open(os.path.join(out, f"dev_{i}.bin"), "wb").write(pl)
            if subprocess.call([f"{T}/twin", f"{T}/raw.bin", f"{T}/tw.out"], stdout=subprocess.DEVNULL) != 0:
                print("  the twin fails on this block too")
                continue
            tw = m.bgzf_split(open(f"{T}/tw.out", "rb").read())[0][0]
            n = min(len(tw), len(pl))
            first = next((k for k in range(n) if tw[k] != pl[k]), n)
            print("  twin", len(tw), "device", len(pl), "first difference at byte", first)
            out = os.path.join(ROOT, "gpurun_out", "dbgdef"); os.makedirs(out, exist_ok=True)
            open(os.path.join(out, f"raw_{i}.bin"), "wb").write(raw)
            open(os.path.join(out, f"dev_{i}.bin"), "wb").write(pl)
            # the device encoder alone on the same payload
            ctx = m.Context(0)
            st, nb = m.bgzf_deflate(ctx, raw, 6)
            alone = m.bgzf_split(st)[0][0]
            print("  device alone equals twin:", alone == tw, "equals cli block:", alone == pl)
            ctx.close()
print("bad blocks", bad, "of", len(B))
