"""world_size-2 gloo test of the multi-GPU arithmetic.

Each rank owns a pool-aligned shard of one synthetic stream, runs filter +
insert counting on its shard alone, and the ranks exchange exactly what
msx_profile_finalize_dist_enqueue (msx_dist.hip) exchanges over RCCL:
all-reduce(sum) of the per-reference counts and counters, then per
proportional-sharing iteration all-reduce(sum) of the `share` vector (sum over
the rank's multi-mappers of 1/S), then the purged count.  The result must
equal the single-process oracle on the whole stream (counts exact, abundances
<= 1e-6 relative).  The per-shard compute here is numpy/oracle (no GPU in this
container); the collective pattern and the sharding rule are the ones under test.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

G, REFS, SEED = 6000, 300, 24680
OPTS = dict(l=80, p=95, z=80, besthit=True)


def shard_counts(hs, emit, n_refs):
    """ui (2 per unique insert) and the multi-mapper lists of one shard, from filter's output stream."""
    gid = hs.name_id[emit]
    tid = hs.tid[emit]
    ui = np.zeros(n_refs, np.int64)
    lists = []
    uniq = multi = 0
    bounds = np.flatnonzero(np.r_[True, gid[1:] != gid[:-1], True])
    for s, e in zip(bounds[:-1], bounds[1:]):
        fids = list(dict.fromkeys(tid[s:e].tolist()))      # first-appearance order
        if len(fids) == 1:
            ui[fids[0]] += 2
            uniq += 1
        else:
            lists.append(np.asarray(fids))
            multi += 1
    return ui, lists, len(bounds) - 1, uniq, multi


def worker(rank, world, port, out, slices):
    import torch
    import torch.distributed as dist
    import msamtools_amd as m
    import oracle_lib as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hs = m.HostSynth(SEED, G, REFS, 4, first_group=rank * G)       # the rank's shard: groups [rank*G, (rank+1)*G)
    emit = orc.run_filter(hs, **OPTS)["emit"]
    ui, lists, inserts, uniq, multi = shard_counts(hs, emit, REFS)
    t_ui = torch.from_numpy(ui)
    t_cnt = torch.tensor([inserts, uniq, multi], dtype=torch.int64)
    dist.all_reduce(t_ui)
    dist.all_reduce(t_cnt)
    U = t_ui.numpy().astype(np.float64) / 2
    a = U.copy()
    flat = np.concatenate(lists) if lists else np.zeros(0, np.int64)
    owner = np.repeat(np.arange(len(lists)), [len(x) for x in lists]) if lists else np.zeros(0, np.int64)
    k, converged = 0, False
    while k < 19:
        S = np.bincount(owner, weights=a[flat], minlength=len(lists))
        recip = np.where(S > 0, 1.0 / np.where(S > 0, S, 1.0), 0.0)
        share = torch.from_numpy(np.bincount(flat, weights=recip[owner], minlength=REFS))
        if slices == 1:
            dist.all_reduce(share)                                   # the per-iteration collective
        else:
            # MSX_DIST_SLICES: the same vector in equal parts of the feature range -- the SAME cuts on every rank (the library
            # computes slice i's part while slice i - 1's all-reduce travels: msx_prop.hip prop_slice_cuts)
            for i in range(slices):
                lo, hi = REFS * i // slices, REFS * (i + 1) // slices
                part = share[lo:hi].clone()
                dist.all_reduce(part)
                share[lo:hi] = part
        new = U + a * share.numpy()
        new[new < 1e-20] = 0
        delta = float(((new - a) ** 2).sum() / REFS)
        a = new
        k += 1
        if delta < 1e-10:
            converged = True
            break
    S = np.bincount(owner, weights=a[flat], minlength=len(lists))
    purged = torch.tensor([int((S == 0).sum())])
    dist.all_reduce(purged)
    if rank == 0:
        np.savez(out, a=a, ui=t_ui.numpy(), cnt=t_cnt.numpy(), purged=purged.numpy(), k=k, conv=converged)
    dist.destroy_process_group()


@pytest.mark.parametrize("slices", [1, 2, 3])
def test_two_rank_sharded_profile_equals_single_process(tmp_path, slices):
    import torch.multiprocessing as mp
    import msamtools_amd as m
    import oracle_lib as orc
    out = str(tmp_path / "rank0.npz")
    port = 29500 + (os.getpid() % 2000) + 7 * slices
    mp.spawn(worker, args=(2, port, out, slices), nprocs=2, join=True)
    got = np.load(out)
    whole = m.HostSynth(SEED, 2 * G, REFS, 4)
    sel = orc.run_filter(whole, **OPTS)["emit"]
    ref = orc.run_profile(whole, REFS, multi="proportional", sel=sel)
    s = ref["stats"]
    assert (got["ui"] == ref["ui"].astype(np.int64)).all()
    assert got["cnt"].tolist() == [s.insert_count, s.uniq_mapper_count, s.multi_mapper_count]
    assert int(got["purged"][0]) == s.purged_insert_count
    assert int(got["k"]) == s.iterations and bool(got["conv"]) == bool(s.converged)
    want = ref["abundance"]
    assert ((got["a"] == 0) == (want == 0)).all()
    rel = np.abs(got["a"] - want) / np.maximum(np.abs(want), 1e-300)
    assert rel.max() <= 1e-6


def test_shards_are_prefix_stable():
    """Rank r's shard equals groups [r*G, (r+1)*G) of the unsharded stream (no pool is split)."""
    import msamtools_amd as m
    whole = m.HostSynth(SEED, 2 * G, REFS, 4)
    for r in range(2):
        part = m.HostSynth(SEED, G, REFS, 4, first_group=r * G)
        s, e = int(whole.group_off[r * G]), int(whole.group_off[(r + 1) * G])
        assert part.n_records == e - s
        assert (part.flag == whole.flag[s:e]).all() and (part.tid == whole.tid[s:e]).all()
        assert (part.group_off.astype(np.int64) + s == whole.group_off[r * G:(r + 1) * G + 1]).all()


def _rendezvous_worker(rank, world, port, q):
    import ctypes as C
    from msamtools_amd import _lib
    lib = _lib.load()
    buf = (C.c_uint8 * 128)()
    if rank == 0:
        for i in range(128):
            buf[i] = (7 * i + 3) & 0xFF
    rc = lib.msx_dist_rendezvous(b"127.0.0.1", port, rank, world, buf, 128, 30)
    q.put((rank, rc, bytes(buf)))


def test_c_rendezvous_hands_the_communicator_id_to_every_rank():
    """msx_dist_rendezvous (the TCP hand-over msx_dist_init_env uses for the 128-byte RCCL id), three
    processes, through the C ABI -- no GPU involved."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    world = 3
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, port, q)) for r in (2, 1, 0)]   # rank 0 last
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
    want = bytes((7 * i + 3) & 0xFF for i in range(128))
    assert [(r, rc) for r, rc, _ in got] == [(0, 0), (1, 0), (2, 0)]
    assert all(b == want for _, _, b in got)


def test_dist_entry_points_without_a_context():
    """Rank / world of "no communicator" are 0 / 1; initialisation needs a context (hence a gfx950 device)."""
    from msamtools_amd import _lib
    lib = _lib.load()
    assert lib.msx_dist_world(None) == 1 and lib.msx_dist_rank(None) == 0
    assert lib.msx_dist_init_env(None) != 0


def _stray_client(port, stop):
    """connects to the rendezvous port, says nothing (a port scanner) or garbage, again and again"""
    import socket
    import time
    k = 0
    while not stop.is_set():
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1) as sk:
                if k % 2:
                    sk.sendall(b"GET / HTTP/1.0\r\n\r\n")
                time.sleep(0.3)
        except OSError:
            time.sleep(0.05)
        k += 1


def test_c_rendezvous_survives_stray_connections():
    """one stray connection used to block rank 0 in recv for ever, a bad hello ended the hand-over.
    Now a connection has 5 s to say a rank's hello and is skipped otherwise."""
    import multiprocessing as mp
    import threading
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() % 2000)
    stop = threading.Event()
    th = threading.Thread(target=_stray_client, args=(port, stop), daemon=True)
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, 2, port, q)) for r in (0, 1)]
    procs[0].start()
    th.start()
    import time
    time.sleep(1.0)                  # the stray client is talking to rank 0 when rank 1 arrives
    procs[1].start()
    got = sorted(q.get(timeout=90) for _ in range(2))
    stop.set()
    for p in procs:
        p.join(timeout=30)
    want = bytes((7 * i + 3) & 0xFF for i in range(128))
    assert [(r, rc) for r, rc, _ in got] == [(0, 0), (1, 0)]
    assert all(b == want for _, _, b in got)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launcher_starts_n_ranks(world):
    """`python bench.py --gpus N` with no launcher around it starts N rank processes itself, each with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's line on stdout and fails if a rank fails
    (with WORLD_SIZE unset `--gpus 8` ran ONE rank).  --dry-launch: the ranks only report their
    environment, so this runs without a GPU; the parent never initialises one."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    d0 = json.loads(r.stdout.decode().strip().split("\n")[-1])
    assert (d0["rank"], d0["world"], d0["local_rank"], d0["child"]) == (0, world, 0, True)
    others = [json.loads(l) for l in r.stderr.decode().split("\n") if l.startswith("{")]
    assert sorted(d["rank"] for d in others) == list(range(1, world))
    assert all(d["world"] == world and d["local_rank"] == d["rank"] and d["master"] == d0["master"] for d in others)
    # under a launcher (WORLD_SIZE set) it is a rank, not a launcher
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-launch"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env2, timeout=120)
    d = json.loads(r.stdout.decode().strip().split("\n")[-1])
    assert (d["rank"], d["world"], d["child"]) == (1, 4, False)


def test_bench_launcher_reports_a_failing_rank(tmp_path):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # (no GPU here: every rank fails at context creation; where there is one, an unknown workload does it)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--groups", "-5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0
