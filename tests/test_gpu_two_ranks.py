"""Two ranks on two GPUs (skipped on a one-GPU box): `bench.py --gpus 2` starts its ranks itself, every rank runs
filter | profile on its own shard, counts and -- inside each of the 19 iterations -- `share` go through RCCL
(msx_profile_finalize_dist_enqueue), and rank 0 repeats both shards on one context without any collective
(msx_profile_finalize_enqueue): same counters, same iteration count, abundances equal to 1e-9 (the
hand-declared RCCL ABI and the 21-collective schedule had only ever run with one rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_gpus():
    try:
        import torch
        return torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    except Exception:
        return 0


@pytest.mark.skipif(n_gpus() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_n_ranks_equal_one_context(ranks):
    if n_gpus() < ranks:
        pytest.skip(f"needs {ranks} GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--workload", "tiny",
                        "--groups", "400000", "--refs", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-roofline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().split("\n")[-1])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak"
    assert d["dist_parity"]["ok"] is True, d["dist_parity"]
