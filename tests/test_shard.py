"""CPU, world_size 2, gloo: the N>1 path.  Batch shards are contiguous and balanced, no
data-path collective, per-sample losses all-gathered in batch order.  The per-shard compute is
a stand-in (the CPU oracle) because the HIP ops need a GPU; what is under test is the sharding
and the collectives, which are identical under RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_shard_bounds_are_contiguous_and_balanced():
    from rfnet_amd.shard import shard_bounds
    for B in (0, 1, 7, 32, 256, 257):
        for R in (1, 2, 3, 4, 8):
            spans = [shard_bounds(B, r, R) for r in range(R)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(R - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from oracle.oracle import Oracle
    from rfnet_amd import shard
    r, w, _ = shard.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    orc = Oracle()
    rng = np.random.RandomState(100)  # same inputs on every rank
    a = torch.from_numpy(rng.randn(B, 64, 3).astype(np.float32))
    c = torch.from_numpy(rng.randn(B, 90, 3).astype(np.float32))

    def chamfer_cpu_standin(x, y):  # per-sample chamfer_big on this rank's shard
        d1, _, d2, _ = orc.nn_distance(x.numpy(), y.numpy())
        return torch.from_numpy((np.sqrt(d1).mean(1) + np.sqrt(d2).mean(1)) / 2)

    lo, hi = shard.shard_bounds(B, rank, world)
    xs, ys = shard.shard_batch([a, c], rank, world)
    assert xs.shape[0] == hi - lo and torch.equal(xs, a[lo:hi])
    full = shard.sharded_per_sample(chamfer_cpu_standin, [a, c])
    assert full.shape == (B,)
    # optional input distribution from rank 0
    src = a if rank == 0 else None
    mine = shard.scatter_from_rank0(src, rank, world)
    assert torch.equal(mine, a[lo:hi])
    np.save(os.path.join(out_dir, f"loss_{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_world_size_2_gloo(tmp_path, B):
    from oracle.oracle import Oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    l0 = np.load(tmp_path / "loss_0.npy")
    l1 = np.load(tmp_path / "loss_1.npy")
    assert np.array_equal(l0, l1)  # every rank ends with the same gathered vector
    rng = np.random.RandomState(100)
    a = rng.randn(B, 64, 3).astype(np.float32)
    c = rng.randn(B, 90, 3).astype(np.float32)
    d1, _, d2, _ = Oracle().nn_distance(a, c)
    assert np.array_equal(l0, ((np.sqrt(d1).mean(1) + np.sqrt(d2).mean(1)) / 2).astype(np.float32))


def test_bench_multi_rank_plumbing_dry_run():
    """bench.py under torch.distributed.run with 2 ranks (gloo, CPU): the launch line the driver
    uses for N>1, in the script's dry-run mode (placeholder work, no operator, no metric).  Checks
    that exactly one JSON line comes out of rank 0 and that both ranks rendezvous and exit cleanly."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2
