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


def _dp_worker(rank, world, port, B, out_dir):
    """Data-parallel training-step plumbing: per-rank backward on the batch shard,
    bucketed gradient all-reduce, and the autograd-carrying loss gather."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from rfnet_amd import shard
    shard.init_from_env(backend="gloo")
    torch.manual_seed(1234 + rank)  # ranks start from DIFFERENT weights: broadcast must fix that
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    shard.broadcast_parameters(list(net.parameters()), src=0)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 32, 3, generator=g)
    y = torch.randn(B, 32, 3, generator=g)
    xs, ys = shard.shard_batch([x, y], rank, world)

    def per_sample(a, c):
        return ((net(a) - c) ** 2).sum(-1).mean(-1)  # (b_local,)

    # (1) loss on the GATHERED vector, gradient shares summed
    full = shard.all_gather_per_sample(per_sample(xs, ys), B, rank, world)
    assert full.shape == (B,) and full.requires_grad
    net.zero_grad()
    full.mean().backward()
    n1 = shard.allreduce_gradients(list(net.parameters()), average=False, bucket_bytes=128)
    g_gather = [p.grad.clone() for p in net.parameters()]
    # (2) the usual recipe: mean of the local losses, averaged gradients (exact for equal shards)
    net.zero_grad()
    per_sample(xs, ys).mean().backward()
    n2 = shard.allreduce_gradients(list(net.parameters()), average=True)
    g_local = [p.grad.clone() for p in net.parameters()]
    torch.save({"w": [p.detach().clone() for p in net.parameters()], "g_gather": g_gather,
                "g_local": g_local, "full": full.detach(), "n1": n1, "n2": n2},
               os.path.join(out_dir, f"dp_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_dp_gradient_allreduce_gloo(tmp_path, B):
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "dp_0.pt")
    r1 = torch.load(tmp_path / "dp_1.pt")
    # same weights (broadcast), same gathered losses, same reduced gradients on both ranks
    for a, c in zip(r0["w"], r1["w"]):
        assert torch.equal(a, c)
    assert torch.equal(r0["full"], r1["full"])
    for key in ("g_gather", "g_local"):
        for a, c in zip(r0[key], r1[key]):
            assert torch.equal(a, c)
    assert r0["n1"] > 1 and r0["n2"] == 1  # 128-byte buckets -> several collectives; default -> one
    # single-process reference of the same step
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    with torch.no_grad():
        for p, w in zip(net.parameters(), r0["w"]):
            p.copy_(w)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 32, 3, generator=g)
    y = torch.randn(B, 32, 3, generator=g)
    per = ((net(x) - y) ** 2).sum(-1).mean(-1)
    assert torch.allclose(per.detach(), r0["full"], rtol=1e-6, atol=1e-7)
    per.mean().backward()
    for p, got in zip(net.parameters(), r0["g_gather"]):
        assert torch.allclose(p.grad, got, rtol=1e-5, atol=1e-6)  # any B: shares sum to the whole
    if B % 2 == 0:
        for p, got in zip(net.parameters(), r0["g_local"]):
            assert torch.allclose(p.grad, got, rtol=1e-5, atol=1e-6)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun: the script must start the two ranks itself (a child
    torch.distributed.run job), relay exactly one JSON line from rank 0 and report the process
    group's world size -- not silently run one rank (VERDICT r1 #1 / ADVICE r1)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--dry-run-cpu"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks"] == 2 and d["gathered"] == 4


def test_bench_refuses_a_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0 and "--gpus 2" in (out.stderr + out.stdout)
