"""GPU: the C ABI called directly through ctypes (no Python wrappers): status codes, workspace
contract, empty shapes, stream argument -- the behaviour a non-Python host binds to."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def test_status_codes_and_workspace_contract(orc):
    from rfnet_amd._lib import lib
    dev = "cuda"
    b, n, m = 2, 100, 300
    rng = np.random.RandomState(0)
    a = torch.from_numpy(rng.randn(b, n, 3).astype(np.float32)).to(dev)
    c = torch.from_numpy(rng.randn(b, m, 3).astype(np.float32)).to(dev)
    d1 = torch.empty(b, n, device=dev); i1 = torch.empty(b, n, dtype=torch.int32, device=dev)
    d2 = torch.empty(b, m, device=dev); i2 = torch.empty(b, m, dtype=torch.int32, device=dev)
    need = lib.rf_nn_distance_workspace_bytes(b, n, m)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream()
    args = (b, n, m, p(a), p(c), p(d1), p(i1), p(d2), p(i2))
    assert lib.rf_nn_distance(*args, p(ws), need - 1, stream.cuda_stream) == -2  # RF_EWORKSPACE
    assert lib.rf_nn_distance(-1, n, m, *args[3:], p(ws), need, stream.cuda_stream) == -1  # RF_EINVAL
    assert lib.rf_nn_distance(b, n, m, None, p(c), p(d1), p(i1), p(d2), p(i2), p(ws), need,
                              stream.cuda_stream) == -1
    assert lib.rf_nn_distance(b, 0, m, *args[3:], p(ws), need, stream.cuda_stream) == -1
    assert lib.rf_nn_distance(0, n, m, *args[3:], p(ws), need, stream.cuda_stream) == 0  # empty batch
    # a real call on a non-default stream; the library never synchronises
    assert lib.rf_nn_distance(*args, p(ws), need, stream.cuda_stream) == 0
    stream.synchronize()
    e = orc.nn_distance(a.cpu().numpy(), c.cpu().numpy())
    assert np.array_equal(d1.cpu().numpy(), e[0]) and np.array_equal(i2.cpu().numpy(), e[3])
    # bad attributes of the other entry points
    assert lib.rf_queryballpoint(b, n, m, C.c_float(0.1), 0, p(a), p(c), p(i1), p(i1), None) == -1
    assert lib.rf_farthestpointsampling(b, 0, 4, p(a), None, p(i1), None) == -1
    assert lib.rf_farthestpointsampling(b, n, 0, p(a), None, p(i1), None) == 0
    # the sampling entry point with caller scratch of a stated size: 0 bytes where the unsorted kernels run, the sorted set where
    # the sorted-cloud kernel does (6000 points, 300 samples); too little is RF_EWORKSPACE; results equal the plain entry's
    assert lib.rf_farthestpointsampling_workspace_bytes(b, n, 40) == 0
    big = torch.from_numpy(rng.rand(2, 6000, 3).astype(np.float32)).to(dev)
    fneed = lib.rf_farthestpointsampling_workspace_bytes(2, 6000, 300)
    assert fneed > 0 and lib.rf_farthestpointsampling_workspace_bytes(2, 6000, 100) == 0
    fws = torch.empty(fneed, dtype=torch.uint8, device=dev)
    o1 = torch.empty(2, 300, dtype=torch.int32, device=dev); o2 = torch.empty_like(o1)
    assert lib.rf_farthestpointsampling_ws(2, 6000, 300, p(big), p(fws), fneed - 1, p(o1), None) == -2
    assert lib.rf_farthestpointsampling_ws(2, 6000, 300, p(big), p(fws), fneed, p(o1), None) == 0
    assert lib.rf_farthestpointsampling(2, 6000, 300, p(big), None, p(o2), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    # (6000 points run 8 per lane: the kernel holds 8192 samples; asked for more it refuses before it writes anything)
    assert lib.rf_farthestpointsampling_sorted(2, 6000, 9000, 0, p(big), p(fws), fneed, p(o1), None, None) == -1
    lv = (C.c_float * 65)()
    assert lib.rf_approxmatch_levels(b, n, m, p(a), p(c), p(d1), lv, 65, p(ws), need, None) == -1
    assert lib.rf_status_string(-1) == b"invalid argument"


def test_empty_sets_and_degenerate_sizes(orc):
    from rfnet_amd import _raw as R
    dev = "cuda"
    z = torch.zeros(0, 5, 3, device=dev)
    out = R.nn_distance(z, torch.zeros(0, 7, 3, device=dev))
    assert [tuple(t.shape) for t in out] == [(0, 5), (0, 5), (0, 7), (0, 7)]
    pts = torch.rand(2, 50, 3, device=dev)
    idx = torch.zeros(2, 0, dtype=torch.int32, device=dev)
    assert tuple(R.gather_point(pts, idx).shape) == (2, 0, 3)
    g = R.gather_point_grad(pts, idx, torch.zeros(2, 0, 3, device=dev))
    assert tuple(g.shape) == (2, 50, 3) and float(g.abs().sum()) == 0.0
    gi = torch.zeros(2, 4, 0, dtype=torch.int32, device=dev)
    assert tuple(R.group_point(pts, gi).shape) == (2, 4, 0, 3)
    # empty dataset: every ball empty, pts_cnt = 0, idx untouched (wrapper zero-fills)
    qi, cnt = R.query_ball_point(0.5, 3, torch.zeros(2, 0, 3, device=dev), pts[:, :4])
    assert int(cnt.sum()) == 0 and int(qi.abs().sum()) == 0
    # three_nn against an empty known set: dist = +inf, idx = 0 (reference: 1e40 cast, tf_interpolate.cpp:66)
    d, i = R.three_nn(pts, torch.zeros(2, 0, 3, device=dev))
    assert torch.isinf(d).all() and int(i.abs().sum()) == 0
    # EMD with a single point on each side: all the mass goes to the one pair
    a = torch.rand(3, 1, 3, device=dev)
    c = torch.rand(3, 1, 3, device=dev)
    mt = R.approx_match(a, c)
    assert np.allclose(mt.cpu().numpy(), orc.approx_match(a.cpu().numpy(), c.cpu().numpy()), rtol=1e-5)
    cost = R.match_cost(a, c, mt)
    assert np.allclose(cost.cpu().numpy(), (mt[:, 0, 0] * (a - c).norm(dim=-1)[:, 0]).cpu().numpy(), rtol=1e-5)


def test_calls_are_independent_of_workspace_contents(orc):
    """The workspace carries no state between calls: poison it and results do not change."""
    from rfnet_amd import _host, _raw as R
    rng = np.random.RandomState(3)
    a = torch.from_numpy(rng.randn(2, 700, 3).astype(np.float32)).cuda()
    c = torch.from_numpy(rng.randn(2, 1300, 3).astype(np.float32)).cuda()
    first = [t.clone() for t in R.nn_distance(a, c)]
    for buf in _host._ws_cache.values():
        buf.fill_(0xFF)
    second = R.nn_distance(a, c)
    for x, y in zip(first, second):
        assert torch.equal(x, y)
    u = torch.from_numpy((rng.rand(2, 300, 3) - .5).astype(np.float32)).cuda()
    v = torch.from_numpy((rng.rand(2, 200, 3) - .5).astype(np.float32)).cuda()
    m1 = R.approx_match(u, v).clone()
    for buf in _host._ws_cache.values():
        buf.fill_(0xFF)
    assert torch.equal(m1, R.approx_match(u, v))
    # the EMD state vectors are not cleared as a whole (am_init writes the padded entries only): ragged sizes on every route --
    # plain sweeps, the fused op with gradients, and a batch large enough for sorted rows at the sharp levels
    e1 = [t.clone() for t in R.earth_mover(u, v, with_grad=True)]
    for buf in _host._ws_cache.values():
        buf.fill_(0xFF)
    e2 = R.earth_mover(u, v, with_grad=True)
    assert torch.equal(e1[0], e2[0]) and torch.allclose(e1[1], e2[1], rtol=1e-5, atol=1e-7) and torch.allclose(e1[2], e2[2], rtol=1e-5, atol=1e-7)
    ub = torch.from_numpy((rng.rand(22, 2000, 3) - .5).astype(np.float32)).cuda()
    vb = torch.from_numpy((rng.rand(22, 1900, 3) - .5).astype(np.float32)).cuda()
    mb = R.approx_match(ub, vb).clone()
    cb = R.earth_mover(ub, vb).clone()
    for buf in _host._ws_cache.values():
        buf.fill_(0xFF)
    assert torch.equal(mb, R.approx_match(ub, vb))
    for buf in _host._ws_cache.values():
        buf.fill_(0xFF)
    assert torch.equal(cb, R.earth_mover(ub, vb))


def test_two_streams_do_not_share_scratch(orc):
    """The same op on two streams at once: each stream gets its own cached workspace."""
    from rfnet_amd import _raw as R
    rng = np.random.RandomState(5)
    a1 = torch.from_numpy(rng.randn(4, 3000, 3).astype(np.float32)).cuda()
    c1 = torch.from_numpy(rng.randn(4, 5000, 3).astype(np.float32)).cuda()
    a2 = torch.from_numpy(rng.randn(4, 3000, 3).astype(np.float32)).cuda()
    c2 = torch.from_numpy(rng.randn(4, 5000, 3).astype(np.float32)).cuda()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(5):  # interleave launches on the two streams
        with torch.cuda.stream(s1):
            o1 = R.nn_distance(a1, c1)
        with torch.cuda.stream(s2):
            o2 = R.nn_distance(a2, c2)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    e1, e2 = orc.nn_distance(a1.cpu().numpy(), c1.cpu().numpy()), orc.nn_distance(a2.cpu().numpy(), c2.cpu().numpy())
    for o1, o2 in outs:
        assert all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(o1, e1))
        assert all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(o2, e2))


def test_bench_rccl_path_on_one_gpu():
    """bench.py with a real RCCL process group of one rank (RF_FORCE_PG=1): communicator bound to the
    device, barrier fences, max-over-ranks all-reduce -- the code path every rank of an N-GPU run takes --
    and exactly ONE line on stdout although RCCL prints a version banner there."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RF_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["value"] > 1e12
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] <= 1.0
    assert d["roofline"]["identical_to_dense_sweep"] is True


def test_bench_c5_through_rccl_on_one_gpu():
    """BASELINE.json configs[4] on one rank's share THROUGH RCCL (RF_FORCE_PG=1: a real nccl process group of one
    rank): communicator bound to the device, barrier fences, the all-gather of per-sample losses, the cross-rank
    loss check -- so that the day a driver runs `--gpus 8 --workload c5` the path has been through RCCL on this
    box.  (The 8-way split itself is covered on CPU/gloo: tests/test_shard.py.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RF_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "c5", "--steps", "3",
                          "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["unit"] == "samples/s" and d["scaling"] == "weak"
    c5 = d["c5"]
    assert c5["gathered_losses_shape"] == [32, 3] and c5["losses_equal_across_ranks"] is True and c5["finite"] is True
    assert c5["mode"].startswith("hip graph") or c5["mode"].startswith("eager"), c5["mode"]
    assert c5["value"] > 1000  # samples/s per GPU (round 2: ~5000)


def test_scatter_from_rank0_takes_a_cpu_source_under_rccl():
    """rank 0 usually holds the batch on the HOST; RCCL moves device buffers only, so the source is staged to the
    receive device first (round-2 advisor finding).  One-rank nccl group on this GPU."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from rfnet_amd import shard
os.environ.update(RF_FORCE_PG="1")
rank, world, local = shard.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl"
full = torch.arange(4 * 5 * 3, dtype=torch.float32).reshape(4, 5, 3)  # on the CPU
mine = shard.scatter_from_rank0(full, rank, world)
assert mine.is_cuda and torch.equal(mine.cpu(), full)
odd = torch.arange(3 * 2, dtype=torch.float32).reshape(3, 2)
assert torch.equal(shard.scatter_from_rank0(odd, rank, world).cpu(), odd)
dist.destroy_process_group()
print("ok")
""" % root
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(RF_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    # (RCCL prints its version banner on stdout too)
    assert out.returncode == 0 and "ok" in out.stdout.split(), out.stdout + out.stderr[-2000:]
