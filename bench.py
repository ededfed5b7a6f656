#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:
    "point-pairs/sec Chamfer (B x N x M) at 1/2/4/8 GPU", configs[1] = Chamfer fwd+bwd, B=32,
    2048 vs 16384; "EMD iters/sec" as an extra field.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * N>1: when WORLD_SIZE is already set (the driver's `python -m torch.distributed.run ...
    bench.py --gpus N`), this process IS one rank.  When it is not, the script starts the N ranks
    ITSELF: the parent -- before any torch.cuda / HIP call -- runs `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD
    process, relays rank 0's JSON line and exits with the child's return code (no exec of a
    process that touched the GPU).  `n_gpus` in the line is the process group's world size.
  * one rank per GPU over RCCL; the batch shards across ranks with NO data-path collective
    ("scaling": "weak": every GPU runs the full B=32 workload on its own samples); only the
    barrier, the max-over-ranks timing and (C5) the all-gather of per-sample losses use RCCL.
  * a "step" = one pass of the hot path over one batch: nn_distance forward (both directions)
    + nn_distance_grad with upstream grads of ones (the reference bench's reduce_sum loss,
    tf_ops/CD/tf_nndistance.py:50), ONE C-ABI call (rf_chamfer_step) on buffers allocated once,
    inputs resident in HBM before the timed region.
  * rank 0 prints ONE JSON line.  `value` = B*N*M*K*world / seconds (pairs/s, whole job).
  * "roofline": the forward is fp32-VALU bound: NOT HBM (20*B*(N+M) bytes) and
    NOT MFMA ((b-a)^2 is not a product of a row and a column factor).  The dominant kernel is
    nnp_sweep, the culled exact sweep (nn_pruned.hip: sort-tile-recursive sort + box-bound culling,
    bit-identical outputs).  `achieved`/`frac` are what the VALU EXECUTED: 8 flop x the directed
    pairs the kernel evaluated (its own counters) / its average launch duration (hipEvents on
    the launch stream during the timed steps) against the 157.3 TFLOP/s fp32 vector peak -- a
    fraction <= 1 by construction.  The culling itself is reported separately as
    `algorithmic_speedup_vs_dense` (dense sweep time / culled sweep time on the same inputs) and
    `algorithmic_achieved` (16 flop x B*N*M / time, may exceed the peak: that is the culling, not
    the pipe).  "roofline_dense" is the dense sweep (nn_sweep, every pair) in the same run.
  * "by_distribution": the same step on the point distributions the operator meets in the model
    (randn, uniform cube, sphere surface, resample_pcd-style duplicates, the untrained RFNet's own
    outputs): ms/step, evaluated fraction, culled vs dense forward, `identical_to_dense_sweep`.
  * "cpu_baseline": the reference's own CPU kernels (nnsearch x2 + the NnDistanceGrad loop,
    oracle/_ref, kind "reference"; the C restatement, kind "port", where that is absent) on one
    host core on the same workload (the reference op is single-threaded); "cpu_baseline_all_cores":
    the same bodies, one batch element per OpenMP thread, on every host core.
    Rank 0, N=1 only.
  * "c3": BASELINE.json configs[2] (FPS 16384 -> 1024 + gather_point + query_ball_point(0.1, 32) +
    group_point, B=32, uniform seed 100) with per-op rooflines; "per_op_roofline": the HBM-bound
    operators of the path (Chamfer backward, match_cost, match_cost_grad) against the 8 TB/s roof;
    "emd.roofline": the EMD half of the metric against its transcendental-VALU issue floor.
  * `--workload c5`: BASELINE.json configs[4] on this rank's share (B=32 per GPU): RFNet recurrent
    forward (3 steps to 16384 points) + chamfer_big + earth_mover at 64^2 / 1024^2, per-sample
    losses all-gathered over the ranks; reported in samples/s.  The default run carries the same
    measurement as the extra field "c5" (few steps).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# before the HIP runtime starts: captured hipMemsetAsync nodes (torch's reductions use them) replay
# garbage under ROCm 7's graph packet capture -- rfnet_amd.enable_graph_safe_runtime() has the story (it sets
# exactly this; spelled out here because it must precede `import torch` + any HIP call); no replay-time cost
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak
# VALU instructions per (row, column) pair of every launch of approx_match's reference schedule (10 levels, the last with
# multiplier 0), counted in rfnet_amd/csrc/approxmatch.hip and confirmed by SQ_INSTS_VALU (profiles/r04_rocprofv3_summary.txt):
# (launches, plain VALU, v_exp_f32[, packed VALU]) per pair -- d2 is 6, an exponential term is mul + exp (+ mul by the row ratio) + fma
EMD_LAUNCH_MIX = {
    # round 5: the sweeps of the three sharp levels take their rows in spatial order and drop, after the distance (6) and the
    # test (~1), the columns whose weights are exactly 0 for the whole wave; a wave keeps 0.088 / 0.185 / 0.418 of its columns at
    # levels -4^7 / -4^6 / -4^5 on C4's uniform clouds (profiles/r05_emd_skip_model.txt, tools/experiments/emd_skip_model.py: a
    # model of this workload, NOT counted in this run) -- so these rows are EXECUTED instruction counts, not 8 + 1 per pair
    "am_p1 (level 0, skipping)": (1, 7 + 0.088 * 2, 0.088),
    "am_p2, level 0 (skipping)": (1, 7 + 0.088 * 2, 0.088),
    "am_p2, level 1 (skipping)": (1, 7 + 0.185 * 2, 0.185),
    # round 5, late: the DENSE sweeps run their lane's two rows as the halves of packed fp32 operations (v_pk_add / v_pk_mul /
    # v_pk_fma: 4th field = packed instructions per pair, two pairs per instruction): P2 8 packed per two pairs, the fused
    # P3 + P1 11, P3 alone 9; from level 2 (-4^5: a wave keeps 42 % of its columns) on the sweeps are dense -- packed, they beat
    # the skipping form there and the form with only its P3 part conditional
    # round 6: from the third level on the sweeps run over the LIVE columns / rows of set 2 only -- remainR is exactly +0 for
    # 44 / 68 / 80 / 87 / 92 / 97 / 99 / 99.7 % of C4's columns after levels 1 .. 8 (the same kind of constant as the keep fractions
    # above: measured on this workload by tools/experiments/emd_live_fractions.py, NOT counted in this run).  The fused P3(v) + P1(v+1)
    # sweeps of level pairs 1+2 .. 8+9 visit 0.91 + 0.56 + 0.32 + 0.20 + 0.135 + 0.08 + 0.033 + 0.008 = 2.25 sweeps' worth of pairs
    # (packed: 5.5 instructions per pair, two exponentials), P2 of levels 2 .. 9 0.56 + ... + 0.003 = 1.34 (am_p2_live_kernel: 4 packed per
    # pair, one exponential, and the wave reduction of a row's terms -- 8 instructions per 8 pairs)
    "am_p3p1 over the live columns, level pairs 1+2 .. 8+9 (packed)": (2.25, 0, 2, 5.5),
    "am_p2 over the live rows, levels 2 .. 9 (packed)": (1.34, 1.0, 1, 4.0),
    "am_p3p1, levels 0+1 (skipping, P3 under its own test)": (1, 7 + 0.185 * 2 + 0.088 * 3, 0.185 + 0.088),
    # round 5: levels 1, 3, 5, 7 take their weight from the next level's by two squarings (2 mul instead of mul + exp)
    # ... and the sharpest level is evaluated only where some column of the wave is within its cut-off of the row (14 % of the
    # (wave, row) pairs at C4: 1 - (1 - 0.0023)^64)
    "am_match (10 levels in one pass)": (1, 36 + 1 + 0.14 * 3, 4 + 0.14),
}


def emd_issue_floor_ms(eb, n, m):
    """Issue floor of approx_match's kernels: counted instructions x issue costs measured by tools/ubench/valu_rate.hip
    (profiles/issue_costs.json: constants from a committed measurement, NOT measured in this run), over the chip's
    1024 SIMDs.  additive: plain x vop2 + exp x v_exp; mix: x the measured non-additivity of the 9-instruction column."""
    try:
        with open(os.path.join(ROOT, "profiles", "issue_costs.json")) as f:
            ic = json.load(f)
        c = ic["cycles_per_wave_instruction_per_simd"]
        mix = [(v + (0.0,))[:4] for v in EMD_LAUNCH_MIX.values()]
        cyc = sum(k * (p * c["vop2_f32"] + e * c["v_exp_f32"] + q * c["v_pk_f32"]) for k, p, e, q in mix)
        nonadd = 9.0 * c["emd_column_mix_9_instr"] / (8.0 * c["vop2_f32"] + c["v_exp_f32"])
        add_ms = cyc * (eb * n * m / 64.0) / 1024.0 / (ic["clock_ghz"] * 1e9) * 1e3
    except (OSError, ValueError, KeyError, TypeError, ZeroDivisionError):  # a missing or reshaped constants file costs the roof, not the bench
        return None
    return {"additive_ms": add_ms, "mix_ms": add_ms * nonadd, "non_additivity": nonadd,
            "valu_per_pair": sum(k * (p + e + q) for k, p, e, q in mix),
            "exp_per_pair": sum(k * e for k, p, e, q in mix),
            "packed_per_pair": sum(k * q for k, p, e, q in mix),
            "constants": "profiles/issue_costs.json (tools/ubench/valu_rate.hip, profiles/r04_valu_rate.txt): not measured in this run"}

HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E spec peak
# transcendental issue roof: v_exp_f32 issues once per 8.14 cycles per SIMD (profiles/issue_costs.json), 1024 SIMDs x 64 lanes, 2.36 GHz
TRANS_ROOF_EXP_PER_S = 1024 * 64 * 2.36e9 / 8.14


# ----------------------------------------------------------------------------- launching -----
def spawn_ranks(args):
    """--gpus N>1 without a launcher: start the N ranks as a child torch.distributed.run job.
    This parent never initialises the GPU (nothing below touches torch.cuda)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    relayed = 0
    for line in proc.stdout:
        if line.startswith("{"):
            sys.stdout.write(line)  # rank 0's single JSON line
            sys.stdout.flush()
            relayed += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and relayed != 1:
        sys.stderr.write(f"bench.py: expected exactly one JSON line from rank 0, saw {relayed}\n")
        rc = 1
    sys.exit(rc)


def dry_run_cpu(args, emit):
    """Multi-rank plumbing check without a GPU (tests/test_shard.py).  Not a measurement, and no
    operator runs: a placeholder tensor op stands between the fences."""
    import torch.distributed as dist

    from rfnet_amd import shard
    rank, world, _ = shard.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"bench.py: process group has {world} ranks but --gpus {args.gpus}")
    g = torch.Generator().manual_seed(100 + rank)
    a = torch.randn(2, 64, 3, generator=g)
    c = torch.randn(2, 128, 3, generator=g)

    def fence():
        if world > 1:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    for _ in range(2):
        per = torch.cdist(a, c).min(-1).values.mean(-1)  # placeholder work
    fence()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    full = shard.all_gather_per_sample(per, 2 * world, rank, world)  # the C5 loss gather, on gloo
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({"dry_run": True, "n_gpus": dist.get_world_size() if world > 1 else 1,
              "ranks": world, "backend": "gloo", "steps": 2, "gathered": int(full.shape[0]),
              "seconds_max": float(tmax.item()),
              "note": "CPU/gloo plumbing check; no operator runs, no metric"})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- CPU baseline --
def cpu_baseline(B, N, M, seed, sample_b, with_grad=True):
    """Reference CPU path timed on this host: NnDistanceOp = nnsearch x2 (tf_nndistance.cpp:79-80)
    and, with_grad, the NnDistanceGrad CPU loop (:126-163) -- the same forward+backward the GPU
    step runs -- single thread, on `sample_b` batch elements of the same workload."""
    from oracle.oracle import Oracle, Ref, ref_available
    rng = np.random.RandomState(seed)
    sample_b = max(1, min(B, sample_b))
    a = rng.randn(sample_b, N, 3).astype(np.float32)
    c = rng.randn(sample_b, M, 3).astype(np.float32)
    if ref_available():
        impl, kind = Ref(), "reference"
    else:
        impl, kind = Oracle(), "port"
    impl.nn_distance(a[:1, :256], c[:1, :256])  # warm
    t0 = time.perf_counter()
    d1, i1, d2, i2 = impl.nn_distance(a, c)
    if with_grad:
        impl.nn_distance_grad(a, c, np.ones_like(d1), i1, np.ones_like(d2), i2)
    dt = time.perf_counter() - t0
    return {
        "value": sample_b * N * M / dt,
        "unit": "pairs/s",
        "cores": 1,
        "kind": kind,
        "sample": f"nn_distance forward (both directions){' + nn_distance_grad' if with_grad else ''}, "
                  f"{sample_b} of {B} batch elements of {N}x{M}, {dt:.2f} s on 1 core",
    }


def cpu_baseline_all_cores(B, N, M, seed, with_grad=True):
    """"additionally an OpenMP-over-batch run on all host cores, with core count printed" --
    the SAME reference bodies (oracle/_ref: nnsearch x2 + the NnDistanceGrad loop), one batch element per
    OpenMP thread (the wrapper loop is oracle/build_ref.sh's; the bodies are the reference's, untouched)."""
    from oracle.oracle import Ref, ref_available
    if not ref_available():
        return None
    ref = Ref()
    rng = np.random.RandomState(seed)
    a = rng.randn(B, N, 3).astype(np.float32)
    c = rng.randn(B, M, 3).astype(np.float32)
    g1, g2 = np.ones((B, N), np.float32), np.ones((B, M), np.float32)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    host_cores = cores
    cores = min(cores, B)  # one batch element per thread: more threads than batch elements would idle
    ref.nn_step_all_cores(a[:cores, :256], c[:cores, :256], threads=cores)  # warm (thread pool)
    t0 = time.perf_counter()
    _, used = ref.nn_step_all_cores(a, c, g1 if with_grad else None, g2 if with_grad else None, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": B * N * M / dt, "unit": "pairs/s", "cores": int(used), "kind": "reference",
            "sample": f"nn_distance forward (both directions){' + nn_distance_grad' if with_grad else ''}, all {B} batch "
                      f"elements of {N}x{M}, one batch element per OpenMP thread: {used} threads busy on a host of "
                      f"{host_cores} cores, {dt:.2f} s", "host_cores": int(host_cores)}


def emd_cpu_baseline(eb, en, seed, sample_b, gpu_cost):
    """The reference's own CPU EMD path beside the GPU one: approxmatch_cpu + matchcost_cpu
    (pc_distance/tf_approxmatch.cpp:23-105, compiled from where they lie into oracle/_ref) on `sample_b` samples of the
    C4 workload, one core (the reference op is single-threaded: one Compute, plain loops).  The CPU path runs ITS schedule
    (11 levels, j = 7..-2 then 0, double accumulators, expf -- SURVEY T4), the GPU the CUDA op's 10 levels: the two costs of the same
    sample differ by that extra level (~1 %), which `gpu_over_cpu_cost` shows; parity proper is tests/test_gpu_emd.py."""
    from oracle.oracle import Ref, ref_available
    if not ref_available():
        return None
    ref = Ref()
    rng = np.random.RandomState(seed)
    a = (rng.random_sample((eb, en, 3)) - 0.5).astype(np.float32)[:sample_b]
    c = (rng.random_sample((eb, en, 3)) - 0.5).astype(np.float32)[:sample_b]
    ref.matchcost_cpu(a[:1, :64], c[:1, :64], ref.approxmatch_cpu(a[:1, :64], c[:1, :64]))  # warm
    t0 = time.perf_counter()
    cost = ref.matchcost_cpu(a, c, ref.approxmatch_cpu(a, c))
    dt = time.perf_counter() - t0
    return {"value": (sample_b / float(eb)) / dt, "unit": "calls/s", "cores": 1, "kind": "reference",
            "sample": f"approxmatch_cpu + matchcost_cpu on {sample_b} of the {eb} samples of one {en} vs {en} call, {dt:.2f} s on 1 core; "
                      f"value = 1 / ({dt:.2f} s x {eb}/{sample_b}): B={eb} calls per second",
            "seconds_per_sample": dt / sample_b,
            "cpu_cost": [float(x) for x in cost], "gpu_cost": [float(x) for x in gpu_cost[:sample_b]],
            "gpu_over_cpu_cost": [float(g) / float(x) for g, x in zip(gpu_cost[:sample_b], cost)],
            "cost_note": "same samples, two schedules: the CPU op sweeps 11 levels in double with expf, the CUDA op (which the GPU path "
                         "restates) 10 levels in fp32 (SURVEY T4)"}


# ----------------------------------------------------------------------------- workloads -----
def make_distributions(B, N, M, rank, dev, want_model=True):
    """(name, xyz1 (B,N,3), xyz2 (B,M,3), note) for the distributions the operator meets."""
    rng = np.random.RandomState(300 + rank)

    def t(x):
        return torch.from_numpy(np.ascontiguousarray(x.astype(np.float32))).to(dev)

    yield ("uniform_cube", t(rng.rand(B, N, 3) - 0.5), t(rng.rand(B, M, 3) - 0.5), "U(-0.5,0.5)^3")
    s1, s2 = rng.randn(B, N, 3), rng.randn(B, M, 3)
    yield ("sphere_surface", t(s1 / np.linalg.norm(s1, axis=-1, keepdims=True)),
           t(s2 / np.linalg.norm(s2, axis=-1, keepdims=True)), "unit sphere surface")
    # data_util.resample_pcd (data_util.py:8-13): a partial scan with fewer points than the input
    # size is filled up with random DUPLICATES of its own points -> exact ties
    uniq = max(N // 3, 1)
    base = rng.rand(B, uniq, 3) - 0.5
    idx = np.concatenate([np.stack([rng.permutation(uniq) for _ in range(B)]),
                          rng.randint(0, uniq, (B, N - uniq))], 1)
    yield ("resample_pcd_duplicates", t(np.take_along_axis(base, idx[..., None], 1)), t(rng.rand(B, M, 3) - 0.5),
           f"xyz1 = {uniq} unique points resampled to {N} (data_util.resample_pcd), xyz2 unique")
    if want_model and M == 16384:
        from rfnet_amd.rfnet import RFNet
        torch.manual_seed(0)
        net = RFNet().to(dev)
        partial = t(rng.rand(B, 3000, 3) - 0.5)
        with torch.no_grad():
            out = net(partial)[3]
        yield ("rfnet_untrained_output", partial[:, :N].contiguous() if N <= 3000 else t(rng.rand(B, N, 3) - 0.5),
               out.contiguous(), "xyz1 = the network input (first N points), xyz2 = untrained RFNet's final output")
        del net


def timed(fn, reps, fence):
    """(wall ms per call between fences, {kernel: ms per call} by the library's hipEvents in a second pass)"""
    from rfnet_amd import _lib
    for _ in range(2):
        fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    fence()
    ms = (time.perf_counter() - t0) / reps * 1e3
    _lib.profile_collect()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    pr = _lib.profile_collect()
    return ms, {k: v[0] / reps for k, v in pr.items()}  # per CALL (a call may launch a kernel several times)


def hbm_roof(bytes_per_call, ms, kernel, note=None):
    gbps = bytes_per_call / (ms * 1e-3) / 1e9
    d = {"bound": "hbm", "kernel": kernel, "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": gbps / HBM_PEAK_GBPS, "bytes_per_call": float(bytes_per_call), "avg_launch_ms": ms}
    if note:
        d["note"] = note
    return d


def run_c3(rank, dev, fence, reps):
    """BASELINE.json configs[2] / SURVEY.md 8(d) C3: xyz = U[0,1)^3 seed 100, (32, 16384, 3);
    idx = farthest_point_sample(1024, xyz); new_xyz = gather_point(xyz, idx);
    query_ball_point(0.1, 32, xyz, new_xyz); group_point(xyz, idx).  Which roof binds each op: SURVEY.md 8(d)
    (FPS: serial-reduction latency, reported as us per iteration and distance updates/s; ball query: VALU scan
    with early exit, reported as pair tests/s against the scan's upper bound; gather / group: HBM)."""
    from rfnet_amd._raw import farthest_point_sample, gather_point, group_point, nn_sort, query_ball_point, sample_and_group, three_nn
    B, n, m, ns, r = 32, 16384, 1024, 32, 0.1
    rng = np.random.RandomState(100 + rank)
    xyz = torch.from_numpy(rng.random_sample((B, n, 3)).astype(np.float32)).to(dev)
    idx = farthest_point_sample(m, xyz)
    new_xyz = gather_point(xyz, idx)
    gi, cnt = query_ball_point(r, ns, xyz, new_xyz)                      # auto: the boxed kernel at this size
    gi_scan, cnt_scan = query_ball_point(r, ns, xyz, new_xyz, form="scan")
    fps_ms, fps_k = timed(lambda: farthest_point_sample(m, xyz), max(3, reps // 2), fence)
    ga_ms, ga_k = timed(lambda: gather_point(xyz, idx), reps, fence)
    qb_ms, qb_k = timed(lambda: query_ball_point(r, ns, xyz, new_xyz), reps, fence)
    handle = nn_sort(xyz)
    qh_ms, qh_k = timed(lambda: query_ball_point(r, ns, xyz, new_xyz, sorted1=handle.buf), reps, fence)
    qs_ms, qs_k = timed(lambda: query_ball_point(r, ns, xyz, new_xyz, form="scan"), reps, fence)
    gp_ms, gp_k = timed(lambda: group_point(xyz, gi), reps, fence)
    aux = torch.cuda.Stream(device=dev)
    one_ms, one_k = timed(lambda: sample_and_group(m, r, ns, xyz), max(3, reps // 2), fence)
    onea_ms, _ = timed(lambda: sample_and_group(m, r, ns, xyz, aux_stream=aux), max(3, reps // 2), fence)
    one = sample_and_group(m, r, ns, xyz, aux_stream=aux)
    torch.cuda.synchronize()
    one_same = (bool(torch.equal(one[0], idx)) and bool(torch.equal(one[1], new_xyz)) and bool(torch.equal(one[2], gi))
                and bool(torch.equal(one[3], cnt)) and bool(torch.equal(one[4], group_point(xyz, gi))))
    fps_kms = fps_k.get("fps_sorted", fps_k.get("fps_reg", fps_k.get("fps_mem", fps_ms)))  # (the sampling kernel alone: its sort is in fps_k)
    from rfnet_amd._raw import farthest_point_sample_reg
    fpr_ms, fpr_k = timed(lambda: farthest_point_sample_reg(m, xyz), max(3, reps // 2), fence)
    fps_same = bool(torch.equal(farthest_point_sample_reg(m, xyz), idx))
    box_kms = qb_k.get("query_ball_boxes", qb_ms)
    updates = float(B) * n * (m - 1)
    # the way back (feature propagation): the three nearest SAMPLED points of every point of the cloud
    tn_d, tn_i = three_nn(xyz, new_xyz)                                  # auto: over sorted copies of both sets at this size
    tn_ds, tn_is = three_nn(xyz, new_xyz, form="scan")
    tn_ms, tn_k = timed(lambda: three_nn(xyz, new_xyz), reps, fence)
    h_new = nn_sort(new_xyz)
    tnh_ms, tnh_k = timed(lambda: three_nn(xyz, new_xyz, sorted1=handle.buf, sorted2=h_new.buf), reps, fence)
    tns_ms, tns_k = timed(lambda: three_nn(xyz, new_xyz, form="scan"), reps, fence)
    tn_kms = tn_k.get("three_nn_boxes", tn_ms)
    return {
        "workload": f"B={B} per GPU, farthest_point_sample {n} -> {m} + gather_point + query_ball_point(r={r}, nsample={ns}) "
                    "+ group_point(c=3), U[0,1)^3 seed 100 (BASELINE.json configs[2])",
        # `ms_per_pass` = the four reference ops in sequence, as in rounds 1-4 (round 5 put the one-call figure under this key;
        # it now has its own: ms_per_pass_one_call)
        "ms_per_pass": fps_ms + ga_ms + qb_ms + gp_ms,
        "ms_per_pass_one_call": onea_ms,
        "ms_per_pass_one_call_is": "rf_sample_and_group: ONE C-ABI call (one sort of the dataset serves FPS -- which runs over the sorted cloud and "
                          "writes new_xyz -- and the boxed ball query, which writes grouped_xyz); outputs identical to the four ops: "
                          + str(one_same),
        "ms_per_pass_one_call_one_stream": one_ms,
        "ms_per_pass_four_ops_scan_ball_query": fps_ms + ga_ms + qs_ms + gp_ms,
        "one_call_kernels_ms": one_k,
        "farthest_point_sample": {
            "ms": fps_ms, "kernels_ms": fps_k, "kernel_ms": fps_kms, "us_per_iteration": fps_kms * 1e3 / (m - 1),
            "form": "over the spatially sorted cloud (nnp_sort + fps_sorted_kernel: 16 waves x 16 consecutive sorted points per lane, a wave "
                    "re-scans only when the new sample can still lower a running minimum in one of its lanes' boxes -- 2.9 of 16 waves per "
                    "iteration at C3); indices identical to the unsorted kernel's: " + str(fps_same),
            "unsorted_kernel": {"ms": fpr_ms, "kernel_ms": fpr_k.get("fps_reg", fpr_ms),
                                "us_per_iteration": fpr_k.get("fps_reg", fpr_ms) * 1e3 / (m - 1)},
            "distance_updates_per_s": updates / (fps_kms * 1e-3),
            "roofline": {"bound": "latency", "what": "serial chain of npoint-1 dependent arg-max reductions, one workgroup "
                         "(one CU) per cloud: B of the 256 CUs busy by construction.  The unsorted kernel's VALU issues 53 % of every wave's "
                         "life (343 instructions per wave and iteration); over the sorted cloud 82 % of the wave-scans are skipped and what "
                         "is left of an iteration is its chain: box test, one lone wave's scan, wave reduction, barrier, slot reduction, "
                         "scalar re-read of the winner.  `distance_updates_per_s` counts the reference's (npoint-1)*n updates (an effective "
                         "figure: most are proven unnecessary and not executed).  Spreading a cloud over 2/4/8 workgroups "
                         "was built and is 1.27-1.43x SLOWER: one all-to-all exchange of the winners costs 0.5-0.9 us against the ~1 us "
                         "of a whole iteration (profiles/r05_fps_cluster.txt; the build: tools/experiments/fps_cluster.patch.txt)",
                         "cus_busy": B, "valu_flops_per_s": 8.0 * updates / (fps_kms * 1e-3),
                         "frac_of_fp32_peak_on_busy_cus": 8.0 * updates / (fps_kms * 1e-3) / 1e12 / (FP32_PEAK_TFLOPS * B / 256.0)}},
        "gather_point": {"ms": ga_ms, "roofline": hbm_roof(28.0 * B * m, ga_k.get("gather_point", ga_ms), "gather_point",
                                                            "28*B*m bytes (SURVEY 8(d)): 0.9 MB -- launch-bound, not a bandwidth test; "
                                                            "inside rf_sample_and_group FPS writes new_xyz itself")},
        "query_ball_point": {
            "ms": qb_ms, "kernels_ms": qb_k, "ms_on_a_sorted_handle": qh_ms, "ms_scan_kernel": qs_ms,
            "form": "boxed: the dataset in sort-tile-recursive order (nnp_sort), one wave per query over the 64-record blocks whose "
                    "box is within the radius, the nsample lowest original indices through an LDS bitmap (grouping.hip)",
            "identical_to_scan_kernel": bool(torch.equal(gi, gi_scan)) and bool(torch.equal(cnt, cnt_scan)),
            "mean_pts_cnt": float(cnt.float().mean().item()),
            "pair_tests_upper_bound_per_s": float(B) * n * m / (box_kms * 1e-3),
            "roofline": {"bound": "valu-issue",
                         "what": "the boxed kernel tests ~4 % of the B*n*m pairs (256 boxes + ~10 blocks of 64 records per query) and is "
                                 "bound by latency and instruction issue: 452 VALU + 303 SALU wave-instructions per query wave, waves parked 47 % "
                                 "and waiting to issue 24 % of their life (profiles/r05_rocprofv3_summary.txt (C); HBM-side 71 MB per "
                                 "launch for a 6.3 MB dataset); `achieved` prices the B*n*m pair tests of "
                                 "the SCAN it replaces (8 flop each) against the fp32 vector peak -- an effective figure, like the "
                                 "culled Chamfer's",
                         "achieved": 8.0 * B * n * m / (box_kms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": 8.0 * B * n * m / (box_kms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         "scan_kernel_frac": 8.0 * B * n * m / (qs_k.get("query_ball_point", qs_ms) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}},
        "three_nn": {
            "what": f"three_nn(xyz, new_xyz): the {m} sampled points as the known set of all {n} points (a10; not part of ms_per_pass)",
            "ms": tn_ms, "kernels_ms": tn_k, "ms_on_sorted_handles": tnh_ms, "ms_scan_kernel": tns_ms,
            "form": "boxed: both sets in sort-tile-recursive order (one nnp_sort launch), a wave = 64 consecutive sorted unknown points, "
                    "candidate superblocks nearest box first, per 16-record block the lanes' box bounds against their third-best, records "
                    "through scalar registers, the triple as 64-bit (distance, index) keys (interpolate.hip three_nn_boxes_kernel)",
            "identical_to_scan_kernel": bool(torch.equal(tn_d, tn_ds)) and bool(torch.equal(tn_i, tn_is)),
            "roofline": {"bound": "valu-issue",
                         "what": "the boxed kernel evaluates ~15 % of the B*n*m pairs (the union of 64 lanes' neighbourhoods: ~10 of 64 "
                                 "blocks per wave); `achieved` prices the B*n*m pair tests of the SCAN it replaces (8 flop each) against "
                                 "the fp32 vector peak -- an effective figure, like the culled Chamfer's",
                         "achieved": 8.0 * B * n * m / (tn_kms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": 8.0 * B * n * m / (tn_kms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         "scan_kernel_frac": 8.0 * B * n * m / (tns_k.get("three_nn", tns_ms) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}},
        "group_point": {"ms": gp_ms, "roofline": hbm_roof(4.0 * B * m * ns * (1 + 2 * 3), gp_k.get("group_point", gp_ms), "group_point",
                                                          "4*B*m*nsample*(1+2c) bytes (SURVEY 8(d)): 29 MB; inside rf_sample_and_group "
                                                          "the ball query writes grouped_xyz itself")},
        "checksum": {"fps_idx_sum": int(idx.long().sum().item()), "ball_idx_sum": int(gi.long().sum().item())},
    }


def run_c5(args, rank, world, dev, steps, warmup, use_pg):
    """BASELINE.json configs[4], this rank's share: B=32 samples of (3000 partial, 16384 gt)."""
    from rfnet_amd import glue, shard
    from rfnet_amd.rfnet import GroundTruth, RFNet
    B = args.batch
    rng = np.random.RandomState(500 + rank)
    partial = torch.from_numpy((rng.rand(B, 3000, 3) - 0.5).astype(np.float32)).to(dev)
    gt = torch.from_numpy((rng.rand(B, 16384, 3) - 0.5).astype(np.float32)).to(dev)
    torch.manual_seed(0)  # same random-init weights on every rank (checkpoint blob is missing: T10)
    net = RFNet().to(dev)

    def compute():
        with torch.no_grad():
            # gt preparation (one FPS run for both subsets + the sorted handle).  With eager launches: in line
            # (on a side stream under the forward it measured 8.45 .. 10.0 ms against 8.9,
            # tools/experiments/c5_overlap_ab.py).  Inside the captured graph it is a forked branch: the
            # serial FPS chain (32 workgroups) runs under the network's forward, 8.49 -> 8.05 ms
            # (tools/experiments/c5_graph_overlap.py)
            g = GroundTruth(gt, 64, 1024, overlap=torch.cuda.is_current_stream_capturing())
            p1, p2, p3, pf = net(partial)
            g.join()
            cd = glue.chamfer_per_sample(gt, pf, sorted1=g.h_gt)[0].mean(1)         # chamfer_big, per sample
            e1 = glue.earth_mover_cost(g.gt1, p1) / 64.0                             # earth_mover terms
            e2 = glue.earth_mover_cost(g.gt2, p2) / 1024.0
            return torch.stack([cd, e1, e2], 1)                                      # (B, 3)

    # The step has static shapes and ~450 kernel launches: captured ONCE into a HIP graph and replayed
    # (same kernels, same results, no per-launch host work, no gaps between dependent launches).  The
    # loss all-gather (RCCL) stays outside the graph.  Falls back to eager launches if capture fails.
    graphed = None
    graph_mode = "eager (--c5-eager)"
    from rfnet_amd._host import graph_replay_ok
    if not args.c5_eager and not graph_replay_ok(dev):
        graph_mode = ("eager (torch reductions do not replay from a HIP graph in this process: the HIP runtime was started "
                      "without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- under a profiler, for instance)")
        sys.stderr.write(f"bench.py: C5 step not captured -- {graph_mode}\n")
    elif not args.c5_eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    compute()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_per = compute()
            graphed = (graph, static_per)
            graph_mode = "hip graph"
            # one replay against the eager result before the graph is trusted (what a capture froze shows here)
            graph.replay()
            torch.cuda.synchronize()
            want = compute()
            if not torch.allclose(static_per, want, rtol=1e-4, atol=1e-6, equal_nan=True):
                graphed = None
                graph_mode = "eager (a replay differed from the eager step)"
                sys.stderr.write(f"bench.py: captured C5 step discarded -- {graph_mode}\n")
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write(f"bench.py: C5 step not captured ({type(exc).__name__}: {exc}); running it eagerly\n")
            graphed = None
            graph_mode = f"eager (capture failed: {type(exc).__name__})"

    def step():
        if graphed is not None:
            graphed[0].replay()
            per = graphed[1]
        else:
            per = compute()
        return shard.all_gather_per_sample(per, B * world, rank, world)              # loss reduction (RCCL)

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        full = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        full = step()
    fence()
    dt = time.perf_counter() - t0
    tm = torch.tensor([dt], dtype=torch.float64, device=dev)
    chk = full.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    if use_pg:
        torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    dt = float(tm.item())

    # Beyond configs[4]: the reference's TRAINING step on the same shard (vv_recon.py:474-504: forward, the
    # whole loss block, backward, Adam) through rfnet_amd.trainrun.TrainStep: forward + loss + backward
    # replayed from a HIP graph (checked against the eager gradients at capture), gradients all-reduced
    # over the ranks (RCCL, one flat 15 MB bucket), TensorFlow-form Adam.  A few steps: an extra, not the metric.
    train = None
    if not args.no_extras:
        from rfnet_amd.trainrun import TrainStep
        ts = TrainStep(net, B, graph=not args.c5_eager)
        for _ in range(2):
            ts(partial, gt)
        fence()
        tsteps = 8
        t0 = time.perf_counter()
        for _ in range(tsteps):
            tl = ts(partial, gt)
        fence()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_pg:
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None))
        train = {"ms_per_step": float(tt.item()) / tsteps * 1e3, "samples_per_s": world * B * tsteps / float(tt.item()),
                 "steps": tsteps, "what": "forward + full training loss (vv_recon.py:474-500) + backward + gradient "
                                          "all-reduce over the ranks + Adam update (rfnet_amd.trainrun.TrainStep)",
                 "mode": ts.graph_note, "loss_after": float(tl), "grad_norm": float(gnorm),
                 "finite": bool(torch.isfinite(gnorm).item())}
    return {
        "train_step": train,
        "workload": f"RFNet recurrent forward (3000 -> 64 -> 1024 -> 16384 points, random-init weights) + "
                    f"chamfer_big(gt, out) + earth_mover at 64^2 and 1024^2, B={B} per GPU "
                    f"(BASELINE.json configs[4]: B={B * world} over {world} GPU), per-sample losses all-gathered",
        "value": world * B * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
        "hip_graph": graphed is not None, "mode": graph_mode,
        "gathered_losses_shape": list(full.shape), "finite": bool(torch.isfinite(full).all().item()),
        "losses_equal_across_ranks": bool(lo.item() == hi.item()),
        "mean_losses": [float(x) for x in full.double().mean(0).tolist()],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--workload", choices=["chamfer", "c5"], default="chamfer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--c5-eager", action="store_true", help="run the C5 step with eager launches instead of a captured HIP graph")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY: exercise the multi-rank plumbing (self-launch, init, fences, "
                         "max-over-ranks, loss gather, rank-0 JSON) on CPU/gloo with placeholder work; prints "
                         "a line marked dry_run and measures nothing")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)  # never returns; nothing above touched the GPU
    # stdout carries exactly ONE line (rank 0's JSON): whatever the libraries underneath print to
    # file descriptor 1 (RCCL's version banner, ...) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    if args.dry_run_cpu:
        return dry_run_cpu(args, emit)

    from rfnet_amd import _lib, shard
    from rfnet_amd._raw import ChamferStep, approx_match, earth_mover, match_cost, nn_distance

    rank, world, local = shard.init_from_env()
    use_pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    if use_pg:
        world = torch.distributed.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"bench.py: process group has {world} ranks but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    B, N, M = args.batch, args.n, args.m

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.workload == "c5":
        c5 = run_c5(args, rank, world, dev, args.steps, args.warmup, use_pg)
        if rank == 0:
            emit({
                "metric": "samples/sec RFNet forward + CD/EMD loss (BASELINE.json configs[4])",
                "value": c5["value"], "unit": "samples/s", "n_gpus": world, "rccl_ranks": world if use_pg else 1,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": c5["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "config": {"workload": c5["workload"], "batch_per_gpu": B,
                                                "sharding": f"batch x{world}"}, "c5": c5})
        if use_pg:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    # synthetic data of configs[1]'s shape; each rank owns its own B samples (weak scaling)
    rng = np.random.RandomState(100 + rank)
    xyz1 = torch.from_numpy(rng.randn(B, N, 3).astype(np.float32)).to(dev)
    xyz2 = torch.from_numpy(rng.randn(B, M, 3).astype(np.float32)).to(dev)
    gd1 = torch.ones(B, N, device=dev)
    gd2 = torch.ones(B, M, device=dev)
    plan = ChamferStep(B, N, M, dev)  # outputs + workspace allocated once; a step is ONE C-ABI call

    def step():
        return plan(xyz1, xyz2, gd1, gd2)

    for _ in range(args.warmup):
        step()
    # un-instrumented timing of exactly K steps -> value
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    checksum = float(out[0].double().sum().item())
    # the same K steps again with per-kernel hipEvents (librfops records them on the launch
    # stream) -> roofline.achieved; its wall time is reported next to the un-instrumented one
    _lib.profile_collect()
    _lib.profile_enable(True)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt_prof = time.perf_counter() - t1
    _lib.profile_enable(False)
    prof = _lib.profile_collect()

    def forward_profile(a, c, mode, reps):
        """avg ms per launch of each forward kernel in `mode`, outputs, culled counters"""
        stats = []
        o = nn_distance(a, c, mode=mode, stats=stats if mode != "dense" else None)
        torch.cuda.synchronize()
        _lib.profile_collect()
        _lib.profile_enable(True)
        for _ in range(reps):
            o = nn_distance(a, c, mode=mode)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        pr = _lib.profile_collect()
        return {k: v[0] / max(v[1], 1) for k, v in pr.items()}, o, stats

    # what the forward did: the culled sweep's own counters, and the dense sweep (every pair
    # evaluated) on the same inputs under the same hipEvent hooks
    reps = max(3, min(20, args.steps))
    auto_ms, chk, culled_stats = forward_profile(xyz1, xyz2, "auto", reps)
    dense_ms, dense_out, _ = forward_profile(xyz1, xyz2, "dense", reps)
    same_as_dense = all(bool(torch.equal(x, y)) for x, y in zip(chk, dense_out))

    extras = {}
    if not args.no_extras:
        # ---- independent steps in flight: throughput when the caller has more than one batch ----
        # `value` above issues the K steps back to back on ONE stream (a step = sort -> sweep -> grad, each
        # waiting for the one before).  A caller with several independent Chamfers (an evaluation over many
        # models, the five Chamfer terms of a training step) can issue them round-robin on S streams, each
        # with its own ChamferStep plan: the 96-workgroup sort and the kernel tails of one step then run
        # under another step's sweep.  Same kernels, same outputs, the latency of ONE step unchanged; cutting
        # a single step's batch over streams instead does NOT pay (tools/experiments/split_step_streams.py).
        pipe = {}
        for ns in (2, 3):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            plans = [ChamferStep(B, N, M, dev) for _ in range(ns)]

            def run(k):
                for i in range(k):
                    with torch.cuda.stream(streams[i % ns]):
                        plans[i % ns](xyz1, xyz2, gd1, gd2)

            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            run(max(args.warmup, ns))
            fence()
            tp = time.perf_counter()
            run(args.steps)
            fence()
            tpipe = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
            if use_pg:
                torch.distributed.all_reduce(tpipe, op=torch.distributed.ReduceOp.MAX)
            same = all(bool(torch.equal(x, y)) for pl in plans
                       for x, y in zip((pl.dist1, pl.idx1, pl.dist2, pl.idx2), out[:4]))
            pipe[str(ns)] = {"value": world * float(B) * N * M * args.steps / float(tpipe.item()), "unit": "pairs/s",
                             "ms_per_step": float(tpipe.item()) / args.steps * 1e3,
                             "outputs_identical_to_the_serial_step": same}
            del plans, streams
        extras["independent_steps_in_flight"] = {
            "what": "the same K steps issued round-robin on S streams, one ChamferStep plan per stream (throughput "
                    "with several independent batches in flight; `value` is the one-stream figure)",
            "streams": pipe}
        # ---- rotating inputs: `value` re-runs the same 7 MB of clouds K times, so inputs are L2/MALL-resident; a caller that
        # feeds fresh network output each step is colder.  Four distinct input sets, round-robin, same plan, same K steps:
        rot = []
        for i in range(4):
            rr = np.random.RandomState(1000 * (i + 1) + 100 + rank)
            rot.append((torch.from_numpy(rr.randn(B, N, 3).astype(np.float32)).to(dev),
                        torch.from_numpy(rr.randn(B, M, 3).astype(np.float32)).to(dev)))
        for i in range(max(args.warmup, 4)):
            plan(rot[i % 4][0], rot[i % 4][1], gd1, gd2)
        fence()
        tp = time.perf_counter()
        for i in range(args.steps):
            plan(rot[i % 4][0], rot[i % 4][1], gd1, gd2)
        fence()
        trot = torch.tensor([time.perf_counter() - tp], dtype=torch.float64, device=dev)
        if use_pg:
            torch.distributed.all_reduce(trot, op=torch.distributed.ReduceOp.MAX)
        extras["rotating_inputs"] = {
            "what": "the same K steps over 4 distinct input sets (randn, own seeds) round-robin through the same plan: "
                    "inputs not resident from the previous step",
            "sets": 4, "ms_per_step": float(trot.item()) / args.steps * 1e3,
            "value": world * float(B) * N * M * args.steps / float(trot.item()), "unit": "pairs/s"}
        del rot
        plan(xyz1, xyz2, gd1, gd2)  # (the plan's outputs back on the headline inputs)
        # ---- the distributions the operator meets in the model ---------------------------------
        byd = {}
        dsteps = max(5, min(20, args.steps))

        def measure(a, c, note):
            for _ in range(2):
                plan(a, c, gd1, gd2)
            fence()
            ts = time.perf_counter()
            for _ in range(dsteps):
                plan(a, c, gd1, gd2)
            fence()
            ms_step = (time.perf_counter() - ts) / dsteps * 1e3
            am, o_auto, st = forward_profile(a, c, "auto", 5)
            dm, o_dense, _ = forward_profile(a, c, "dense", 5)
            ent = {"ms_per_step": ms_step, "forward_ms_auto": sum(am.values()), "forward_ms_dense": sum(dm.values()),
                   "auto_kernels_ms": am, "note": note,
                   "identical_to_dense_sweep": all(bool(torch.equal(x, y)) for x, y in zip(o_auto, o_dense))}
            if "nnp_sweep" in am and len(st) >= 14:
                ev = float(st[12] + st[13])  # directed pairs evaluated, summed by the kernel
                ent.update({"forward": "culled", "evaluated_fraction_of_2BNM": ev / (2.0 * B * N * M),
                            "heaviest_wave_block_scans": [int(st[8]), int(st[9])],
                            "culled_faster_than_dense": sum(am.values()) < sum(dm.values())})
            else:
                ent["forward"] = "dense"
            return ent

        byd["randn"] = measure(xyz1, xyz2, "standard normal (the headline data)")
        for name, a, c, note in make_distributions(B, N, M, rank, dev):
            byd[name] = measure(a, c, note)
            del a, c
        extras["by_distribution"] = byd
        # ---- the reference's own bench shape (tf_ops/CD/tf_nndistance.py:35-61): randn(32,16384,3) vs randn(32,1024,3), fwd+bwd
        rrng = np.random.RandomState(100 + rank)
        ra = torch.from_numpy(rrng.randn(B, 16384, 3).astype(np.float32)).to(dev)
        rc = torch.from_numpy(rrng.randn(B, 1024, 3).astype(np.float32)).to(dev)
        rplan = ChamferStep(B, 16384, 1024, dev)
        rg1, rg2 = torch.ones(B, 16384, device=dev), torch.ones(B, 1024, device=dev)
        for _ in range(3):
            rout = rplan(ra, rc, rg1, rg2)
        fence()
        tr = time.perf_counter()
        for _ in range(dsteps):
            rout = rplan(ra, rc, rg1, rg2)
        fence()
        r_ms = (time.perf_counter() - tr) / dsteps * 1e3
        r_dense = nn_distance(ra, rc, mode="dense")
        extras["reference_bench_shape"] = {
            "workload": f"nn_distance fwd+bwd, B={B} per GPU, 16384 vs 1024 (the reference's own smoke/bench shape, "
                        "tf_ops/CD/tf_nndistance.py:35-61), randn seed 100, upstream grads of ones",
            "ms_per_step": r_ms, "value": world * float(B) * 16384 * 1024 / (r_ms * 1e-3), "unit": "pairs/s",
            "identical_to_dense_sweep": all(bool(torch.equal(x, y)) for x, y in zip(rout[:4], r_dense))}
        del ra, rc, rplan, rg1, rg2, rout, r_dense
        # ---- the reference's TRUE input size: the model's partial scans hold 3000 points, not 2048
        # (vv_recon.py:29,464; recon_test.py:20,57), and the forward meets 3000 x 16384 three times (vv_recon.py:213,225,238) --
        # the one ragged (not a multiple of 64) shape of the path
        trng = np.random.RandomState(100 + rank)
        ta = torch.from_numpy(trng.randn(B, 3000, 3).astype(np.float32)).to(dev)
        tc = torch.from_numpy(trng.randn(B, 16384, 3).astype(np.float32)).to(dev)
        tplan = ChamferStep(B, 3000, 16384, dev)
        tg1, tg2 = torch.ones(B, 3000, device=dev), torch.ones(B, 16384, device=dev)
        for _ in range(3):
            tout = tplan(ta, tc, tg1, tg2)
        fence()
        tr = time.perf_counter()
        for _ in range(dsteps):
            tout = tplan(ta, tc, tg1, tg2)
        fence()
        t_ms = (time.perf_counter() - tr) / dsteps * 1e3
        t_dense = nn_distance(ta, tc, mode="dense")
        t_k, _, t_st = forward_profile(ta, tc, "auto", 5)
        extras["reference_true_shape"] = {
            "workload": f"nn_distance fwd+bwd, B={B} per GPU, 3000 vs 16384 (the reference's true partial-scan size, "
                        "vv_recon.py:29,464), randn seed 100, upstream grads of ones",
            "ms_per_step": t_ms, "value": world * float(B) * 3000 * 16384 / (t_ms * 1e-3), "unit": "pairs/s",
            "forward_kernels_ms": t_k,
            "evaluated_fraction_of_2BNM": (float(t_st[12] + t_st[13]) / (2.0 * B * 3000 * 16384)) if len(t_st) >= 14 else None,
            "identical_to_dense_sweep": all(bool(torch.equal(x, y)) for x, y in zip(tout[:4], t_dense))}
        del ta, tc, tplan, tg1, tg2, tout, t_dense

        # ---- second half of BASELINE.json's metric string, "EMD iters/sec": one iter = one
        # approx_match + match_cost batch call on configs[3] (B=32, 2048 vs 2048, the reference's
        # 10-level schedule)
        eb, en = 32, 2048
        erng = np.random.RandomState(100 + rank)
        e1 = torch.from_numpy((erng.random_sample((eb, en, 3)) - 0.5).astype(np.float32)).to(dev)
        e2 = torch.from_numpy((erng.random_sample((eb, en, 3)) - 0.5).astype(np.float32)).to(dev)
        emd_steps = max(5, min(20, args.steps))
        for _ in range(2):
            cost = match_cost(e1, e2, approx_match(e1, e2))
        fence()
        t2 = time.perf_counter()
        for _ in range(emd_steps):
            cost = match_cost(e1, e2, approx_match(e1, e2))
        fence()
        dt_emd = time.perf_counter() - t2
        emd_checksum = float(cost.double().sum().item())
        emd_gpu_cost = [float(x) for x in cost.double().cpu().tolist()]
        for _ in range(2):
            fcost = earth_mover(e1, e2)
        fence()
        t2 = time.perf_counter()
        for _ in range(emd_steps):
            fcost = earth_mover(e1, e2)
        fence()
        dt_emdf = time.perf_counter() - t2
        emd_fused_checksum = float(fcost.double().sum().item())
        # the EMD half of the metric against ITS roof.  approx_match is 30 level sweeps of B*n*m evaluations, each
        # one v_exp_f32 (transcendental: 8 issue cycles) + 8 plain VALU ops; nothing in it is a product of a row
        # and a column factor, so MFMA does not apply (the K=4 distance-GEMM trial is recorded
        # there).  The roof is the measured issue floor of that instruction mix (tools/ubench/valu_rate.hip:
        # 3.4 cycles per instruction incl. the exp at the clock the chip holds): issue_floor_ms for C4.
        am_ms, am_k = timed(lambda: approx_match(e1, e2), max(3, emd_steps // 2), fence)
        mt = approx_match(e1, e2)
        mc_ms, mc_k = timed(lambda: match_cost(e1, e2, mt), emd_steps, fence)
        from rfnet_amd._raw import match_cost_grad
        mg_ms, mg_k = timed(lambda: match_cost_grad(e1, e2, mt), emd_steps, fence)
        del mt
        # (no kernel events -- a renamed kernel, events unavailable under a tool: fall back to the wall time, never divide by 0)
        am_kernel_ms = sum(am_k.values()) or am_ms
        # the issue floor models the sweeps (fp32 VALU + v_exp_f32) and the materialisation; the row sort, the init and the packing
        # kernels are reported beside it
        am_swept_ms = sum(v for k, v in am_k.items() if k in ("am_p1", "am_p2", "am_p3p1", "am_p3", "am_match")) or am_kernel_ms
        efl = emd_issue_floor_ms(eb, en, en)
        lane_ops = (efl["valu_per_pair"] if efl else 250.0) * eb * en * en
        # SURVEY 8(d) C4's secondary run: BASELINE's "50 Sinkhorn iters" as the 10 reference levels each repeated 5x
        lv50 = [float(x) for x in np.repeat([-4.0 ** j for j in range(7, -2, -1)] + [0.0], 5)]
        from rfnet_amd._raw import approx_match as approx_match_lv
        x50_steps = max(3, emd_steps // 4)
        for _ in range(1):
            c50 = match_cost(e1, e2, approx_match_lv(e1, e2, levels=lv50))
        fence()
        t2 = time.perf_counter()
        for _ in range(x50_steps):
            c50 = match_cost(e1, e2, approx_match_lv(e1, e2, levels=lv50))
        fence()
        dt_x50 = time.perf_counter() - t2
        x50_checksum = float(c50.double().sum().item())
        extras["per_op_roofline"] = {
            "match_cost": hbm_roof(4.0 * eb * en * en + 12.0 * eb * 2 * en, mc_k.get("mc_partial", mc_ms), "mc_partial",
                                   "4*B*n*m + 12*B*(n+m) bytes (SURVEY 8(d)): one pass over match"),
            "match_cost_grad": hbm_roof(4.0 * eb * en * en + 24.0 * eb * 2 * en, mg_k.get("mc_grad", mg_ms), "mc_grad",
                                        "ONE pass over match for both gradients (the reference makes two: SURVEY 8(d) counts 2x)"),
        }
        del e1, e2, cost, fcost

        # ---- north_star's own target shape: B=32 x 16384 vs 16384 forward ----------------------
        ns_n = 16384
        nrng = np.random.RandomState(200 + rank)
        y1 = torch.from_numpy(nrng.randn(B, ns_n, 3).astype(np.float32)).to(dev)
        y2 = torch.from_numpy(nrng.randn(B, ns_n, 3).astype(np.float32)).to(dev)
        ns_steps = max(3, min(10, args.steps))
        for _ in range(2):
            nn_distance(y1, y2)
        fence()
        t3 = time.perf_counter()
        for _ in range(ns_steps):
            nn_distance(y1, y2)
        fence()
        dt_ns = time.perf_counter() - t3
        ns_auto_ms, ns_out, ns_stats = forward_profile(y1, y2, "auto", 3)
        ns_dense_ms, ns_dense_out, _ = forward_profile(y1, y2, "dense", 2)
        ns_same = all(bool(torch.equal(x, y)) for x, y in zip(ns_out, ns_dense_out))
        del y1, y2, ns_out, ns_dense_out

        tmax = torch.tensor([dt_emd, dt_ns, dt_emdf, dt_x50], dtype=torch.float64, device=dev)
        if use_pg:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt_emd, dt_ns, dt_emdf, dt_x50 = (float(tmax[i].item()) for i in range(4))
        extras["emd"] = {
            "roofline": {
                "bound": "valu+trans", "kernel": "am_p1 + am_p2 + am_p3p1 + am_match (approx_match: 20 sweeps -- from the third level on over the live columns / rows of set 2 only -- + the materialisation; the row sort, one packing launch)",
                "lane_ops_per_pair": efl["valu_per_pair"] if efl else None, "exp_per_pair": efl["exp_per_pair"] if efl else None,
                "packed_instructions_per_pair": efl["packed_per_pair"] if efl else None,
                "achieved": lane_ops / (am_kernel_ms * 1e-3) / 1e12, "unit": "T lane-ops/s",
                "issue_floor_ms": efl["mix_ms"] if efl else None,
                "issue_floor_additive_ms": efl["additive_ms"] if efl else None,
                "issue_floor_source": (efl["constants"] + "; counted instructions per launch: EMD_LAUNCH_MIX in bench.py") if efl else
                                      "profiles/issue_costs.json missing",
                "frac": (efl["mix_ms"] / am_swept_ms) if efl else None,
                "swept_kernel_sum_ms": am_swept_ms,
                "live_columns": "from the third level on the sweeps run over the columns / rows of set 2 whose remainR is not exactly +0 (a dead "
                                "column adds fma(e, +0, acc) = acc): 56 % of C4's columns at level -1024, 3 % at level -1; every sum is over the same terms as "
                                "the reference's (another order: tolerance).  Round 5's Taylor expansion of the three broadest levels is gone "
                                "(tools/experiments/emd_fgt_route.patch.txt)",
                "frac_of_fp32_peak": 2.0 * lane_ops / (am_kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                # against the CHIP, not the model above: the reference schedule's 30*B*n*m exponentials per call over the whole
                # approx_match kernel time, against the transcendental issue roof (one v_exp_f32 wave-instruction per 8.14 cycles per
                # SIMD x 1024 SIMDs x 64 lanes x 2.36 GHz = 1.9e13 exp/s); `chip_frac_executed` prices only the exponentials the
                # kernels still execute (skipped and dead columns and squared levels execute none)
                "trans_roof_exp_per_s": TRANS_ROOF_EXP_PER_S,
                "chip_frac": 30.0 * eb * en * en / (am_kernel_ms * 1e-3) / TRANS_ROOF_EXP_PER_S,
                "chip_frac_executed": ((efl["exp_per_pair"] * eb * en * en / (am_kernel_ms * 1e-3) / TRANS_ROOF_EXP_PER_S) if efl else None),
                "chip_frac_is": "exp/s over the transcendental issue roof of the chip; `frac` is against the issue floor of the executed "
                                "instruction mix (a model with committed constants)",
                "avg_kernel_sum_ms": am_kernel_ms, "approx_match_ms_wall": am_ms, "kernels_ms": am_k,
                "mfma": "not applicable: every matrix element needs its own exp(level * d2) (8 of the 9 ops and all of the "
                        "transcendental work); d2 via the |a|^2+|b|^2-2ab GEMM is ruled out because exp(-16384 d2) amplifies "
                        "its cancellation error to ~7e-4 relative (DESIGN.md 5.5, K=4 trial recorded there)",
                "dense_equivalent": {"lane_ops_per_pair": 254, "exp_per_pair": 32,
                                     "what": "the schedule with every pair of every level evaluated; the terms left out are exact zeros"},
                "note": "frac = issue floor derived from EXECUTED instructions (counted, with the model's keep fractions for the "
                        "skipping sweeps; a packed instruction = two fp32 operations per lane, counted once and priced at its own measured "
                        "issue cost; the 1.155 non-additivity factor was measured on the scalar column mix) x measured issue costs / measured kernel time; "
                        "frac_of_fp32_peak counts 2 flop per lane-op against the 157.3 TFLOP/s vector peak"},
            "metric": "EMD iters/sec (approx_match + match_cost batch calls)",
            "value": world * emd_steps / dt_emd, "unit": "calls/s", "ms_per_call": dt_emd / emd_steps * 1e3,
            "level_sweeps_per_s": world * emd_steps * 30 / dt_emd,
            "exp_evals_per_s": world * emd_steps * 30.0 * eb * en * en / dt_emd,
            "workload": f"B={eb} per GPU, {en} vs {en}, reference 10-level schedule "
                        "(BASELINE.json configs[3]); uniform(-0.5,0.5) seed 100",
            "steps": emd_steps, "checksum": emd_checksum,
            "extended_50": {"metric": "approx_match (50-level schedule: the 10 reference levels x 5) + match_cost batch calls",
                            "value": world * x50_steps / dt_x50, "unit": "calls/s", "ms_per_call": dt_x50 / x50_steps * 1e3,
                            "level_sweeps_per_s": world * x50_steps * 150 / dt_x50, "steps": x50_steps, "checksum": x50_checksum,
                            "parity": "tests/test_gpu_emd.py::test_extended_schedule_50_levels_c4_size (one sample vs the oracle "
                                      "on the same schedule; no reference counterpart: SURVEY 8(d) C4)"},
            "fused": {"op": "rf_earth_mover (same cost, match never written to HBM)",
                      "value": world * emd_steps / dt_emdf, "unit": "calls/s",
                      "ms_per_call": dt_emdf / emd_steps * 1e3, "checksum": emd_fused_checksum},
        }
        extras["north_star_16384sq"] = {
            "workload": f"nn_distance forward, B={B} per GPU, {ns_n} vs {ns_n} (north_star target shape)",
            "value": world * B * ns_n * ns_n * ns_steps / dt_ns, "unit": "pairs/s",
            "ms_per_call": dt_ns / ns_steps * 1e3, "steps": ns_steps,
            "identical_to_dense_sweep": ns_same,
            "kernels_ms": ns_auto_ms, "dense_kernels_ms": ns_dense_ms,
        }
        if "nnp_sweep" in ns_auto_ms and len(ns_stats) >= 14:
            ev = float(ns_stats[12] + ns_stats[13])  # directed pairs the sweep evaluated, summed by the kernel itself
            t_sw = ns_auto_ms["nnp_sweep"] * 1e-3
            extras["north_star_16384sq"]["roofline"] = {
                "bound": "valu", "kernel": "nnp_sweep", "achieved": 8.0 * ev / t_sw / 1e12, "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": 8.0 * ev / t_sw / 1e12 / FP32_PEAK_TFLOPS,
                "evaluated_fraction_of_2BNM": ev / (2.0 * B * ns_n * ns_n), "avg_launch_ms": t_sw * 1e3,
                "algorithmic_achieved": 16.0 * B * ns_n * ns_n / t_sw / 1e12,
                "forward_speedup_vs_dense": sum(ns_dense_ms.values()) / sum(ns_auto_ms.values())}
        # ---- configs[2]: FPS + gather + ball query + group ------------------------------------
        extras["c3"] = run_c3(rank, dev, fence, max(5, min(20, args.steps)))
        # ---- configs[4] on this rank's share (also `--workload c5`) -----------------------------
        extras["c5"] = run_c5(args, rank, world, dev, steps=3, warmup=1, use_pg=use_pg)

    tm = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_pg:
        torch.distributed.all_reduce(tm, op=torch.distributed.ReduceOp.MAX)
    dt = float(tm.item())

    if rank == 0:
        pairs_per_step = B * N * M
        culled = "nnp_sweep" in prof
        kname = "nnp_sweep" if culled else "nn_sweep"
        sweep_ms, sweep_n = prof.get(kname, (0.0, 0))
        sweep_avg_s = (sweep_ms / max(sweep_n, 1)) * 1e-3
        dsweep_avg_s = dense_ms.get("nn_sweep", 0.0) * 1e-3
        flops = 16.0 * B * N * M
        hbm_bytes = 20.0 * B * (N + M)
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = {k: tj.get(k, {}).get(f"{B}x{N}x{M}") for k in ("nn_sweep", "nnp_sweep")}
            except Exception:
                traffic = {}
        if culled and len(culled_stats) >= 14 and sweep_avg_s:
            evaluated = float(culled_stats[12] + culled_stats[13])  # directed pairs evaluated, summed by the kernel (a launch may mix 1024- and 16-pair scans)
            executed_tf = 8.0 * evaluated / sweep_avg_s / 1e12
            roof = {
                "bound": "valu",
                "pipe": "fp32 VALU (no MFMA use; peak = fp32 vector peak)",
                "kernel": kname,
                "achieved": executed_tf,
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": executed_tf / FP32_PEAK_TFLOPS,
                "traffic": traffic.get(kname),
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                  "committed; NOT measured in this run)",
                "note": "achieved = 8 flop x directed pairs the kernel evaluated (its own counters) / avg launch "
                        "duration; the culling is reported as algorithmic_speedup_vs_dense",
                "executed_flops_per_launch": 8.0 * evaluated,
                "evaluated_directed_pairs": evaluated,
                "evaluated_fraction_of_2BNM": evaluated / (2.0 * B * N * M),
                "avg_launch_ms": sweep_avg_s * 1e3,
                "launches": sweep_n,
                "algorithmic_flops_per_launch": flops,
                "algorithmic_achieved": flops / sweep_avg_s / 1e12,
                "algorithmic_speedup_vs_dense": (dsweep_avg_s / sweep_avg_s) if dsweep_avg_s else None,
                "forward_speedup_vs_dense": (sum(dense_ms.values()) / sum(auto_ms.values())) if auto_ms else None,
                "identical_to_dense_sweep": same_as_dense,
            }
        else:
            ach = flops / sweep_avg_s / 1e12 if sweep_avg_s else None
            roof = {"bound": "valu", "pipe": "fp32 VALU", "kernel": kname, "achieved": ach, "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS if ach else None,
                    "traffic": traffic.get(kname), "flops_per_launch": flops, "avg_launch_ms": sweep_avg_s * 1e3,
                    "launches": sweep_n}
        kernel_sum_ms = sum(v[0] for v in prof.values()) / args.steps
        line = {
            "metric": "point-pairs/sec Chamfer (BxNxM)",
            "value": world * pairs_per_step * args.steps / dt,
            "value_is": (f"effective rate: B*N*M pairs per step / time; the culled sweep evaluated "
                         f"{100.0 * evaluated / (2.0 * B * N * M):.1f} % of the 2*B*N*M directed pairs and returns outputs "
                         "bit-identical to the sweep that evaluates all of them (roofline.identical_to_dense_sweep)") if culled and len(culled_stats) >= 14 and sweep_avg_s
                        else "B*N*M pairs per step / time; every pair evaluated",
            "unit": "pairs/s",
            "n_gpus": world,
            "rccl_ranks": world if use_pg else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"Chamfer nn_distance fwd+bwd, B={B} per GPU, {N} vs {M} points "
                            "(BASELINE.json configs[1]), randn seed 100",
                "batch_per_gpu": B, "n": N, "m": M, "sharding": f"batch x{world}",
                "forward": "culled sweep (nn_pruned.hip)" if culled else "dense sweep (nn_distance.hip)",
                "step": "rf_chamfer_step: one C-ABI call, caller-allocated outputs",
                "scaling_note": f"weak: every GPU runs B={B}; N GPUs = a global batch of {B}*N (BASELINE.json configs[4]'s "
                                "B=256 is 8 x 32)",
            },
            "roofline": roof,
            "roofline_dense": {
                "bound": "valu", "kernel": "nn_sweep (RF_NN_DENSE: every pair evaluated)",
                "achieved": flops / dsweep_avg_s / 1e12 if dsweep_avg_s else None,
                "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / dsweep_avg_s / 1e12 / FP32_PEAK_TFLOPS if dsweep_avg_s else None,
                "traffic": traffic.get("nn_sweep"),
                "avg_launch_ms": dsweep_avg_s * 1e3,
            },
            "roofline_hbm": {
                "bound": "hbm", "kernel": kname,
                "achieved": hbm_bytes / sweep_avg_s / 1e9 if sweep_avg_s else None,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": hbm_bytes / sweep_avg_s / 1e9 / HBM_PEAK_GBPS if sweep_avg_s else None,
                "bytes_per_launch": hbm_bytes,
            },
            "roofline_backward": hbm_roof(
                44.0 * B * (N + M), (prof.get("nnp_grad_sorted", prof.get("nn_grad", (0.0, 1)))[0] /
                                     max(prof.get("nnp_grad_sorted", prof.get("nn_grad", (0.0, 1)))[1], 1)) or 1e-9,
                "nnp_grad_sorted" if "nnp_grad_sorted" in prof else "nn_grad",
                "44*B*(N+M) algorithmic bytes incl. zero fill (SURVEY 8(d)); the sorted-space backward is bound by its fixed chain "
                "(launch, masks, list, barriers, scattered stores: ~9 us without a single visit, profiles/r03_ab_grad_sorted.txt), not by "
                "HBM; its scatter terms meet in LDS as DOUBLES since ds_add_f64 runs 23x the rate of ds_add_f32 on this chip "
                "(profiles/r05_lds_atomic_rate.txt: 16.3 -> 12.5 us)"),
            "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items())},
            "kernel_sum_ms_per_step": kernel_sum_ms,
            # (kernel_sum comes from the SECOND, hipEvent-instrumented pass over the same K steps, whose event records sit between the
            # launches and lengthen it: ms_per_step_instrumented.  The un-instrumented step can therefore be SHORTER than that sum;
            # no "host overhead" is derived from the two.)
            "ms_per_step_instrumented": dt_prof / args.steps * 1e3,
            "checksum": checksum,
            # how far the checker itself is pinned: the GPU path is bit-exact / in
            # tolerance against oracle/rfops_oracle.c in tests/; this says what pins THAT oracle
            "parity": {
                "gpu_vs_oracle": "tests/ -m gpu (bit-exact dist/idx/FPS/ball-query, stated tolerances for EMD and gradients)",
                "oracle_pinned_by_reference_build": ["nn_distance", "nn_distance_grad", "three_nn", "three_interpolate(+grad)",
                                                     "query_ball_point idx", "group_point(+grad)", "match_cost", "match_cost_grad (grad2, grad1.x)",
                                                     "approx_match through the reference's 11-level CPU schedule"],
                "oracle_restated_from_source_only": ["farthest_point_sample", "gather_point(+grad)", "approx_match 10-level CUDA schedule",
                                                     "query_ball_point pts_cnt", "auction_match", "prob_sample", "select_top_k",
                                                     "RFNet graph + loss block (oracle/rfnet_oracle.py)"],
            },
        }
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, N, M, 100, sample_b=B, with_grad=True)
            allc = cpu_baseline_all_cores(B, N, M, 100, with_grad=True)
            if allc:
                line["cpu_baseline_all_cores"] = allc
            if "emd" in line:
                ecb = emd_cpu_baseline(32, 2048, 100, 4, emd_gpu_cost)
                if ecb:
                    line["emd"]["cpu_baseline"] = ecb
                    line["emd"]["vs_cpu_baseline"] = line["emd"]["value"] / ecb["value"]
            if "north_star_16384sq" in line:
                cb = cpu_baseline(B, 16384, 16384, 200, sample_b=3, with_grad=False)
                line["north_star_16384sq"]["cpu_baseline"] = cb
                line["north_star_16384sq"]["vs_cpu_baseline"] = line["north_star_16384sq"]["value"] / cb["value"]
        emit(line)
    if use_pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
