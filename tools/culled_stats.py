#!/usr/bin/env python3
"""Per-direction anatomy of the culled Chamfer sweep at one shape: time of each direction alone
(rf_nn_distance_sorted with one direction), the sort, and the kernel's own counters.
usage: python tools/culled_stats.py [B N M] [--dist randn|uniform|rfnet]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _lib  # noqa: E402
from rfnet_amd import _raw as R  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_collect()
    _lib.profile_enable(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    return {k: v[0] / v[1] for k, v in _lib.profile_collect().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*", type=int, default=[32, 2048, 16384])
    ap.add_argument("--dist", default="randn")
    a = ap.parse_args()
    B, N, M = a.shape
    rng = np.random.RandomState(100)
    if a.dist == "randn":
        x1, x2 = rng.randn(B, N, 3), rng.randn(B, M, 3)
    elif a.dist == "uniform":
        x1, x2 = rng.rand(B, N, 3) - 0.5, rng.rand(B, M, 3) - 0.5
    else:
        from rfnet_amd.rfnet import RFNet
        torch.manual_seed(0)
        net = RFNet().cuda()
        part = torch.from_numpy((rng.rand(B, 3000, 3) - 0.5).astype(np.float32)).cuda()
        with torch.no_grad():
            out = net(part)[3]
        x1, x2 = part[:, :N].cpu().numpy(), out.cpu().numpy()
        np.save(os.path.join("gpurun_out", "rfnet_out_sample.npy"), x2[:2])
        np.save(os.path.join("gpurun_out", "rfnet_in_sample.npy"), part[:2].cpu().numpy())
    t1 = torch.from_numpy(x1.astype(np.float32)).cuda()
    t2 = torch.from_numpy(x2.astype(np.float32)).cuda()
    for _ in range(3):  # (warm: the first launch of a process runs several times longer, and the phase stamps of the instrumented builds with it)
        R.nn_distance(t1, t2, mode="culled")
    torch.cuda.synchronize()
    st = []
    R.nn_distance(t1, t2, mode="culled", stats=st)
    for d in range(2):
        if not st[4 * d]:
            continue  # (a phase-stamp build: no per-wave counters)
        w, steps, mx, scans = st[4 * d:4 * d + 4]
        unit = st[14 + d] or 1024
        pairs = st[12 + d]  # summed by the kernel: exact also when a launch mixes the two scan forms
        print(f"dir{d}: waves {w} steps/wave {steps / max(w, 1):.1f} (max {mx}) block scans/wave {scans / max(w, 1):.1f} "
              f"(max {st[8 + d]}; {unit} pairs each) evaluated pairs {pairs:.3e} of {B * N * M:.3e} = {pairs / (B * N * M):.4f}")
    if any(st[25:32]):
        names = ["keys", "seed", "tile list", "query x superblock", "quad x block", "drain", "epilogue"]
        tot = sum(st[25:32])
        print("quad-tile phases (s_memtime, summed over waves): " + "  ".join(f"{n} {100 * v / tot:.0f} %" for n, v in zip(names, st[25:32]))
              + f"   | {tot / max(st[24], 1):.0f} ticks per sampled wave ({st[24]} waves; 2.4 ticks per ns)")
    if len(st) > 22 and st[22] and st[22] < 1 << 20 and not any(st[25:32]):  # (RFP_SG_STAMPS build: one-wave groups, one wave in 64)
        names = ["prologue + keys", "step heads (key min, box load, block tests)", "block scans", "re-scans", "second traversal", "outputs + emit"]
        tot = sum(st[16:22])
        print("one-wave-group phases (s_memtime, summed over sampled waves): " + "  ".join(f"{n} {100 * v / tot:.0f} %" for n, v in zip(names, st[16:22]))
              + f"   | {tot / st[22]:.0f} ticks per sampled wave ({st[22]} waves; 2.4 ticks per ns)")
    print("full culled forward:", timed(lambda: R.nn_distance(t1, t2, mode="culled")))
    h1, h2 = R.nn_sort(t1), R.nn_sort(t2)
    print("sort N alone:", timed(lambda: R.nn_sort(t1)))
    print("sort M alone:", timed(lambda: R.nn_sort(t2)))
    print("sweep both (sorted):", timed(lambda: R.nn_distance_sorted(h1, h2)))
    print("sweep dir1 only (N queries in M):", timed(lambda: R.nn_distance_sorted(h1, h2, True, False)))
    print("sweep dir2 only (M queries in N):", timed(lambda: R.nn_distance_sorted(h1, h2, False, True)))
    print("dense both:", timed(lambda: R.nn_distance(t1, t2, mode="dense")))
    print("dense dir1 only:", timed(lambda: R.nn_distance_dir(t1, t2, True, False)) if False else "")


if __name__ == "__main__":
    main()
