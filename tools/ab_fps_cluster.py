#!/usr/bin/env python3
"""farthest_point_sample at C3 (32 x 16384 -> 1024, U[0,1)^3 seed 100): the one-workgroup-per-cloud kernel against the
cluster form (k = 2, 4, 8 workgroups per cloud; membership by arrival or keyed on the block index = one XCD per
cluster under round-robin placement).  Same device, same process; indices compared, error word read, hipEvent time of
20 calls each.  Also: smaller shapes, tie-heavy lattices, and two cluster launches in flight on two streams.
usage: python tools/ab_fps_cluster.py [--quick]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from rfnet_amd import _lib, _raw as R


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    quick = "--quick" in sys.argv
    rng = np.random.RandomState(100)
    xyz = torch.from_numpy(rng.random_sample((32, 16384, 3)).astype(np.float32)).cuda()
    ref = R.farthest_point_sample(1024, xyz)
    base = timed(lambda: R.farthest_point_sample(1024, xyz))
    print(f"C3 one workgroup per cloud        {base:.4f} ms  ({base / 1023 * 1e3:.3f} us/iteration)")
    for k in (2, 4, 8):
        for sm in (False, True):
            out, st = R.farthest_point_sample_cluster(1024, xyz, k=k, static_map=sm, return_state=True)
            torch.cuda.synchronize()
            err = int(st[1].item())
            same = bool((out == ref).all().item())
            ms = timed(lambda: R.farthest_point_sample_cluster(1024, xyz, k=k, static_map=sm))
            print(f"C3 cluster k={k} {'block-keyed (XCD)' if sm else 'by arrival       '} {ms:.4f} ms  "
                  f"({ms / 1023 * 1e3:.3f} us/iteration)  identical={same} error_word={err}")
    # kernel-only time from the library's own event brackets
    _lib.profile_collect(); _lib.profile_enable(True)
    for _ in range(10):
        R.farthest_point_sample(1024, xyz)
        for k in (2, 4, 8):
            R.farthest_point_sample_cluster(1024, xyz, k=k)
    torch.cuda.synchronize(); _lib.profile_enable(False)
    print("kernel events (all k mixed):", {k_: round(v[0] / v[1], 4) for k_, v in _lib.profile_collect().items()})
    if quick:
        return
    # other shapes / ties
    bad = 0
    cases = 0
    for (b, n, m) in [(8, 16384, 64), (8, 3000, 32), (16, 8192, 512), (8, 5000, 700), (8, 1500, 1500), (24, 12000, 256)]:
        for kind in ("uniform", "lattice", "dup"):
            if kind == "uniform":
                a = rng.random_sample((b, n, 3))
            elif kind == "lattice":
                a = rng.randint(0, 12, size=(b, n, 3)) / 11.0   # many exact ties
            else:
                a = rng.random_sample((b, n, 3)); a[:, n // 2:] = a[:, :n - n // 2]
            x = torch.from_numpy(a.astype(np.float32)).cuda()
            want = R.farthest_point_sample(m, x)
            for k in (2, 4, 8):
                for sm in (False, True):
                    got, st = R.farthest_point_sample_cluster(m, x, k=k, static_map=sm, return_state=True)
                    torch.cuda.synchronize()
                    cases += 1
                    if int(st[1].item()) or not bool((got == want).all().item()):
                        bad += 1
                        print("MISMATCH", b, n, m, kind, k, sm, int(st[1].item()))
    print(f"shapes/ties: {cases} cluster runs, {bad} mismatches")
    # two launches in flight: more workgroups than the chip holds at once, on two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t0 = time.time()
    outs = []
    for rep in range(5):
        with torch.cuda.stream(s1):
            outs.append(R.farthest_point_sample_cluster(1024, xyz, k=8, return_state=True))
        with torch.cuda.stream(s2):
            outs.append(R.farthest_point_sample_cluster(1024, xyz, k=8, return_state=True))
    torch.cuda.synchronize()
    ok = all(bool((o == ref).all().item()) and int(st[1].item()) == 0 for o, st in outs)
    print(f"two streams x 5 launches of k=8 (256 workgroups each): identical={ok}  {time.time() - t0:.3f} s wall")


if __name__ == "__main__":
    main()
