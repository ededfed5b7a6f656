#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run ONLY in the build container (needs /root/reference).

Two kinds of expected outputs are stored, and every key says which one it is:

  ref_*   produced by the reference's own CPU bodies, compiled from /root/reference by
          oracle/build_ref.sh (oracle/_ref/libref.so).  These pin the oracle.
  cuda_*  produced by oracle/rfops_oracle.c, the restatement of the reference CUDA ops,
          for ops that have NO compilable reference body (FPS, gather_point, the 10-level
          CUDA approx_match, match_cost on [b][m][n], pts_cnt).  They pin the HIP path
          against the oracle across toolchain changes, not the oracle itself.

Fixtures are data only (inputs + expected outputs); no reference source is stored.
Usage:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle.oracle import Oracle, Ref  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    orc, ref = Oracle(), Ref()

    # ---- C1: BASELINE.json configs[0]: B=4, N=M=1024, RandomState(100).randn ------------
    rng = np.random.RandomState(100)
    x1 = rng.randn(4, 1024, 3).astype(np.float32)
    x2 = rng.randn(4, 1024, 3).astype(np.float32)
    d1, i1, d2, i2 = ref.nn_distance(x1, x2)
    save("nn_distance_c1", xyz1=x1, xyz2=x2, ref_dist1=d1, ref_idx1=i1, ref_dist2=d2, ref_idx2=i2)

    # ---- ragged sizes (not multiples of any tile), and a duplicated-points case ----------
    rng = np.random.RandomState(100)
    out = {}
    for tag, (n, m) in {"a": (513, 1030), "b": (64, 3000)}.items():
        a = rng.randn(2, n, 3).astype(np.float32)
        c = rng.randn(2, m, 3).astype(np.float32)
        r = ref.nn_distance(a, c)
        gd1 = rng.randn(2, n).astype(np.float32)
        gd2 = rng.randn(2, m).astype(np.float32)
        g1, g2 = ref.nn_distance_grad(a, c, gd1, r[1], gd2, r[3])
        out.update({f"{tag}_xyz1": a, f"{tag}_xyz2": c, f"{tag}_ref_dist1": r[0],
                    f"{tag}_ref_idx1": r[1], f"{tag}_ref_dist2": r[2], f"{tag}_ref_idx2": r[3],
                    f"{tag}_gd1": gd1, f"{tag}_gd2": gd2, f"{tag}_ref_grad1": g1,
                    f"{tag}_ref_grad2": g2})
    # duplicates as data_util.resample_pcd makes them (data_util.py:8-13): exact ties
    base = rng.randn(2, 300, 3).astype(np.float32)
    sel = np.concatenate([rng.permutation(300), rng.randint(300, size=700)])
    a = np.ascontiguousarray(base[:, sel])
    c = np.ascontiguousarray(base[:, rng.randint(300, size=777)])
    r = ref.nn_distance(a, c)
    out.update({"dup_xyz1": a, "dup_xyz2": c, "dup_ref_dist1": r[0], "dup_ref_idx1": r[1],
                "dup_ref_dist2": r[2], "dup_ref_idx2": r[3]})
    save("nn_distance_ragged", **out)

    # ---- interpolation: shapes of tf_ops/interpolation/tf_interpolate_op_test.py ----------
    rng = np.random.RandomState(100)
    unknown = rng.random_sample((1, 128, 3)).astype(np.float32)
    known = rng.random_sample((1, 8, 3)).astype(np.float32)
    points = rng.random_sample((1, 8, 16)).astype(np.float32)
    dist, idx = ref.three_nn(unknown, known)
    weight = np.full((1, 128, 3), 1.0 / 3.0, np.float32)
    interp = ref.three_interpolate(points, idx, weight)
    gout = rng.random_sample((1, 128, 16)).astype(np.float32)
    ginterp = ref.three_interpolate_grad(points, idx, weight, gout)
    # a second, bigger, non-uniform-weight case with m < 3 edge case alongside
    u2 = rng.randn(3, 257, 3).astype(np.float32)
    k2 = rng.randn(3, 70, 3).astype(np.float32)
    d2_, i2_ = ref.three_nn(u2, k2)
    k_small = rng.randn(3, 2, 3).astype(np.float32)
    d3_, i3_ = ref.three_nn(u2, k_small)
    w2 = rng.random_sample((3, 257, 3)).astype(np.float32)
    p2 = rng.randn(3, 70, 5).astype(np.float32)
    o2 = ref.three_interpolate(p2, i2_, w2)
    go2 = rng.randn(3, 257, 5).astype(np.float32)
    g2_ = ref.three_interpolate_grad(p2, i2_, w2, go2)
    save("interpolate", xyz1=unknown, xyz2=known, points=points, weight=weight, grad_out=gout,
         ref_dist=dist, ref_idx=idx, ref_out=interp, ref_grad_points=ginterp,
         b_xyz1=u2, b_xyz2=k2, b_ref_dist=d2_, b_ref_idx=i2_, b_xyz2_small=k_small,
         b_small_ref_dist=d3_, b_small_ref_idx=i3_, b_points=p2, b_weight=w2, b_ref_out=o2,
         b_grad_out=go2, b_ref_grad_points=g2_)

    # ---- grouping: uniform [0,1)^3 like tf_grouping.py:80-88, queries drawn from dataset ---
    rng = np.random.RandomState(100)
    ds = rng.random_sample((2, 512, 3)).astype(np.float32)
    q = np.ascontiguousarray(ds[:, rng.permutation(512)[:128]])
    feat = rng.random_sample((2, 512, 5)).astype(np.float32)
    out = {"xyz1": ds, "xyz2": q, "points": feat}
    for r_, ns in ((0.1, 32), (0.3, 64)):
        tag = f"r{int(r_ * 10)}_k{ns}"
        idx = ref.query_ball_point(r_, ns, ds, q)
        _, cnt = orc.query_ball_point(r_, ns, ds, q)
        grouped = ref.group_point(feat, idx)
        gout = rng.random_sample((2, 128, ns, 5)).astype(np.float32)
        gpts = ref.group_point_grad(feat, idx, gout)
        out.update({f"{tag}_ref_idx": idx, f"{tag}_cuda_pts_cnt": cnt, f"{tag}_ref_grouped": grouped,
                    f"{tag}_grad_out": gout, f"{tag}_ref_grad_points": gpts})
    save("grouping", **out)

    # ---- EMD: approxmatch_cpu / matchcost_cpu (11-level CPU variant, [b][n][m]) and the
    #      CUDA-schedule restatement ([b][m][n]) at 64x64 and 257x130 -----------------------
    rng = np.random.RandomState(100)
    out = {}
    for tag, (n, m) in {"sq": (64, 64), "rag": (257, 130)}.items():
        a = (rng.random_sample((2, n, 3)) - 0.5).astype(np.float32)
        c = (rng.random_sample((2, m, 3)) - 0.5).astype(np.float32)
        m_ref = ref.approxmatch_cpu(a, c)  # [b][n][m]
        cost_ref = ref.matchcost_cpu(a, c, m_ref)
        _, g2_ref = ref.matchcostgrad_cpu(a, c, m_ref)
        m_cuda = orc.approx_match(a, c)  # [b][m][n]
        cost_cuda = orc.match_cost(a, c, m_cuda)
        g1c, g2c = orc.match_cost_grad(a, c, m_cuda)
        out.update({f"{tag}_xyz1": a, f"{tag}_xyz2": c, f"{tag}_ref_match_nm": m_ref,
                    f"{tag}_ref_cost": cost_ref, f"{tag}_ref_grad2": g2_ref,
                    f"{tag}_cuda_match_mn": m_cuda, f"{tag}_cuda_cost": cost_cuda,
                    f"{tag}_cuda_grad1": g1c, f"{tag}_cuda_grad2": g2c})
    save("emd", **out)

    # ---- sampling (CUDA-only in the reference): restatement outputs -----------------------
    rng = np.random.RandomState(100)
    pts = rng.random_sample((3, 3000, 3)).astype(np.float32)
    idx = orc.farthest_point_sample(64, pts)
    gathered = orc.gather_point(pts, idx)
    gout = rng.randn(3, 64, 3).astype(np.float32)
    ginp = orc.gather_point_grad(pts, idx, gout)
    # tie case for the (k mod 512) rule: every point duplicated at k and k+512+1 etc.
    tie = np.ascontiguousarray(np.tile(rng.random_sample((1, 600, 3)).astype(np.float32), (1, 3, 1)))
    idx_tie = orc.farthest_point_sample(40, tie)
    save("sampling", inp=pts, cuda_fps_idx=idx, cuda_gathered=gathered, grad_out=gout,
         cuda_grad_inp=ginp, tie_inp=tie, cuda_tie_fps_idx=idx_tie)


if __name__ == "__main__":
    main()
