"""TEST INFRASTRUCTURE ONLY: numpy front-ends of the C oracle and of oracle/_ref.

`Oracle`  -> liboracle.so, the C restatement of the reference CUDA ops (rfops_oracle.c).
`Ref`     -> _ref/libref.so, the reference's own TF-free CPU bodies compiled from
             /root/reference by build_ref.sh (exists only where that was built).

Every method takes / returns C-contiguous numpy arrays with the reference op's shapes
and dtypes (float32 / int32).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_F)


def _i(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_I)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def build(force=False):
    """Compile liboracle.so (and _ref/libref.so when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rfops_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libref.so")
    if os.path.isdir(os.environ.get("RFNET_REFERENCE", "/root/reference")) and (
        force or not os.path.exists(ref)
    ):
        subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build())

    # -- Chamfer ---------------------------------------------------------------
    def nn_distance(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1, i1 = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
        d2, i2 = np.empty((b, m), np.float32), np.empty((b, m), np.int32)
        self.lib.orc_nn_distance(b, n, m, _f(xyz1), _f(xyz2), _f(d1), _i(i1), _f(d2), _i(i2))
        return d1, i1, d2, i2

    def nn_distance_grad(self, xyz1, xyz2, gd1, idx1, gd2, idx2):
        xyz1, xyz2, gd1, gd2 = _f32(xyz1), _f32(xyz2), _f32(gd1), _f32(gd2)
        idx1, idx2 = _i32(idx1), _i32(idx2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.orc_nn_distance_grad(b, n, m, _f(xyz1), _f(xyz2), _f(gd1), _i(idx1), _f(gd2),
                                      _i(idx2), _f(g1), _f(g2))
        return g1, g2

    # -- EMD ---------------------------------------------------------------------
    def default_levels(self):
        lv = np.zeros(16, np.float32)
        k = self.lib.orc_approxmatch_default_levels(_f(lv), 16)
        return lv[:k].copy()

    def approx_match(self, xyz1, xyz2, levels=None):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = np.empty((b, m, n), np.float32)
        lv = self.default_levels() if levels is None else _f32(levels)
        self.lib.orc_approxmatch_levels(b, n, m, _f(xyz1), _f(xyz2), _f(match), _f(lv), len(lv))
        return match

    def match_cost(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = np.empty((b,), np.float32)
        self.lib.orc_matchcost(b, n, m, _f(xyz1), _f(xyz2), _f(match), _f(cost))
        return cost

    def match_cost_grad(self, xyz1, xyz2, match):
        xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.orc_matchcostgrad(b, n, m, _f(xyz1), _f(xyz2), _f(match), _f(g1), _f(g2))
        return g1, g2

    # -- sampling ----------------------------------------------------------------
    def farthest_point_sample(self, npoint, inp):
        inp = _f32(inp)
        b, n, _ = inp.shape
        out = np.zeros((b, npoint), np.int32)
        self.lib.orc_farthest_point_sample(b, n, npoint, _f(inp), _i(out))
        return out

    def gather_point(self, inp, idx):
        inp, idx = _f32(inp), _i32(idx)
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = np.empty((b, m, 3), np.float32)
        self.lib.orc_gather_point(b, n, m, _f(inp), _i(idx), _f(out))
        return out

    def gather_point_grad(self, inp, idx, out_g):
        inp, idx, out_g = _f32(inp), _i32(idx), _f32(out_g)
        b, n, _ = inp.shape
        m = idx.shape[1]
        g = np.empty((b, n, 3), np.float32)
        self.lib.orc_gather_point_grad(b, n, m, _f(out_g), _i(idx), _f(g))
        return g

    # -- grouping ----------------------------------------------------------------
    def query_ball_point(self, radius, nsample, xyz1, xyz2, fill=0):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = np.full((b, m, nsample), fill, np.int32)
        cnt = np.empty((b, m), np.int32)
        self.lib.orc_query_ball_point(b, n, m, C.c_float(radius), nsample, _f(xyz1), _f(xyz2),
                                      _i(idx), _i(cnt))
        return idx, cnt

    def group_point(self, points, idx):
        points, idx = _f32(points), _i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = np.empty((b, m, ns, c), np.float32)
        self.lib.orc_group_point(b, n, c, m, ns, _f(points), _i(idx), _f(out))
        return out

    def group_point_grad(self, points, idx, grad_out):
        points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        g = np.empty((b, n, c), np.float32)
        self.lib.orc_group_point_grad(b, n, c, m, ns, _f(grad_out), _i(idx), _f(g))
        return g

    # -- interpolation -------------------------------------------------------------
    def three_nn(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist, idx = np.empty((b, n, 3), np.float32), np.empty((b, n, 3), np.int32)
        self.lib.orc_three_nn(b, n, m, _f(xyz1), _f(xyz2), _f(dist), _i(idx))
        return dist, idx

    def three_interpolate(self, points, idx, weight):
        points, idx, weight = _f32(points), _i32(idx), _f32(weight)
        b, m, c = points.shape
        n = idx.shape[1]
        out = np.empty((b, n, c), np.float32)
        self.lib.orc_three_interpolate(b, m, c, n, _f(points), _i(idx), _f(weight), _f(out))
        return out

    def three_interpolate_grad(self, points, idx, weight, grad_out):
        points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
        b, m, c = points.shape
        n = idx.shape[1]
        g = np.empty((b, m, c), np.float32)
        self.lib.orc_three_interpolate_grad(b, n, c, m, _f(grad_out), _i(idx), _f(weight), _f(g))
        return g


def _orc_auction(self, xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    assert self.lib.orc_auction_match_supported(n), "n must be < 1024 or one of 1024, 2048, 4096"
    ml, mr = np.empty((b, n), np.int32), np.empty((b, n), np.int32)
    self.lib.orc_auction_match(b, n, _f(xyz1), _f(xyz2), _i(ml), _i(mr))
    return ml, mr


def _orc_select_top_k(self, k, dist):
    dist = _f32(dist)
    b, m, n = dist.shape
    idx, val = np.empty((b, m, n), np.int32), np.empty((b, m, n), np.float32)
    self.lib.orc_selection_sort(b, n, m, int(k), _f(dist), _i(idx), _f(val))
    return idx, val


def _orc_prob_sample(self, inp_p, inp_r):
    inp_p, inp_r = _f32(inp_p), _f32(inp_r)
    b, n = inp_p.shape
    m = inp_r.shape[1]
    temp, out = np.empty((b, n), np.float32), np.empty((b, m), np.int32)
    self.lib.orc_prob_sample(b, n, m, _f(inp_p), _f(inp_r), _f(temp), _i(out))
    return out, temp


Oracle.prob_sample = _orc_prob_sample
Oracle.auction_match = _orc_auction
Oracle.select_top_k = _orc_select_top_k


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref.so"))


class Ref:
    """The reference's own CPU bodies (symbols keep the reference's names)."""

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libref.so"))

    def nn_distance(self, xyz1, xyz2):
        """NnDistanceOp::Compute, tf_ops/CD/tf_nndistance.cpp:79-80 (two nnsearch calls)."""
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1, i1 = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
        d2, i2 = np.empty((b, m), np.float32), np.empty((b, m), np.int32)
        self.lib.nnsearch(b, n, m, _f(xyz1), _f(xyz2), _f(d1), _i(i1))
        self.lib.nnsearch(b, m, n, _f(xyz2), _f(xyz1), _f(d2), _i(i2))
        return d1, i1, d2, i2

    def nn_distance_grad(self, xyz1, xyz2, gd1, idx1, gd2, idx2):
        xyz1, xyz2, gd1, gd2 = _f32(xyz1), _f32(xyz2), _f32(gd1), _f32(gd2)
        idx1, idx2 = _i32(idx1), _i32(idx2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.ref_nn_distance_grad(b, n, m, _f(xyz1), _f(xyz2), _f(gd1), _i(idx1), _f(gd2),
                                      _i(idx2), _f(g1), _f(g2))
        return g1, g2

    def omp_max_threads(self):
        return int(self.lib.ref_omp_max_threads())

    def nn_step_all_cores(self, xyz1, xyz2, gd1=None, gd2=None, threads=0):
        """The same reference bodies (nnsearch x2, then the NnDistanceGrad loop when gd1/gd2 are given), one
        batch element per OpenMP thread (oracle/build_ref.sh: the wrapper loop only, the bodies untouched).
        -> (dist1, idx1, dist2, idx2[, grad1, grad2]), threads used."""
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        threads = int(threads) or self.omp_max_threads()
        d1, i1 = np.empty((b, n), np.float32), np.empty((b, n), np.int32)
        d2, i2 = np.empty((b, m), np.float32), np.empty((b, m), np.int32)
        self.lib.ref_nn_forward_omp(b, n, m, _f(xyz1), _f(xyz2), _f(d1), _i(i1), _f(d2), _i(i2), threads)
        if gd1 is None:
            return (d1, i1, d2, i2), threads
        gd1, gd2 = _f32(gd1), _f32(gd2)
        g1, g2 = np.empty((b, n, 3), np.float32), np.empty((b, m, 3), np.float32)
        self.lib.ref_nn_grad_omp(b, n, m, _f(xyz1), _f(xyz2), _f(gd1), _i(i1), _f(gd2), _i(i2), _f(g1), _f(g2), threads)
        return (d1, i1, d2, i2, g1, g2), threads

    def approxmatch_cpu(self, xyz1, xyz2):
        """Returns the reference CPU layout [b][n][m] (11 levels, double) -- SURVEY T4."""
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        match = np.empty((b, n, m), np.float32)
        self.lib.approxmatch_cpu(b, n, m, _f(xyz1), _f(xyz2), _f(match))
        return match

    def matchcost_cpu(self, xyz1, xyz2, match_nm):
        xyz1, xyz2, match_nm = _f32(xyz1), _f32(xyz2), _f32(match_nm)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        cost = np.empty((b,), np.float32)
        self.lib.matchcost_cpu(b, n, m, _f(xyz1), _f(xyz2), _f(match_nm), _f(cost))
        return cost

    def matchcostgrad_cpu(self, xyz1, xyz2, match_nm):
        """grad1 y/z are unreliable in the reference; grad2 is clean."""
        xyz1, xyz2, match_nm = _f32(xyz1), _f32(xyz2), _f32(match_nm)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.zeros((b, n, 3), np.float32), np.zeros((b, m, 3), np.float32)
        self.lib.matchcostgrad_cpu(b, n, m, _f(xyz1), _f(xyz2), _f(match_nm), _f(g1), _f(g2))
        return g1, g2

    def three_nn(self, xyz1, xyz2):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        dist, idx = np.empty((b, n, 3), np.float32), np.empty((b, n, 3), np.int32)
        self.lib.threenn_cpu(b, n, m, _f(xyz1), _f(xyz2), _f(dist), _i(idx))
        return dist, idx

    def three_interpolate(self, points, idx, weight):
        points, idx, weight = _f32(points), _i32(idx), _f32(weight)
        b, m, c = points.shape
        n = idx.shape[1]
        out = np.empty((b, n, c), np.float32)
        self.lib.threeinterpolate_cpu(b, m, c, n, _f(points), _i(idx), _f(weight), _f(out))
        return out

    def three_interpolate_grad(self, points, idx, weight, grad_out):
        points, idx, weight, grad_out = _f32(points), _i32(idx), _f32(weight), _f32(grad_out)
        b, m, c = points.shape
        n = idx.shape[1]
        g = np.zeros((b, m, c), np.float32)  # the TF op zero-fills (tf_interpolate.cpp:253)
        self.lib.threeinterpolate_grad_cpu(b, n, c, m, _f(grad_out), _i(idx), _f(weight), _f(g))
        return g

    def query_ball_point(self, radius, nsample, xyz1, xyz2, fill=0):
        xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        idx = np.full((b, m, nsample), fill, np.int32)
        r = np.array([radius], np.float32)
        self.lib.query_ball_point_cpu(b, n, m, _f(r), nsample, _f(xyz1), _f(xyz2), _i(idx))
        return idx

    def group_point(self, points, idx):
        points, idx = _f32(points), _i32(idx)
        b, n, c = points.shape
        _, m, ns = idx.shape
        out = np.empty((b, m, ns, c), np.float32)
        self.lib.group_point_cpu(b, n, c, m, ns, _f(points), _i(idx), _f(out))
        return out

    def group_point_grad(self, points, idx, grad_out):
        points, idx, grad_out = _f32(points), _i32(idx), _f32(grad_out)
        b, n, c = points.shape
        _, m, ns = idx.shape
        g = np.zeros((b, n, c), np.float32)
        self.lib.group_point_grad_cpu(b, n, c, m, ns, _f(grad_out), _i(idx), _f(g))
        return g
