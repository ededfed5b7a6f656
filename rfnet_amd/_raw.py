"""Raw (non-differentiable) calls into librfops.so, one per reference OpKernel.

Each function validates shapes with the reference OpKernel's own checks and wording
(`OP_REQUIRES(... errors::InvalidArgument(msg))`, cited per function), allocates outputs and
scratch on the caller's GPU (what TF's allocate_output / allocate_temp did), and enqueues the
HIP kernels on torch's current stream through the C ABI.  No computation happens in Python.
"""
import ctypes as C

import numpy as np
import torch

from . import _host as H
from ._lib import check, lib

F32, I32 = torch.float32, torch.int32


def _shape3(t, last=None):
    return t.dim() == 3 and (last is None or t.shape[2] == last)


# ------------------------------------------------------------------ Chamfer ----------------
NN_MODES = {"auto": 0, "dense": 1, "culled": 2}


@H.on_input_device
def nn_distance(xyz1, xyz2, mode="auto", stats=None):
    """NnDistanceGpuOp::Compute, tf_ops/CD/tf_nndistance.cpp:172-204.

    `mode` pins the sweep ("dense": every pair; "culled": nn_pruned.hip) -- same outputs; `stats`
    (a list) receives the culled sweep's 8 counters (rfops.h)."""
    st = H.Staged()
    a, b_ = st.take(xyz1, F32), st.take(xyz2, F32)
    if a.dim() != 3:
        raise H.invalid("NnDistance requires xyz1 be of shape (batch,#points,3)")
    if a.shape[2] != 3:
        raise H.invalid("NnDistance only accepts 3d point set xyz1")
    if b_.dim() != 3:
        raise H.invalid("NnDistance requires xyz2 be of shape (batch,#points,3)")
    if b_.shape[2] != 3:
        raise H.invalid("NnDistance only accepts 3d point set xyz2")
    if b_.shape[0] != a.shape[0]:
        raise H.invalid("NnDistance expects xyz1 and xyz2 have same batch size")
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    d1, i1 = H.empty((b, n), F32, dev), H.empty((b, n), I32, dev)
    d2, i2 = H.empty((b, m), F32, dev), H.empty((b, m), I32, dev)
    ws, wsz = H.workspace(lib.rf_nn_distance_mode_workspace_bytes(b, n, m, NN_MODES[mode]), dev, "nn")
    cnt = (C.c_ulonglong * 32)() if stats is not None else None
    check(lib.rf_nn_distance_mode(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(d1), H.ptr(i1), H.ptr(d2),
                                  H.ptr(i2), H.ptr(ws), wsz, H.stream(dev), NN_MODES[mode],
                                  C.cast(cnt, C.c_void_p) if cnt is not None else None), "rf_nn_distance")
    if stats is not None:
        stats[:] = list(cnt)
    return tuple(st.give(t) for t in (d1, i1, d2, i2))


@H.on_input_device
def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """NnDistanceGradGpuOp::Compute, tf_ops/CD/tf_nndistance.cpp:216-251."""
    st = H.Staged()
    a, b_ = st.take(xyz1, F32), st.take(xyz2, F32)
    gd1, gd2 = st.take(grad_dist1, F32), st.take(grad_dist2, F32)
    i1, i2 = st.take(idx1, I32), st.take(idx2, I32)
    if a.dim() != 3:
        raise H.invalid("NnDistanceGrad requires xyz1 be of shape (batch,#points,3)")
    if a.shape[2] != 3:
        raise H.invalid("NnDistanceGrad only accepts 3d point set xyz1")
    if b_.dim() != 3:
        raise H.invalid("NnDistanceGrad requires xyz2 be of shape (batch,#points,3)")
    if b_.shape[2] != 3:
        raise H.invalid("NnDistanceGrad only accepts 3d point set xyz2")
    if b_.shape[0] != a.shape[0]:
        raise H.invalid("NnDistanceGrad expects xyz1 and xyz2 have same batch size")
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    if tuple(gd1.shape) != (b, n):
        raise H.invalid("NnDistanceGrad requires grad_dist1 be of shape(batch,#points)")
    if tuple(i1.shape) != (b, n):
        raise H.invalid("NnDistanceGrad requires idx1 be of shape(batch,#points)")
    if tuple(gd2.shape) != (b, m):
        raise H.invalid("NnDistanceGrad requires grad_dist2 be of shape(batch,#points)")
    if tuple(i2.shape) != (b, m):
        raise H.invalid("NnDistanceGrad requires idx2 be of shape(batch,#points)")
    dev = st.device_()
    a, b_, gd1, gd2, i1, i2 = st.up(a, b_, gd1, gd2, i1, i2)
    g1, g2 = H.empty((b, n, 3), F32, dev), H.empty((b, m, 3), F32, dev)
    check(lib.rf_nn_distance_grad(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(gd1), H.ptr(i1), H.ptr(gd2),
                                  H.ptr(i2), H.ptr(g1), H.ptr(g2), H.stream(dev)),
          "rf_nn_distance_grad")
    return st.give(g1), st.give(g2)


def _nn_inputs(st, xyz1, xyz2, op="NnDistance"):
    a, b_ = st.take(xyz1, F32), st.take(xyz2, F32)
    if a.dim() != 3:
        raise H.invalid(f"{op} requires xyz1 be of shape (batch,#points,3)")
    if a.shape[2] != 3:
        raise H.invalid(f"{op} only accepts 3d point set xyz1")
    if b_.dim() != 3:
        raise H.invalid(f"{op} requires xyz2 be of shape (batch,#points,3)")
    if b_.shape[2] != 3:
        raise H.invalid(f"{op} only accepts 3d point set xyz2")
    if b_.shape[0] != a.shape[0]:
        raise H.invalid(f"{op} expects xyz1 and xyz2 have same batch size")
    return a, b_


@H.on_input_device
def nn_distance_dir(xyz1, xyz2, want1=True, want2=True):
    """nn_distance with only the direction(s) the caller uses (rf_nn_distance_dir): the reference's
    glue drops outputs -- merge_layer keeps idx2 (vv_recon.py:134-135), fidelity_loss dist1 (:386-390),
    zero_groupnear dist2 (:415-419).  -> (dist1, idx1, dist2, idx2) with None for a skipped direction."""
    if not (want1 or want2):
        raise H.invalid("nn_distance_dir needs at least one direction")
    st = H.Staged()
    a, b_ = _nn_inputs(st, xyz1, xyz2)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    d1 = H.empty((b, n), F32, dev) if want1 else None
    i1 = H.empty((b, n), I32, dev) if want1 else None
    d2 = H.empty((b, m), F32, dev) if want2 else None
    i2 = H.empty((b, m), I32, dev) if want2 else None
    ws, wsz = H.workspace(lib.rf_nn_distance_dir_workspace_bytes(b, n, m, int(want1), int(want2)), dev, "nn")
    check(lib.rf_nn_distance_dir(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(d1), H.ptr(i1), H.ptr(d2), H.ptr(i2),
                                 H.ptr(ws), wsz, H.stream(dev), int(want1), int(want2)), "rf_nn_distance_dir")
    return tuple(None if t is None else st.give(t) for t in (d1, i1, d2, i2))


class SortedCloud:
    """A batch of clouds in the culled sweep's space-filling-curve order (rf_nn_sort): sort once, use
    in any number of nn_distance_sorted / chamfer_loss / merge_layer calls while `xyz` is unchanged.
    The handle is this object's device buffer; the library itself keeps no state."""

    def __init__(self, xyz):
        st = H.Staged()
        t = st.take(xyz, F32)
        if not _shape3(t, 3):
            raise H.invalid("NnDistance requires xyz1 be of shape (batch,#points,3)")
        dev = st.device_()
        self.xyz, = st.up(t)
        self.b, self.n = int(t.shape[0]), int(t.shape[1])
        nbytes = lib.rf_nn_sort_bytes(self.b, self.n)
        if nbytes == 0:
            raise H.invalid("nn_sort handles clouds of 1..65536 points")
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.rf_nn_sort(self.b, self.n, H.ptr(self.xyz), H.ptr(self.buf), int(nbytes), H.stream(dev)),
                  "rf_nn_sort")

    @property
    def device(self):
        return self.buf.device


def nn_sort(xyz):
    return SortedCloud(xyz)


def nn_distance_sorted(s1, s2, want1=True, want2=True):
    """rf_nn_distance_sorted on two SortedCloud handles -> (dist1, idx1, dist2, idx2), None for a
    skipped direction.  Bit-identical to nn_distance(s1.xyz, s2.xyz)."""
    if s1.b != s2.b:
        raise H.invalid("NnDistance expects xyz1 and xyz2 have same batch size")
    if s1.device != s2.device:
        raise ValueError("all GPU inputs of one op must live on the same device")
    if not (want1 or want2):
        raise H.invalid("nn_distance_sorted needs at least one direction")
    dev, b, n, m = s1.device, s1.b, s1.n, s2.n
    with torch.cuda.device(dev):
        d1 = H.empty((b, n), F32, dev) if want1 else None
        i1 = H.empty((b, n), I32, dev) if want1 else None
        d2 = H.empty((b, m), F32, dev) if want2 else None
        i2 = H.empty((b, m), I32, dev) if want2 else None
        check(lib.rf_nn_distance_sorted(b, n, m, H.ptr(s1.buf), H.ptr(s2.buf), H.ptr(d1), H.ptr(i1), H.ptr(d2),
                                        H.ptr(i2), H.stream(dev)), "rf_nn_distance_sorted")
    return d1, i1, d2, i2


class ChamferStep:
    """nn_distance + nn_distance_grad of one shape as ONE C-ABI call on buffers allocated once
    (rf_chamfer_step): the host path of a training step is a single ctypes crossing with
    precomputed arguments -- no tensor allocation, no shape checks, no Python per-output work."""

    def __init__(self, b, n, m, device):
        dev = torch.device(device)
        self.dev, self.shape = dev, (b, n, m)
        self.dist1, self.idx1 = H.empty((b, n), F32, dev), H.empty((b, n), I32, dev)
        self.dist2, self.idx2 = H.empty((b, m), F32, dev), H.empty((b, m), I32, dev)
        self.grad1, self.grad2 = H.empty((b, n, 3), F32, dev), H.empty((b, m, 3), F32, dev)
        nbytes = int(lib.rf_chamfer_step_workspace_bytes(b, n, m))
        self.ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        self._tail = (H.ptr(self.dist1), H.ptr(self.idx1), H.ptr(self.dist2), H.ptr(self.idx2),
                      H.ptr(self.grad1), H.ptr(self.grad2), H.ptr(self.ws), nbytes)
        self._fn = lib.rf_chamfer_step

    def __call__(self, xyz1, xyz2, grad_dist1, grad_dist2):
        """Inputs: contiguous fp32 GPU tensors of the planned shape (not re-validated: this is the
        hot loop).  Returns (dist1, idx1, dist2, idx2, grad_xyz1, grad_xyz2) -- the plan's own buffers,
        overwritten by the next call."""
        b, n, m = self.shape
        st = self._fn(b, n, m, xyz1.data_ptr(), xyz2.data_ptr(), grad_dist1.data_ptr(), grad_dist2.data_ptr(),
                      *self._tail, torch.cuda.current_stream(self.dev).cuda_stream)
        if st:
            check(st, "rf_chamfer_step")
        return self.dist1, self.idx1, self.dist2, self.idx2, self.grad1, self.grad2


def _sorted_ptr(s, b, n, dev):
    if s is None:
        return None
    if (s.b, s.n) != (b, n) or s.device != dev:
        raise ValueError("sorted handle does not match its cloud (batch, points, device)")
    return H.ptr(s.buf)


@H.on_input_device
def chamfer_loss(xyz1, xyz2, sorted1=None, sorted2=None, want1=True, want2=True):
    """rf_chamfer_loss: per-sample mean sqrt(dist) both ways, (b, 2), next to the nn_distance outputs
    of the computed directions -> (loss, dist1, idx1, dist2, idx2)."""
    if not (want1 or want2):
        raise H.invalid("chamfer_loss needs at least one direction")
    st = H.Staged()
    a, b_ = _nn_inputs(st, xyz1, xyz2)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    loss = H.empty((b, 2), F32, dev)
    d1 = H.empty((b, n), F32, dev) if want1 else None
    i1 = H.empty((b, n), I32, dev) if want1 else None
    d2 = H.empty((b, m), F32, dev) if want2 else None
    i2 = H.empty((b, m), I32, dev) if want2 else None
    p1, p2 = _sorted_ptr(sorted1, b, n, dev), _sorted_ptr(sorted2, b, m, dev)
    ws, wsz = H.workspace(lib.rf_chamfer_loss_workspace_bytes(b, n, m, int(want1), int(want2), int(p1 is not None),
                                                              int(p2 is not None)), dev, "nn")
    check(lib.rf_chamfer_loss(b, n, m, H.ptr(a), H.ptr(b_), p1, p2, H.ptr(loss), H.ptr(d1), H.ptr(i1), H.ptr(d2),
                              H.ptr(i2), H.ptr(ws), wsz, H.stream(dev)), "rf_chamfer_loss")
    return tuple(None if t is None else st.give(t) for t in (loss, d1, i1, d2, i2))


@H.on_input_device
def chamfer_loss_grad(xyz1, xyz2, dist1, idx1, dist2, idx2, grad_loss):
    """rf_chamfer_loss_grad -> (grad_xyz1, grad_xyz2); dist/idx of a direction that was not computed
    are None."""
    st = H.Staged()
    a, b_ = _nn_inputs(st, xyz1, xyz2, "NnDistanceGrad")
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    gl = st.take(grad_loss, F32)
    if tuple(gl.shape) != (b, 2):
        raise H.invalid("chamfer_loss_grad requires grad_loss be of shape (batch,2)")
    opt = []
    for t, dt, shape in ((dist1, F32, (b, n)), (idx1, I32, (b, n)), (dist2, F32, (b, m)), (idx2, I32, (b, m))):
        if t is None:
            opt.append(None)
            continue
        t = st.take(t, dt)
        if tuple(t.shape) != shape:
            raise H.invalid("NnDistanceGrad requires idx/dist be of shape(batch,#points)")
        opt.append(t)
    dev = st.device_()
    a, b_, gl = st.up(a, b_, gl)
    opt = [None if t is None else st.up(t)[0] for t in opt]
    g1, g2 = H.empty((b, n, 3), F32, dev), H.empty((b, m, 3), F32, dev)
    check(lib.rf_chamfer_loss_grad(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(opt[0]), H.ptr(opt[1]), H.ptr(opt[2]),
                                   H.ptr(opt[3]), H.ptr(gl), H.ptr(g1), H.ptr(g2), H.stream(dev)),
          "rf_chamfer_loss_grad")
    return st.give(g1), st.give(g2)


@H.on_input_device
def merge_layer(rawpts, newpts, decfactor, sorted_raw=None):
    """rf_merge_layer (vv_recon.py:132-139) -> (refined (b,m,3), idx2 (b,m))."""
    st = H.Staged()
    a, b_ = _nn_inputs(st, rawpts, newpts)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    dec = torch.as_tensor(decfactor, dtype=F32).detach().reshape(-1)[:1].to(dev).contiguous()
    out, i2 = H.empty((b, m, 3), F32, dev), H.empty((b, m), I32, dev)
    ps = _sorted_ptr(sorted_raw, b, n, dev)
    ws, wsz = H.workspace(lib.rf_merge_layer_workspace_bytes(b, n, m, int(ps is not None)), dev, "nn")
    check(lib.rf_merge_layer(b, n, m, H.ptr(a), H.ptr(b_), ps, H.ptr(dec), H.ptr(out), H.ptr(i2), H.ptr(ws), wsz,
                             H.stream(dev)), "rf_merge_layer")
    return st.give(out), st.give(i2)


@H.on_input_device
def merge_layer_grad(rawpts, newpts, decfactor, idx2, grad_refined, want_raw=False):
    """rf_merge_layer_grad -> (grad_newpts (b,m,3), grad_dec (b,), grad_raw (b,n,3) or None)."""
    st = H.Staged()
    a, b_ = _nn_inputs(st, rawpts, newpts)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    ix, go = st.take(idx2, I32), st.take(grad_refined, F32)
    if tuple(ix.shape) != (b, m) or tuple(go.shape) != (b, m, 3):
        raise H.invalid("merge_layer_grad expects idx2 (batch,#new) and grad (batch,#new,3)")
    dev = st.device_()
    a, b_, ix, go = st.up(a, b_, ix, go)
    dec = torch.as_tensor(decfactor, dtype=F32).detach().reshape(-1)[:1].to(dev).contiguous()
    gn, gd = H.empty((b, m, 3), F32, dev), H.empty((b,), F32, dev)
    gr = H.empty((b, n, 3), F32, dev) if want_raw else None
    check(lib.rf_merge_layer_grad(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(dec), H.ptr(ix), H.ptr(go), H.ptr(gn),
                                  H.ptr(gd), H.ptr(gr), H.stream(dev)), "rf_merge_layer_grad")
    return st.give(gn), st.give(gd), (None if gr is None else st.give(gr))


# ------------------------------------------------------------------ EMD --------------------
def _emd_inputs(st, xyz1, xyz2, opname):
    a, b_ = st.take(xyz1, F32), st.take(xyz2, F32)
    if not _shape3(a, 3):
        raise H.invalid(f"{opname} expects (batch_size,num_points,3) xyz1 shape")
    if not (_shape3(b_, 3) and b_.shape[0] == a.shape[0]):
        raise H.invalid(f"{opname} expects (batch_size,num_points,3) xyz2 shape, and batch_size must match")
    return a, b_


# rf_approxmatch_mode / rf_earth_mover_mode (include/rfops.h): "swept" pins the route -- a sample's bits do not depend on the batch
EMD_MODES = {"auto": 0, "swept": 1}


@H.on_input_device
def approx_match(xyz1, xyz2, levels=None, mode="auto"):
    """ApproxMatchGpuOp::Compute, pc_distance/tf_approxmatch.cpp:148-172 -> match (b,m,n).

    `levels` (optional, extension): explicit annealing schedule; default = the reference's 10.
    `mode` (extension): "auto" | "swept" (batch-invariant bits per sample, as the reference's per-sample loop).
    """
    if mode not in EMD_MODES:
        raise H.invalid(f"ApproxMatch: mode must be one of {sorted(EMD_MODES)}")
    st = H.Staged()
    a, b_ = _emd_inputs(st, xyz1, xyz2, "ApproxMatch")
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    match = H.empty((b, m, n), F32, dev)
    nlv = 0 if levels is None else len(levels)
    if mode != "auto":
        ws, wsz = H.workspace(lib.rf_approxmatch_mode_workspace_bytes(b, n, m, nlv, EMD_MODES[mode]), dev, "am")
        lv = None if levels is None else (C.c_float * nlv)(*[float(v) for v in levels])
        check(lib.rf_approxmatch_mode(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(match), lv, nlv, H.ptr(ws), wsz, H.stream(dev),
                                      EMD_MODES[mode]), "rf_approxmatch_mode")
        return st.give(match)
    ws, wsz = H.workspace(lib.rf_approxmatch_workspace_bytes(b, n, m, nlv), dev, "am")
    if levels is None:
        check(lib.rf_approxmatch(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(match), H.ptr(ws), wsz,
                                 H.stream(dev)), "rf_approxmatch")
    else:
        lv = (C.c_float * nlv)(*[float(v) for v in levels])
        check(lib.rf_approxmatch_levels(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(match), lv, nlv,
                                        H.ptr(ws), wsz, H.stream(dev)), "rf_approxmatch_levels")
    return st.give(match)


def _match_check(mt, b, n, m):
    if not (mt.dim() == 3 and mt.shape[0] == b and mt.shape[1] == m and mt.shape[2] == n):
        raise H.invalid("MatchCost expects (batch_size,#query,#dataset) match shape")


@H.on_input_device
def match_cost(xyz1, xyz2, match):
    """MatchCostGpuOp::Compute, pc_distance/tf_approxmatch.cpp:204-230 -> cost (b)."""
    st = H.Staged()
    a, b_ = _emd_inputs(st, xyz1, xyz2, "MatchCost")
    mt = st.take(match, F32)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    _match_check(mt, b, n, m)
    dev = st.device_()
    a, b_, mt = st.up(a, b_, mt)
    cost = H.empty((b,), F32, dev)
    ws, wsz = H.workspace(lib.rf_matchcost_workspace_bytes(b, n, m), dev, "mc")
    check(lib.rf_matchcost(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(mt), H.ptr(cost), H.ptr(ws), wsz,
                           H.stream(dev)), "rf_matchcost")
    return st.give(cost)


@H.on_input_device
def match_cost_grad(xyz1, xyz2, match):
    """MatchCostGradGpuOp::Compute, pc_distance/tf_approxmatch.cpp:265-295."""
    st = H.Staged()
    a, b_ = _emd_inputs(st, xyz1, xyz2, "MatchCostGrad")
    mt = st.take(match, F32)
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    _match_check(mt, b, n, m)
    dev = st.device_()
    a, b_, mt = st.up(a, b_, mt)
    g1, g2 = H.empty((b, n, 3), F32, dev), H.empty((b, m, 3), F32, dev)
    check(lib.rf_matchcost_grad(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(mt), H.ptr(g1), H.ptr(g2),
                                H.stream(dev)), "rf_matchcost_grad")
    return st.give(g1), st.give(g2)


@H.on_input_device
def earth_mover(xyz1, xyz2, with_grad=False, mode="auto"):
    """Row f1: the fused form of `earth_mover`'s op chain (vv_recon.py:392-399):
    approx_match -> match_cost [-> MatchCostGrad], without materialising match.
    -> cost (b)  or  (cost, grad1 (b,n,3), grad2 (b,m,3)) with `with_grad`.
    mode="swept": cost[i] bit-identical whatever the batch around sample i (rf_earth_mover_mode)."""
    if mode not in EMD_MODES:
        raise H.invalid(f"ApproxMatch: mode must be one of {sorted(EMD_MODES)}")
    st = H.Staged()
    a, b_ = _emd_inputs(st, xyz1, xyz2, "ApproxMatch")
    b, n, m = a.shape[0], a.shape[1], b_.shape[1]
    dev = st.device_()
    a, b_ = st.up(a, b_)
    cost = H.empty((b,), F32, dev)
    g1 = H.empty((b, n, 3), F32, dev) if with_grad else None
    g2 = H.empty((b, m, 3), F32, dev) if with_grad else None
    ws, wsz = H.workspace(lib.rf_earth_mover_mode_workspace_bytes(b, n, m, EMD_MODES[mode]), dev, "emd")
    check(lib.rf_earth_mover_mode(b, n, m, H.ptr(a), H.ptr(b_), H.ptr(cost),
                                  H.ptr(g1) if with_grad else None, H.ptr(g2) if with_grad else None,
                                  H.ptr(ws), wsz, H.stream(dev), EMD_MODES[mode]), "rf_earth_mover_mode")
    if with_grad:
        return st.give(cost), st.give(g1), st.give(g2)
    return st.give(cost)


# ------------------------------------------------------------------ sampling ---------------
@H.on_input_device
def farthest_point_sample(npoint, inp, _pin_reg=False):
    """FarthestPointSampleGpuOp, tf_ops/sampling/tf_sampling.cpp:95-123 -> (b,npoint) int32.
    (_pin_reg: the kernels of the unsorted cloud whatever the size -- rf_farthestpointsampling; a measurement aid.)"""
    npoint = int(npoint)
    if npoint <= 0:
        raise H.invalid("FarthestPointSample expects positive npoint")
    st = H.Staged()
    p = st.take(inp, F32)
    if not _shape3(p, 3):
        raise H.invalid("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n = p.shape[0], p.shape[1]
    dev = st.device_()
    p, = st.up(p)
    out = H.empty((b, npoint), I32, dev)
    if _pin_reg:
        nt = lib.rf_farthestpointsampling_temp_floats(b, n)
        temp = H.empty((nt,), F32, dev) if nt else None
        check(lib.rf_farthestpointsampling(b, n, npoint, H.ptr(p), H.ptr(temp), H.ptr(out),
                                           H.stream(dev)), "rf_farthestpointsampling")
        return st.give(out)
    ws, wsz = H.workspace(lib.rf_farthestpointsampling_workspace_bytes(b, n, npoint), dev, "fps")
    check(lib.rf_farthestpointsampling_ws(b, n, npoint, H.ptr(p), H.ptr(ws), wsz, H.ptr(out), H.stream(dev)),
          "rf_farthestpointsampling_ws")
    return st.give(out)


def farthest_point_sample_reg(npoint, inp):
    """farthest_point_sample pinned to the register-resident kernel of the unsorted cloud (measurement aid)."""
    return farthest_point_sample(npoint, inp, _pin_reg=True)


@H.on_input_device
def farthest_point_sample_sorted(npoint, inp, form=0, with_xyz=False):
    """farthest_point_sample over the spatially sorted cloud (sampling.hip fps_sorted_kernel): the same indices, for clouds of
    1025..16384 points.  -> idx (b, npoint) [, new_xyz (b, npoint, 3)]"""
    npoint = int(npoint)
    if npoint <= 0:
        raise H.invalid("FarthestPointSample expects positive npoint")
    st = H.Staged()
    p = st.take(inp, F32)
    if not _shape3(p, 3):
        raise H.invalid("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n = p.shape[0], p.shape[1]
    dev = st.device_()
    p, = st.up(p)
    out = H.empty((b, npoint), I32, dev)
    nx = H.empty((b, npoint, 3), F32, dev) if with_xyz else None
    ws, wsz = H.workspace(lib.rf_farthestpointsampling_sorted_workspace_bytes(b, n), dev, "fps")
    check(lib.rf_farthestpointsampling_sorted(b, n, npoint, int(form), H.ptr(p), H.ptr(ws), wsz, H.ptr(out), H.ptr(nx),
                                              H.stream(dev)), "rf_farthestpointsampling_sorted")
    if with_xyz:
        return st.give(out), st.give(nx)
    return st.give(out)


@H.on_input_device
def gather_point(inp, idx):
    """GatherPointGpuOp, tf_sampling.cpp:126-148 -> (b,m,3)."""
    st = H.Staged()
    p, ix = st.take(inp, F32), st.take(idx, I32)
    if not _shape3(p, 3):
        raise H.invalid("GatherPoint expects (batch_size,num_points,3) inp shape")
    if not (ix.dim() == 2 and ix.shape[0] == p.shape[0]):
        raise H.invalid("GatherPoint expects (batch_size,num_result) idx shape")
    b, n, m = p.shape[0], p.shape[1], ix.shape[1]
    dev = st.device_()
    p, ix = st.up(p, ix)
    out = H.empty((b, m, 3), F32, dev)
    check(lib.rf_gatherpoint(b, n, m, H.ptr(p), H.ptr(ix), H.ptr(out), H.stream(dev)),
          "rf_gatherpoint")
    return st.give(out)


@H.on_input_device
def gather_point_grad(inp, idx, out_g):
    """GatherPointGradGpuOp, tf_sampling.cpp:151-178 -> (b,n,3)."""
    st = H.Staged()
    p, ix, og = st.take(inp, F32), st.take(idx, I32), st.take(out_g, F32)
    if not _shape3(p, 3):
        raise H.invalid("GatherPointGradGpuOp expects (batch_size,num_points,3) inp")
    if not (ix.dim() == 2 and ix.shape[0] == p.shape[0]):
        raise H.invalid("GatherPointGradGpuOp expects (batch_size,num_result) idx shape")
    b, n, m = p.shape[0], p.shape[1], ix.shape[1]
    if tuple(og.shape) != (b, m, 3):
        raise H.invalid("GatherPointGradGpuOp expects (batch_size,num_result,3) out_g shape")
    dev = st.device_()
    p, ix, og = st.up(p, ix, og)
    g = H.empty((b, n, 3), F32, dev)
    check(lib.rf_scatteraddpoint(b, n, m, H.ptr(og), H.ptr(ix), H.ptr(g), H.stream(dev)),
          "rf_scatteraddpoint")
    return st.give(g)


# ------------------------------------------------------------------ grouping ---------------
QB_BOXES_MIN_N = 2048  # datasets from this size on take the boxed kernel (its sort costs ~20 us whatever the size)


@H.on_input_device
def query_ball_point(radius, nsample, xyz1, xyz2, sorted1=None, form="auto"):
    """QueryBallPointGpuOp, tf_ops/grouping/tf_grouping.cpp:68-110 -> idx (b,m,nsample), pts_cnt (b,m).

    Rows whose ball is empty are not written by the kernel (as in the reference, whose output
    buffer is then uninitialised); this wrapper allocates idx zero-filled so they read 0.
    form: "auto" (the boxed kernel for datasets of QB_BOXES_MIN_N points and more, the scan below that), "boxes",
    "scan" -- same results; sorted1: an rf_nn_sort handle of xyz1 (nn_sort), skips the boxed form's own sort.
    """
    nsample = int(nsample)
    if nsample <= 0:
        raise H.invalid("QueryBallPoint expects positive nsample")
    st = H.Staged()
    d, q = st.take(xyz1, F32), st.take(xyz2, F32)
    if not _shape3(d, 3):
        raise H.invalid("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    if not _shape3(q, 3):
        raise H.invalid("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    b, n, m = d.shape[0], d.shape[1], q.shape[1]
    dev = st.device_()
    d, q = st.up(d, q)
    idx = H.zeros((b, m, nsample), I32, dev)
    cnt = H.empty((b, m), I32, dev)
    rt = None
    if isinstance(radius, torch.Tensor) and radius.is_cuda:
        # the reference's form: radius is an op input tensor on the device (tf_grouping.cpp:18,93-95)
        if radius.device != dev:
            raise ValueError(f"all GPU inputs of one op must live on the same device: got {dev} and {radius.device}")
        rt = radius.detach().reshape(-1)[:1].to(F32).contiguous()
        r = 0.0
    else:
        r = float(np.float32(float(radius)))
    wsz = lib.rf_queryballpoint_boxes_workspace_bytes(b, n) if nsample <= 64 and b <= 65535 else 0
    boxes = form == "boxes" or (form == "auto" and n >= QB_BOXES_MIN_N)
    if form == "boxes" and not wsz:
        raise H.invalid("QueryBallPoint: the boxed form takes datasets of 64..65536 points and nsample <= 64")
    if boxes and wsz and b * m > 0:
        ws = H.empty((wsz // 4,), F32, dev)
        check(lib.rf_queryballpoint_boxes(b, n, m, r, H.ptr(rt), nsample, H.ptr(d), H.ptr(q),
                                          H.ptr(sorted1) if sorted1 is not None else None, H.ptr(idx), H.ptr(cnt),
                                          H.ptr(ws), wsz, H.stream(dev)), "rf_queryballpoint_boxes")
    elif rt is not None:
        check(lib.rf_queryballpoint_dev(b, n, m, H.ptr(rt), nsample, H.ptr(d), H.ptr(q), H.ptr(idx),
                                        H.ptr(cnt), H.stream(dev)), "rf_queryballpoint_dev")
    else:
        check(lib.rf_queryballpoint(b, n, m, r, nsample, H.ptr(d), H.ptr(q), H.ptr(idx), H.ptr(cnt),
                                    H.stream(dev)), "rf_queryballpoint")
    return st.give(idx), st.give(cnt)


@H.on_input_device
def sample_and_group(npoint, radius, nsample, xyz, aux_stream=None):
    """rf_sample_and_group: farthest_point_sample -> gather_point -> query_ball_point -> group_point(xyz) as one call
    -> (fps_idx (b,npoint), new_xyz (b,npoint,3), idx (b,npoint,nsample), pts_cnt (b,npoint), grouped_xyz (b,npoint,nsample,3)),
    bit-identical to the four ops.  aux_stream: a torch.cuda.Stream on which the dataset's sort runs beside FPS."""
    npoint, nsample = int(npoint), int(nsample)
    if npoint <= 0:
        raise H.invalid("FarthestPointSample expects positive npoint")
    if nsample <= 0:
        raise H.invalid("QueryBallPoint expects positive nsample")
    st = H.Staged()
    p = st.take(xyz, F32)
    if not _shape3(p, 3):
        raise H.invalid("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n = p.shape[0], p.shape[1]
    dev = st.device_()
    p, = st.up(p)
    wsz = lib.rf_sample_and_group_workspace_bytes(b, n) if nsample <= 64 and b <= 65535 else 0
    if not wsz:
        raise H.invalid("sample_and_group takes clouds of 64..65536 points and nsample <= 64")
    rt = None
    if isinstance(radius, torch.Tensor) and radius.is_cuda:
        rt = radius.detach().reshape(-1)[:1].to(F32).contiguous()
        r = 0.0
    else:
        r = float(np.float32(float(radius)))
    fi = H.empty((b, npoint), I32, dev)
    nx = H.empty((b, npoint, 3), F32, dev)
    gi = H.empty((b, npoint, nsample), I32, dev)
    cnt = H.empty((b, npoint), I32, dev)
    gx = H.empty((b, npoint, nsample, 3), F32, dev)
    ws = H.empty((wsz // 4,), F32, dev)
    check(lib.rf_sample_and_group(b, n, npoint, r, H.ptr(rt), nsample, H.ptr(p), H.ptr(fi), H.ptr(nx), H.ptr(gi), H.ptr(cnt),
                                  H.ptr(gx), H.ptr(ws), wsz, H.stream(dev),
                                  aux_stream.cuda_stream if aux_stream is not None else None), "rf_sample_and_group")
    if aux_stream is not None:
        ws.record_stream(aux_stream)  # the sort wrote the workspace on that stream
    return st.give(fi), st.give(nx), st.give(gi), st.give(cnt), st.give(gx)


@H.on_input_device
def group_point(points, idx):
    """GroupPointGpuOp, tf_grouping.cpp:147-175 -> (b,m,nsample,c)."""
    st = H.Staged()
    p, ix = st.take(points, F32), st.take(idx, I32)
    if p.dim() != 3:
        raise H.invalid("GroupPoint expects (batch_size, num_points, channel) points shape")
    if not (ix.dim() == 3 and ix.shape[0] == p.shape[0]):
        raise H.invalid("GroupPoint expects (batch_size, npoints, nsample) idx shape")
    b, n, c = p.shape
    m, ns = ix.shape[1], ix.shape[2]
    dev = st.device_()
    p, ix = st.up(p, ix)
    out = H.empty((b, m, ns, c), F32, dev)
    check(lib.rf_grouppoint(b, n, c, m, ns, H.ptr(p), H.ptr(ix), H.ptr(out), H.stream(dev)),
          "rf_grouppoint")
    return st.give(out)


@H.on_input_device
def group_point_grad(points, idx, grad_out, form="auto"):
    """GroupPointGradGpuOp, tf_grouping.cpp:178-212 -> (b,n,c).
    form: "auto" (rf_grouppoint_grad_ws: sorted slots where they pay) | "atomic" (the reference's form, rf_grouppoint_grad)."""
    st = H.Staged()
    p, ix, go = st.take(points, F32), st.take(idx, I32), st.take(grad_out, F32)
    if p.dim() != 3:
        raise H.invalid("GroupPointGrad expects (batch_size, num_points, channel) points shape")
    if not (ix.dim() == 3 and ix.shape[0] == p.shape[0]):
        raise H.invalid("GroupPointGrad expects (batch_size, npoints, nsample) idx shape")
    b, n, c = p.shape
    m, ns = ix.shape[1], ix.shape[2]
    if tuple(go.shape) != (b, m, ns, c):
        raise H.invalid("GroupPointGrad expects (batch_size, npoints, nsample, channel) grad_out shape")
    dev = st.device_()
    p, ix, go = st.up(p, ix, go)
    g = H.empty((b, n, c), F32, dev)
    if form == "atomic":
        check(lib.rf_grouppoint_grad(b, n, c, m, ns, H.ptr(go), H.ptr(ix), H.ptr(g), H.stream(dev)),
              "rf_grouppoint_grad")
        return st.give(g)
    ws, wsz = H.workspace(lib.rf_grouppoint_grad_workspace_bytes(b, n, c, m, ns), dev, "gpg")
    check(lib.rf_grouppoint_grad_ws(b, n, c, m, ns, H.ptr(go), H.ptr(ix), H.ptr(g), H.ptr(ws), wsz, H.stream(dev)),
          "rf_grouppoint_grad_ws")
    return st.give(g)


# ------------------------------------------------------------------ interpolation ----------
# three_nn: pairs per call from which sorting both sets is repaid (tools/ab_three_nn.py: the sort is ~25 us for sets of up to
# 16384 points -- one workgroup per cloud with the points in registers -- and ~270 us beyond)
TN_BOXES_MIN_PAIRS = 100_000_000
TN_BOXES_MIN_PAIRS_LARGE = 4_000_000_000
TN_BOXES_MIN_KNOWN = 512  # fewer known points: a handful of blocks, nothing to skip


@H.on_input_device
def three_nn(xyz1, xyz2, form="auto", sorted1=None, sorted2=None):
    """ThreeNNOp, tf_ops/interpolation/tf_interpolate.cpp:157-187 -> dist (b,n,3), idx (b,n,3).

    form: "auto" (the boxed kernel over sorted copies of the two sets from TN_BOXES_MIN_PAIRS pairs per call on, the scan
    below that), "boxes", "scan" -- same results, ties included; sorted1 / sorted2: rf_nn_sort handles of xyz1 / xyz2
    (nn_sort), which skip the boxed form's own sort of that set."""
    st = H.Staged()
    u, k = st.take(xyz1, F32), st.take(xyz2, F32)
    if not _shape3(u, 3):
        raise H.invalid("ThreeNN expects (b,n,3) xyz1 shape.")
    if not _shape3(k, 3):
        raise H.invalid("ThreeNN expects (b,m,3) xyz2 shape.")
    b, n, m = u.shape[0], u.shape[1], k.shape[1]
    dev = st.device_()
    u, k = st.up(u, k)
    dist, idx = H.empty((b, n, 3), F32, dev), H.empty((b, n, 3), I32, dev)
    wsz = lib.rf_threenn_boxes_workspace_bytes(b, n, m) if b * n * m > 0 else 0
    need = TN_BOXES_MIN_PAIRS if max(n if sorted1 is None else 0, m if sorted2 is None else 0) <= 16384 else TN_BOXES_MIN_PAIRS_LARGE
    if sorted1 is not None and sorted2 is not None:
        # nothing to sort: the boxed kernel alone wins from much smaller calls on (32 x 2048 x 512: 0.025 against 0.033 ms)
        boxes = form == "boxes" or (form == "auto" and n >= 1024 and m >= 256 and b * n * m >= TN_BOXES_MIN_PAIRS // 4)
    else:
        boxes = form == "boxes" or (form == "auto" and n >= 1024 and m >= TN_BOXES_MIN_KNOWN and
                                    (b * n * m >= need or (m >= 1536 and n >= 4096 and need == TN_BOXES_MIN_PAIRS)))
    # (the second clause: with few waves on the chip the scan is a chain of m candidates at ~0.05 us each whatever b -- 1 x 16384 x 2048:
    # 0.103 ms against 0.045 boxed)
    if form == "boxes" and not wsz:
        raise H.invalid("ThreeNN: the boxed form takes sets of 1..65536 points")
    if boxes and wsz:
        ws = H.empty((wsz // 4,), F32, dev)
        check(lib.rf_threenn_boxes(b, n, m, H.ptr(u), H.ptr(k), H.ptr(sorted1) if sorted1 is not None else None,
                                   H.ptr(sorted2) if sorted2 is not None else None, H.ptr(dist), H.ptr(idx), H.ptr(ws), wsz,
                                   H.stream(dev)), "rf_threenn_boxes")
    else:
        check(lib.rf_threenn(b, n, m, H.ptr(u), H.ptr(k), H.ptr(dist), H.ptr(idx), H.stream(dev)),
              "rf_threenn")
    return st.give(dist), st.give(idx)


@H.on_input_device
def three_interpolate(points, idx, weight):
    """ThreeInterpolateOp, tf_interpolate.cpp:191-222 -> (b,n,c)."""
    st = H.Staged()
    p, ix, w = st.take(points, F32), st.take(idx, I32), st.take(weight, F32)
    if p.dim() != 3:
        raise H.invalid("ThreeInterpolate expects (b,m,c) points shape")
    b, m, c = p.shape
    if not (ix.dim() == 3 and ix.shape[0] == b and ix.shape[2] == 3):
        raise H.invalid("ThreeInterpolate expects (b,n,3) idx shape")
    n = ix.shape[1]
    if tuple(w.shape) != (b, n, 3):
        raise H.invalid("ThreeInterpolate expects (b,n,3) weight shape")
    dev = st.device_()
    p, ix, w = st.up(p, ix, w)
    out = H.empty((b, n, c), F32, dev)
    check(lib.rf_threeinterpolate(b, m, c, n, H.ptr(p), H.ptr(ix), H.ptr(w), H.ptr(out),
                                  H.stream(dev)), "rf_threeinterpolate")
    return st.give(out)


@H.on_input_device
def three_interpolate_grad(points, idx, weight, grad_out, form="auto"):
    """ThreeInterpolateGradOp, tf_interpolate.cpp:225-262 -> (b,m,c).
    form: "auto" (rf_threeinterpolate_grad_ws) | "inline" (rf_threeinterpolate_grad: the LDS tile, or atomics beyond it)."""
    st = H.Staged()
    p, ix, w, go = (st.take(points, F32), st.take(idx, I32), st.take(weight, F32),
                    st.take(grad_out, F32))
    if p.dim() != 3:
        raise H.invalid("ThreeInterpolateGrad expects (b,m,c) points shape")
    b, m, c = p.shape
    if not (ix.dim() == 3 and ix.shape[0] == b):
        raise H.invalid("ThreeInterpolateGrad expects (b,n,3) idx shape")
    n = ix.shape[1]
    if tuple(w.shape) != (b, n, 3):
        raise H.invalid("ThreeInterpolateGrad expects (b,n,3) weight shape")
    if tuple(go.shape) != (b, n, c):
        raise H.invalid("ThreeInterpolateGrad expects (b,n,c) grad_out shape")
    dev = st.device_()
    p, ix, w, go = st.up(p, ix, w, go)
    g = H.empty((b, m, c), F32, dev)
    if form == "inline":
        check(lib.rf_threeinterpolate_grad(b, n, c, m, H.ptr(go), H.ptr(ix), H.ptr(w), H.ptr(g),
                                           H.stream(dev)), "rf_threeinterpolate_grad")
        return st.give(g)
    ws, wsz = H.workspace(lib.rf_threeinterpolate_grad_workspace_bytes(b, n, c, m), dev, "tig")
    check(lib.rf_threeinterpolate_grad_ws(b, n, c, m, H.ptr(go), H.ptr(ix), H.ptr(w), H.ptr(g), H.ptr(ws), wsz,
                                          H.stream(dev)), "rf_threeinterpolate_grad_ws")
    return st.give(g)


# ------------------------------------------------------------------ import surface (f3) -----
@H.on_input_device
def auction_match(xyz1, xyz2):
    """AuctionMatchGpuOp, tf_ops/emd/tf_auctionmatch.cpp:27-58 -> matchl (b,n), matchr (b,n) int32."""
    st = H.Staged()
    a, b_ = st.take(xyz1, F32), st.take(xyz2, F32)
    if not _shape3(a, 3):
        raise H.invalid("ApproxMatch expects (batch_size,num_points,3) xyz1 shape")  # sic (reference wording)
    b, n = a.shape[0], a.shape[1]
    if n > 4096:
        raise H.invalid("AuctionMatch handles at most 4096 dataset points")
    if not (_shape3(b_, 3) and b_.shape[0] == b and b_.shape[1] == n):
        raise H.invalid("AuctionMatch expects (batch_size,num_points,3) xyz2 shape, and shape must match with xyz1")
    if n and not lib.rf_auctionmatch_supported(n):
        raise H.invalid("AuctionMatch is defined for n < 1024 or n in {1024, 2048, 4096}: for other n "
                        "the reference kernel reads out of bounds (tf_auctionmatch_g.cu:148,185)")
    dev = st.device_()
    a, b_ = st.up(a, b_)
    ml, mr = H.empty((b, n), I32, dev), H.empty((b, n), I32, dev)
    ws, wsz = H.workspace(lib.rf_auctionmatch_workspace_bytes(b, n), dev, "auction")
    check(lib.rf_auctionmatch(b, n, H.ptr(a), H.ptr(b_), H.ptr(ml), H.ptr(mr), H.ptr(ws), wsz,
                              H.stream(dev)), "rf_auctionmatch")
    return st.give(ml), st.give(mr)


@H.on_input_device
def select_top_k(k, dist):
    """SelectionSortGpuOp, tf_ops/grouping/tf_grouping.cpp:113-143 -> idx (b,m,n) int32, dist_out (b,m,n)."""
    k = int(k)
    if k <= 0:
        raise H.invalid("SelectionSort expects positive k")
    st = H.Staged()
    d = st.take(dist, F32)
    if d.dim() != 3:
        raise H.invalid("SelectionSort expects (b,m,n) dist shape.")
    b, m, n = d.shape
    if n > 16384:
        raise H.invalid("select_top_k handles rows of at most 16384 entries")
    dev = st.device_()
    d, = st.up(d)
    idx, out = H.empty((b, m, n), I32, dev), H.empty((b, m, n), F32, dev)
    check(lib.rf_selectionsort(b, n, m, k, H.ptr(d), H.ptr(idx), H.ptr(out), H.stream(dev)),
          "rf_selectionsort")
    return st.give(idx), st.give(out)


@H.on_input_device
def prob_sample(inp, inpr, return_cumsum=False):
    """ProbSampleGpuOp, tf_ops/sampling/tf_sampling.cpp:66-92 -> (b,m) int32 (and, on request, the
    op's temp tensor: the (b,n) cumulative sums)."""
    st = H.Staged()
    p, r = st.take(inp, F32), st.take(inpr, F32)
    if p.dim() != 2:
        raise H.invalid("ProbSample expects (batch_size,num_choices) inp shape")
    if not (r.dim() == 2 and r.shape[0] == p.shape[0]):
        raise H.invalid("ProbSample expects (batch_size,num_points) inpr shape")
    b, n, m = p.shape[0], p.shape[1], r.shape[1]
    dev = st.device_()
    p, r = st.up(p, r)
    temp, out = H.empty((b, n), F32, dev), H.empty((b, m), I32, dev)
    check(lib.rf_probsample(b, n, m, H.ptr(p), H.ptr(r), H.ptr(temp), H.ptr(out), H.stream(dev)),
          "rf_probsample")
    if return_cumsum:
        return st.give(out), st.give(temp)
    return st.give(out)


# ------------------------------------------------------------------ model graph helper (f2) --
_ACT = {None: 0, "none": 0, "relu": 1, "tanh": 2}


def point_affine(y, p, w, r, act="relu", out=None):
    """rf_point_affine: act(y + p @ w + r) in one pass.  y (b,n,c) or None, p (b,n,kp<=16) or None with
    w (kp,c), r (b,c) / (b,1,c) per sample or (c,) shared.  GPU tensors only (a model-graph helper,
    not one of the reference's ops)."""
    ref = y if y is not None else p
    b, n = ref.shape[0], ref.shape[1]
    c = r.shape[-1]
    kp = 0 if p is None else p.shape[-1]
    if not lib.rf_point_affine_supported(c, kp):
        raise H.invalid("point_affine: channels must be a multiple of 4 (<= 1024), narrow input <= 16 channels")
    dev = ref.device
    per_sample = r.dim() > 1 and r.shape[0] == b and r.numel() == b * c
    if not per_sample and r.numel() != c:
        raise H.invalid("point_affine: r must be (b,c), (b,1,c) or (c,)")
    y_ = None if y is None else y.contiguous()
    p_ = None if p is None else p.contiguous()
    w_ = None if p is None else w.contiguous()
    r_ = r.contiguous()
    if out is None:
        out = H.empty((b, n, c), F32, dev)
    with torch.cuda.device(dev):
        check(lib.rf_point_affine(b, n, c, H.ptr(y_), H.ptr(p_), kp, H.ptr(w_), H.ptr(r_), int(per_sample),
                                  _ACT[act], H.ptr(out), H.stream(dev)), "rf_point_affine")
    return out


def maxpool_points_idx(x):
    """rf_maxpool_points_idx: max over the points axis of a (b, n, c) GPU tensor -> values (b, 1, c) and
    the point index of each maximum (b, c) int32 (lowest index among ties)."""
    b, n, c = x.shape
    dev = x.device
    x_ = x.contiguous()
    out, idx = H.empty((b, 1, c), F32, dev), H.empty((b, c), I32, dev)
    with torch.cuda.device(dev):
        ws, wsz = H.workspace(lib.rf_maxpool_points_idx_workspace_bytes(b, n, c), dev, "maxpool")
        check(lib.rf_maxpool_points_idx(b, n, c, H.ptr(x_), H.ptr(out), H.ptr(idx), H.ptr(ws), wsz, H.stream(dev)),
              "rf_maxpool_points_idx")
    return out, idx


_ACT_GRAD = {None: 0, "none": 0, "relu": 1, "tanh": 2, "leaky_relu": 3}


def act_grad_colsum_supported(c):
    return c > 0 and c % 4 == 0 and c <= 1024


def act_grad_colsum(grad, out, act, inplace=False):
    """rf_act_grad_colsum: g = grad * act'(out) and sums[i,:] = sum_j g[i,j,:] for (b, n, c) GPU tensors,
    one pass.  `out` is the layer's OUTPUT (None with act None).  Returns (g (b,n,c), sums (b,c));
    with act None g is `grad` itself.  inplace: g overwrites grad."""
    b, n, c = grad.shape
    dev = grad.device
    a = _ACT_GRAD[act]
    grad_ = grad.contiguous()
    out_ = None if a == 0 else out.contiguous()
    g = grad_ if (a == 0 or inplace) else H.empty((b, n, c), F32, dev)
    sums = H.empty((b, c), F32, dev)
    with torch.cuda.device(dev):
        ws, wsz = H.workspace(lib.rf_act_grad_colsum_workspace_bytes(b, n, c), dev, "maxpool")
        check(lib.rf_act_grad_colsum(b, n, c, H.ptr(grad_), H.ptr(out_), a, H.ptr(g), H.ptr(sums), H.ptr(ws), wsz,
                                     H.stream(dev)), "rf_act_grad_colsum")
    return g, sums


def maxpool_points(x):
    """rf_maxpool_points: max over the points axis of a (b, n, c) GPU tensor -> (b, 1, c) (keepdim)."""
    b, n, c = x.shape
    dev = x.device
    x_ = x.contiguous()
    out = H.empty((b, 1, c), F32, dev)
    with torch.cuda.device(dev):
        ws, wsz = H.workspace(lib.rf_maxpool_points_workspace_bytes(b, n, c), dev, "maxpool")
        check(lib.rf_maxpool_points(b, n, c, H.ptr(x_), H.ptr(out), H.ptr(ws), wsz, H.stream(dev)), "rf_maxpool_points")
    return out
