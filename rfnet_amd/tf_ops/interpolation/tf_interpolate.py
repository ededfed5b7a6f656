"""3-nearest-neighbour interpolation: drop-in for the reference module
tf_ops/interpolation/tf_interpolate.py (three_nn :8-17, three_interpolate :19-28 + gradient
:29-34).  CPU-only ops in the reference; HIP kernels here."""
import torch

from ... import _raw


def three_nn(xyz1, xyz2):
    '''
    Input:
        xyz1: (b,n,3) float32 array, unknown points
        xyz2: (b,m,3) float32 array, known points
    Output:
        dist: (b,n,3) float32 array, SQUARED distances to the 3 nearest known points
        idx: (b,n,3) int32 array, indices to known points
    '''
    return _raw.three_nn(xyz1, xyz2)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        ctx.save_for_backward(points, idx, weight)
        return _raw.three_interpolate(points, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        points, idx, weight = ctx.saved_tensors
        return _raw.three_interpolate_grad(points, idx, weight, grad_out.contiguous()), None, None


def three_interpolate(points, idx, weight):
    '''
    Input:
        points: (b,m,c) float32 array, known points
        idx: (b,n,3) int32 array, indices to known points
        weight: (b,n,3) float32 array, weights on known points
    Output:
        out: (b,n,c) float32 array, interpolated point values
    '''
    if all(isinstance(t, torch.Tensor) for t in (points, idx, weight)) and points.requires_grad:
        return _ThreeInterpolate.apply(points, idx, weight)
    return _raw.three_interpolate(points, idx, weight)


def three_interpolate_grad(points, idx, weight, grad_out):
    """The reference's ThreeInterpolateGrad op (tf_interpolate.cpp:34-46)."""
    return _raw.three_interpolate_grad(points, idx, weight, grad_out)
