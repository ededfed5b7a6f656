"""Farthest-point sampling and point gathering: drop-in for the reference module
tf_ops/sampling/tf_sampling.py (gather_point :29-37 + gradient :43-47,
farthest_point_sample :48-56; prob_sample is commented out in the reference)."""
import torch

from ... import _raw


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        ctx.save_for_backward(inp, idx)
        return _raw.gather_point(inp, idx)

    @staticmethod
    def backward(ctx, out_g):
        inp, idx = ctx.saved_tensors
        return _raw.gather_point_grad(inp, idx, out_g.contiguous()), None


def gather_point(inp, idx):
    '''
input:
    batch_size * ndataset * 3   float32
    batch_size * npoints        int32
returns:
    batch_size * npoints * 3    float32
    '''
    if isinstance(inp, torch.Tensor) and isinstance(idx, torch.Tensor) and inp.requires_grad:
        return _GatherPoint.apply(inp, idx)
    return _raw.gather_point(inp, idx)


def gather_point_grad(inp, idx, out_g):
    """The reference's GatherPointGrad op (tf_sampling.cpp:53-63)."""
    return _raw.gather_point_grad(inp, idx, out_g)


def farthest_point_sample(npoint, inp):
    '''
input:
    int32
    batch_size * ndataset * 3   float32
returns:
    batch_size * npoint         int32
(no gradient, like ops.NoGradient('FarthestPointSample'))
    '''
    return _raw.farthest_point_sample(npoint, inp)


def prob_sample(inp, inpr):
    '''
input:
    batch_size * ncategory float32   (unnormalised weights)
    batch_size * npoints   float32   (uniform numbers in [0,1))
returns:
    batch_size * npoints   int32     (inverse-CDF sample of a category per number)
The ProbSample op exists in the reference's library (tf_sampling.cpp:14-27,66-92); its Python
wrapper is commented out there (tf_sampling.py:13-22).
    '''
    return _raw.prob_sample(inp, inpr)
