"""Approximate earth-mover matching: drop-in for the reference module
pc_distance/tf_approxmatch.py (approx_match :10-18, match_cost :27-36, gradient :44-50)."""
import torch

from .. import _raw


def approx_match(xyz1, xyz2):
    '''
input:
    xyz1 : batch_size * #dataset_points * 3
    xyz2 : batch_size * #query_points * 3
returns:
    match : batch_size * #query_points * #dataset_points
(no gradient, like ops.NoGradient('ApproxMatch') in the reference)
    '''
    return _raw.approx_match(xyz1, xyz2)


def approx_match_levels(xyz1, xyz2, levels):
    """Extension (no reference counterpart): the same algorithm on an explicit schedule of
    exp() multipliers; `levels=None` is the reference's {-4^7 .. -4^-1, 0}."""
    return _raw.approx_match(xyz1, xyz2, levels=levels)


class _MatchCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, match):
        ctx.save_for_backward(xyz1, xyz2, match)
        return _raw.match_cost(xyz1, xyz2, match)

    @staticmethod
    def backward(ctx, grad_cost):
        # reference :44-50: grads of the op scaled by grad_cost[:,None,None]; None for match
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = _raw.match_cost_grad(xyz1, xyz2, match)
        s = grad_cost.reshape(-1, 1, 1)
        return g1 * s, g2 * s, None


def match_cost(xyz1, xyz2, match):
    '''
input:
    xyz1 : batch_size * #dataset_points * 3
    xyz2 : batch_size * #query_points * 3
    match : batch_size * #query_points * #dataset_points
returns:
    cost : batch_size
    '''
    if all(isinstance(t, torch.Tensor) for t in (xyz1, xyz2, match)) and (
            xyz1.requires_grad or xyz2.requires_grad):
        return _MatchCost.apply(xyz1, xyz2, match)
    return _raw.match_cost(xyz1, xyz2, match)


def match_cost_grad(xyz1, xyz2, match):
    """The reference's MatchCostGrad op (tf_approxmatch.cpp:16-21): (grad1, grad2)."""
    return _raw.match_cost_grad(xyz1, xyz2, match)


class _EarthMoverCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, mode):
        # one launch sequence yields the cost and MatchCostGrad's outputs; the backward only scales
        cost, g1, g2 = _raw.earth_mover(xyz1, xyz2, with_grad=True, mode=mode)
        ctx.save_for_backward(g1, g2)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        g1, g2 = ctx.saved_tensors
        s = grad_cost.reshape(-1, 1, 1)
        return g1 * s, g2 * s, None


def earth_mover_cost(xyz1, xyz2, mode="auto"):
    """match_cost(xyz1, xyz2, approx_match(xyz1, xyz2)) as ONE fused op: cost (batch_size), with the
    reference's gradient (MatchCostGrad scaled by grad_cost, match held constant) -- but the
    (batch, #query, #dataset) match tensor is never written.  Extension for the loss glue
    (`earth_mover`, vv_recon.py:392-399); the two-op chain above stays available unchanged.
    mode="swept" pins the route so that cost[i] does not depend on the batch around sample i (include/rfops.h, RF_EMD_SWEPT)."""
    if all(isinstance(t, torch.Tensor) for t in (xyz1, xyz2)) and (
            xyz1.requires_grad or xyz2.requires_grad):
        return _EarthMoverCost.apply(xyz1, xyz2, mode)
    return _raw.earth_mover(xyz1, xyz2, mode=mode)
