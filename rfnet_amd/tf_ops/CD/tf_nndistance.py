"""Chamfer nearest-neighbour distance: drop-in for the reference module
tf_ops/CD/tf_nndistance.py (and its duplicate pc_distance/tf_nndistance.py).

Same name, argument meaning and outputs as the reference's `nn_distance` (:9-19); tensors are
torch tensors (GPU: zero-copy) or numpy arrays instead of tf.Tensor, and the registered
gradient `_nn_distance_grad` (:26-32) becomes a torch.autograd.Function.
"""
import torch

from ... import _raw


class _NnDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        dist1, idx1, dist2, idx2 = _raw.nn_distance(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, grad_dist1, grad_idx1, grad_dist2, grad_idx2):
        # reference: nn_distance_grad(xyz1,xyz2,grad_dist1,idx1,grad_dist2,idx2), grad_idx* ignored
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        g1, g2 = _raw.nn_distance_grad(xyz1, xyz2, grad_dist1.contiguous(), idx1,
                                       grad_dist2.contiguous(), idx2)
        return g1, g2


def nn_distance(xyz1, xyz2):
    '''
Computes the distance of nearest neighbors for a pair of point clouds
input: xyz1: (batch_size,#points_1,3)  the first point cloud
input: xyz2: (batch_size,#points_2,3)  the second point cloud
output: dist1: (batch_size,#point_1)   squared distance from first to second
output: idx1:  (batch_size,#point_1)   nearest neighbor from first to second (int32)
output: dist2: (batch_size,#point_2)   squared distance from second to first
output: idx2:  (batch_size,#point_2)   nearest neighbor from second to first (int32)
    '''
    if isinstance(xyz1, torch.Tensor) and isinstance(xyz2, torch.Tensor) and (
            xyz1.requires_grad or xyz2.requires_grad):
        return _NnDistance.apply(xyz1, xyz2)
    return _raw.nn_distance(xyz1, xyz2)


def nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2):
    """The reference's NnDistanceGrad op (tf_nndistance.cpp:10-18), exposed for direct use."""
    return _raw.nn_distance_grad(xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2)
