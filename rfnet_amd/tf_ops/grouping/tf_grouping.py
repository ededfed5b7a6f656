"""Ball query and grouping: drop-in for the reference module tf_ops/grouping/tf_grouping.py
(query_ball_point :8-20, group_point :33-41 + gradient :42-46, knn_point :48-73;
select_top_k :22-31)."""
import torch

from ... import _raw


def query_ball_point(radius, nsample, xyz1, xyz2):
    '''
    Input:
        radius: float32, ball search radius
        nsample: int32, number of points selected in each ball region
        xyz1: (batch_size, ndataset, 3) float32 array, input points
        xyz2: (batch_size, npoint, 3) float32 array, query points
    Output:
        idx: (batch_size, npoint, nsample) int32 array, indices to input points
        pts_cnt: (batch_size, npoint) int32 array, number of unique points in each local region
    '''
    return _raw.query_ball_point(radius, nsample, xyz1, xyz2)


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(points, idx)
        return _raw.group_point(points, idx)

    @staticmethod
    def backward(ctx, grad_out):
        points, idx = ctx.saved_tensors
        return _raw.group_point_grad(points, idx, grad_out.contiguous()), None


def group_point(points, idx):
    '''
    Input:
        points: (batch_size, ndataset, channel) float32 array, points to sample from
        idx: (batch_size, npoint, nsample) int32 array, indices to points
    Output:
        out: (batch_size, npoint, nsample, channel) float32 array, values sampled from points
    '''
    if isinstance(points, torch.Tensor) and isinstance(idx, torch.Tensor) and points.requires_grad:
        return _GroupPoint.apply(points, idx)
    return _raw.group_point(points, idx)


def group_point_grad(points, idx, grad_out):
    """The reference's GroupPointGrad op (tf_grouping.cpp:56-64)."""
    return _raw.group_point_grad(points, idx, grad_out)


def knn_point(k, xyz1, xyz2):
    '''
    Input:
        k: int32, number of k in k-nn search
        xyz1: (batch_size, ndataset, c) float32 array, input points
        xyz2: (batch_size, npoint, c) float32 array, query points
    Output:
        val: (batch_size, npoint, k) float32 array, NEGATED squared L2 distances (top_k of -dist)
        idx: (batch_size, npoint, k) int32 array, indices to input points
    Pure tensor ops in the reference too (tf.nn.top_k of -dist, tf_grouping.py:64-73).
    '''
    xyz1 = torch.as_tensor(xyz1)
    xyz2 = torch.as_tensor(xyz2)
    dist = ((xyz1.unsqueeze(1) - xyz2.unsqueeze(2)) ** 2).sum(-1)
    val, idx = torch.topk(-dist, k=int(k), dim=-1)
    return val, idx.to(torch.int32)


def select_top_k(k, dist):
    '''
    Input:
        k: int32, number of k SMALLEST elements selected
        dist: (b,m,n) float32 array, distance matrix, m query points, n dataset points
    Output:
        idx: (b,m,n) int32 array, first k in n are indices to the top k
        dist_out: (b,m,n) float32 array, first k in n are the top k
    '''
    return _raw.select_top_k(k, dist)
