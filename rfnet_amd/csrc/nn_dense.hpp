// nn_dense.hpp -- internal interface of nn_distance.hip (the dense Chamfer sweep and the backward)
// for the composite entry points in chamfer_ext.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rfd {

// RF_NN_AUTO -> RF_NN_DENSE or RF_NN_CULLED for this shape
int resolve_mode(int b, int n, int m, int mode);
// dirs: bit 0 = direction 1 (dist1/idx1), bit 1 = direction 2 (dist2/idx2)
size_t dense_workspace_bytes(int b, int n, int m, int dirs);
int dense_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                      float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s, int dirs);

// Upstream gradients of the backward: either plain arrays (gd1 (b,n), gd2 (b,m): NnDistanceGrad as
// the reference has it) or, for the fused Chamfer LOSS, derived inside the kernel from the
// distances: gd_dir[i][j] = (gl[i][dir] / npts_dir) * 0.5 / sqrt(dist_dir[i][j]), i.e. the backward of
// loss[i][dir] = mean_j sqrt(dist_dir[i][j]) (chamfer_big / fidelity_loss, vv_recon.py:381-390).
struct GradSource {
    const float *gd1, *gd2;      // plain mode
    const float *dist1, *dist2;  // loss mode (gd1/gd2 NULL)
    const float *gl;             // (b, 2) upstream grads of the per-sample losses
};
int nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const GradSource &g,
                     const int *idx1, const int *idx2, float *grad_xyz1, float *grad_xyz2, hipStream_t s);

}  // namespace rfd
