// interpolate.hip -- three_nn, three_interpolate and its gradient for gfx950.
//
// The reference has ONLY CPU kernels for these ops (threenn_cpu, threeinterpolate_cpu,
// threeinterpolate_grad_cpu: tf_ops/interpolation/tf_interpolate.cpp:60-153), so the parity
// target is the g++ x86-64 arithmetic: the squared distance is the UNFUSED float expression
// ((dx*dx)+(dy*dy))+(dz*dz), the interpolation is (p1*w1 + p2*w2) + p3*w3 with every product
// rounded.  This file is compiled with -ffp-contract=off and uses no fmaf(), so dist / idx /
// out are bit-exact with oracle/rfops_oracle.c (and with the reference CPU bodies).
//
// three_nn: one lane per unknown point; the known set is staged through LDS in tiles and
// broadcast-read, so each known point is fetched from HBM once per workgroup.
#include "common.hpp"

namespace {

constexpr int TN_TPB = 256;
constexpr int TN_TILE = 1024;

__global__ __launch_bounds__(TN_TPB) void three_nn_kernel(int n, int m,
                                                          const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2,
                                                          float *__restrict__ dist,
                                                          int *__restrict__ idx) {
    __shared__ float4 tile[TN_TILE];
    const int bi = blockIdx.y;
    const int j = blockIdx.x * TN_TPB + threadIdx.x;
    const float *U = xyz1 + (size_t)bi * n * 3;
    const float *K = xyz2 + (size_t)bi * m * 3;
    const int jj = min(j, n - 1);
    const float x1 = U[jj * 3], y1 = U[jj * 3 + 1], z1 = U[jj * 3 + 2];
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int t0 = 0; t0 < m; t0 += TN_TILE) {
        const int cnt = min(TN_TILE, m - t0);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt; k += TN_TPB) {
            const float *p = K + (size_t)(t0 + k) * 3;
            tile[k] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            const float4 c = tile[k];
            float dx = c.x - x1, dy = c.y - y1, dz = c.z - z1;
            float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            float d = (xx + yy) + zz;
            int kk = t0 + k;
            // strict '<' insertion: an earlier index keeps its place on ties
            if (d < b1) {
                b3 = b2; i3 = i2;
                b2 = b1; i2 = i1;
                b1 = d;  i1 = kk;
            } else if (d < b2) {
                b3 = b2; i3 = i2;
                b2 = d;  i2 = kk;
            } else if (d < b3) {
                b3 = d;  i3 = kk;
            }
        }
    }
    if (j < n) {
        size_t o = ((size_t)bi * n + j) * 3;
        dist[o] = b1; dist[o + 1] = b2; dist[o + 2] = b3;
        idx[o] = i1;  idx[o + 1] = i2;  idx[o + 2] = i3;
    }
}

__global__ void three_interpolate_kernel(int m, int c, int n, long total,
                                         const float *__restrict__ points,
                                         const int *__restrict__ idx,
                                         const float *__restrict__ weight, float *__restrict__ out) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long bj = e / c;  // b*n + j
    int l = (int)(e - bj * c);
    long bi = bj / n;
    const float *P = points + bi * m * c;
    float a = P[(long)idx[bj * 3 + 0] * c + l] * weight[bj * 3 + 0];
    float b = P[(long)idx[bj * 3 + 1] * c + l] * weight[bj * 3 + 1];
    float d = P[(long)idx[bj * 3 + 2] * c + l] * weight[bj * 3 + 2];
    out[e] = (a + b) + d;
}

__global__ void three_interpolate_grad_kernel(int m, int c, int n, long total,
                                              const float *__restrict__ grad_out,
                                              const int *__restrict__ idx,
                                              const float *__restrict__ weight,
                                              float *__restrict__ grad_points) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long bj = e / c;
    int l = (int)(e - bj * c);
    long bi = bj / n;
    float *G = grad_points + bi * m * c;
    float g = grad_out[e];
#pragma unroll
    for (int t = 0; t < 3; t++) atomicAdd(&G[(long)idx[bj * 3 + t] * c + l], g * weight[bj * 3 + t]);
}

}  // namespace

extern "C" {

int rf_threenn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
               rf_stream_t stream) {
    if (b < 0 || b > 65535 || n < 0 || m < 0) return RF_EINVAL;  // the batch is grid.y
    if (b == 0 || n == 0) return RF_OK;
    if (!xyz1 || !dist || !idx || (m > 0 && !xyz2)) return RF_EINVAL;
    RF_LAUNCH("three_nn", three_nn_kernel, dim3(rf::ceil_div(n, TN_TPB), b), dim3(TN_TPB), 0,
              (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
    return RF_OK;
}

int rf_threeinterpolate(int b, int m, int c, int n, const float *points, const int *idx,
                        const float *weight, float *out, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || c < 0) return RF_EINVAL;
    long total = (long)b * n * c;
    if (total == 0) return RF_OK;
    if (!points || !idx || !weight || !out) return RF_EINVAL;
    RF_LAUNCH("three_interpolate", three_interpolate_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, m, c, n, total, points, idx, weight, out);
    return RF_OK;
}

int rf_threeinterpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                             const float *weight, float *grad_points, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || c < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * m * c) {
        if (!grad_points) return RF_EINVAL;
        RF_HIP(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, s));
    }
    long total = (long)b * n * c;
    if (total == 0 || m == 0) return RF_OK;
    if (!grad_out || !idx || !weight) return RF_EINVAL;
    RF_LAUNCH("three_interpolate_grad", three_interpolate_grad_kernel, dim3(rf::ceil_div(total, 256)),
              dim3(256), 0, s, m, c, n, total, grad_out, idx, weight, grad_points);
    return RF_OK;
}

}  // extern "C"
