// interpolate.hip -- three_nn, three_interpolate and its gradient for gfx950.
//
// The reference has ONLY CPU kernels for these ops (threenn_cpu, threeinterpolate_cpu,
// threeinterpolate_grad_cpu: tf_ops/interpolation/tf_interpolate.cpp:60-153), so the parity
// target is the g++ x86-64 arithmetic: the squared distance is the UNFUSED float expression
// ((dx*dx)+(dy*dy))+(dz*dz), the interpolation is (p1*w1 + p2*w2) + p3*w3 with every product
// rounded.  This file is compiled with -ffp-contract=off and uses no fmaf(), so dist / idx /
// out are bit-exact with oracle/rfops_oracle.c (and with the reference CPU bodies).
//
// three_nn: one lane per unknown point; the known set is wave-uniform, so it is streamed through
// SGPRs by scalar loads (two register sets used alternately, as query_ball_lanes_kernel) and the
// VALU ops take the SGPR operands directly -- no LDS tile, no barrier.  A candidate enters the
// lane's sorted triple only if d < b3; that test is one compare, and the insertion chain sits
// behind a wave-uniform branch (taken for ~half of the candidates at m = 1024, ever more rarely as
// m grows), instead of being predicated over every pair.
#include "common.hpp"

namespace {

constexpr int TN_TPB = 256;
constexpr int TN_SUB = 8;  // known points per scalar-load sub-chunk

__global__ __launch_bounds__(TN_TPB) void three_nn_kernel(int n, int m,
                                                          const float *__restrict__ xyz1,
                                                          const float *__restrict__ xyz2,
                                                          float *__restrict__ dist,
                                                          int *__restrict__ idx) {
    const int bi = blockIdx.y;
    const int j = blockIdx.x * TN_TPB + threadIdx.x;
    const float *__restrict__ U = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ K = xyz2 + (size_t)bi * m * 3;
    const int jj = min(j, n - 1);
    const float x1 = U[jj * 3], y1 = U[jj * 3 + 1], z1 = U[jj * 3 + 2];
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    // One candidate.  A macro, not a lambda: with the triple captured by reference the compiler
    // kept the indices in scratch memory.  The insertion is select-only (2 compares, 10 selects):
    // strict '<' everywhere, so an earlier index keeps its place on ties, exactly the reference's
    // if / else-if chain (tf_interpolate.cpp:78-93).
#define TN_CONSIDER(cx, cy, cz, kk)                                                           \
    {                                                                                         \
        const float dx_ = (cx) - x1, dy_ = (cy) - y1, dz_ = (cz) - z1;                         \
        const float xx_ = dx_ * dx_, yy_ = dy_ * dy_, zz_ = dz_ * dz_;                         \
        const float d_ = (xx_ + yy_) + zz_;                                                    \
        const bool in_ = d_ < b3;                                                              \
        if (__ballot(in_) != 0ull) { /* wave-uniform */                                        \
            asm volatile("; some lane inserts"); /* keeps this a real branch (grouping.hip) */ \
            if (in_) {                                                                         \
                const bool c1_ = d_ < b1, c2_ = d_ < b2;                                       \
                b3 = c2_ ? b2 : d_;                                                            \
                i3 = c2_ ? i2 : (kk);                                                          \
                b2 = c1_ ? b1 : (c2_ ? d_ : b2);                                               \
                i2 = c1_ ? i1 : (c2_ ? (kk) : i2);                                             \
                b1 = c1_ ? d_ : b1;                                                            \
                i1 = c1_ ? (kk) : i1;                                                          \
            }                                                                                  \
        }                                                                                      \
    }
    const int m_full = (m / TN_SUB) * TN_SUB;
    if (m_full > 0) {
        float pa[3 * TN_SUB], pb[3 * TN_SUB];
        auto fetch = [&](float (&dst)[3 * TN_SUB], int k) {
            const float *cp = K + (size_t)min(k, m - TN_SUB) * 3;  // uniform -> s_load; clamped in bounds
#pragma unroll
            for (int i = 0; i < 3 * TN_SUB; i++) dst[i] = cp[i];
        };
#define TN_SCAN8(c, k0)                  \
    _Pragma("unroll") for (int u = 0; u < TN_SUB; u++) TN_CONSIDER(c[u * 3], c[u * 3 + 1], c[u * 3 + 2], (k0) + u)
        fetch(pa, 0);
        for (int k0 = 0; k0 < m_full; k0 += 2 * TN_SUB) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): pa has arrived
            __builtin_amdgcn_sched_barrier(0);
            fetch(pb, k0 + TN_SUB);
            __builtin_amdgcn_sched_barrier(0);
            TN_SCAN8(pa, k0);
            if (k0 + TN_SUB >= m_full) break;
            __builtin_amdgcn_s_waitcnt(0xC07F);  // pb has arrived
            __builtin_amdgcn_sched_barrier(0);
            fetch(pa, k0 + 2 * TN_SUB);
            __builtin_amdgcn_sched_barrier(0);
            TN_SCAN8(pb, k0 + TN_SUB);
        }
    }
#pragma unroll 1
    for (int k = m_full; k < m; k++) TN_CONSIDER(K[k * 3], K[k * 3 + 1], K[k * 3 + 2], k);
#undef TN_SCAN8
#undef TN_CONSIDER
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
        RF_ZERO(grad_points, sizeof(float) * (size_t)b * m * c, s);
    }
    long total = (long)b * n * c;
    if (total == 0 || m == 0) return RF_OK;
    if (!grad_out || !idx || !weight) return RF_EINVAL;
    RF_LAUNCH("three_interpolate_grad", three_interpolate_grad_kernel, dim3(rf::ceil_div(total, 256)),
              dim3(256), 0, s, m, c, n, total, grad_out, idx, weight, grad_points);
    return RF_OK;
}

}  // extern "C"
