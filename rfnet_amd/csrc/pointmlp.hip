// pointmlp.hip -- the elementwise tail of RFNet's per-point dense layers, fused (row f2).
//
// The reference's conv2d (vv_recon.py:47-65) is conv + bias_add + activation, and most layers are
// fed tf.concat([per-point features, tf.tile(global code word)]) (e.g. :101,127,144,148,280,288,299,
// 317,343).  The split form computed by rfnet_amd.rfnet.RFNet.dcat,
//     act( y[b,n,:] + sum_k p[b,n,k] w[k,:] + r[b,:] ),
// has three kinds of terms: y = the library GEMM of the wide per-point inputs (or nothing),
// p = a NARROW per-point input (the xyz coordinates: k = 3; at most 16 channels), whose "GEMM" is a
// handful of FMAs per output element, and r = bias + (global code word) @ W, one row per sample.
// As separate tensor ops that is a K=3 GEMM, two broadcast adds and an activation: four passes over
// a (batch, points, channels) tensor (up to 32 x 16384 x 128 floats).  Here it is ONE pass: read y
// (if any), write out -- HBM-bound, 16-byte accesses, the k weights rows of a thread's four
// channels in registers.
#include "common.hpp"

namespace {

constexpr int PA_TPB = 256;
constexpr int PA_KMAX = 16;
constexpr int PA_STRIP = 64;  // points per workgroup

template <int ACT>
__device__ __forceinline__ float pa_act(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return tanhf(v);
    return v;
}

template <int ACT>
__global__ __launch_bounds__(PA_TPB) void point_affine_kernel(int n, int c, int kp, long r_stride,
                                                              const float *__restrict__ y,
                                                              const float *__restrict__ p,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ r,
                                                              float *__restrict__ out) {
    const int quads = c >> 2;                    // float4 groups per point
    const int ppi = PA_TPB / quads;              // points per pass of the workgroup
    const int cq = threadIdx.x % quads, pl = threadIdx.x / quads;
    const int bi = blockIdx.y;
    if (pl >= ppi) return;
    const float4 rr = *(const float4 *)(r + (size_t)bi * r_stride + cq * 4);
    float4 wk[PA_KMAX];
#pragma unroll
    for (int k = 0; k < PA_KMAX; k++)
        wk[k] = k < kp ? *(const float4 *)(w + (size_t)k * c + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int n0 = blockIdx.x * PA_STRIP;
    const int n1 = min(n, n0 + PA_STRIP);
    for (int j = n0 + pl; j < n1; j += ppi) {
        const size_t e = ((size_t)bi * n + j) * c + cq * 4;
        float4 acc = y ? *(const float4 *)(y + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p) {
            const float *pp = p + ((size_t)bi * n + j) * kp;
#pragma unroll
            for (int k = 0; k < PA_KMAX; k++) {
                if (k < kp) {
                    const float pv = pp[k];
                    acc.x = fmaf(pv, wk[k].x, acc.x);
                    acc.y = fmaf(pv, wk[k].y, acc.y);
                    acc.z = fmaf(pv, wk[k].z, acc.z);
                    acc.w = fmaf(pv, wk[k].w, acc.w);
                }
            }
        }
        float4 o;
        o.x = pa_act<ACT>(acc.x + rr.x);
        o.y = pa_act<ACT>(acc.y + rr.y);
        o.z = pa_act<ACT>(acc.z + rr.z);
        o.w = pa_act<ACT>(acc.w + rr.w);
        *(float4 *)(out + e) = o;
    }
}

// ---- max over the points axis: (b, n, c) -> (b, c)   (tf.reduce_max(axis=1), vv_recon.py:90,107,129,...)
// Stage 1: one workgroup per (strip of MP_STRIP points, sample): a thread owns four channels (16-byte
// loads) of every ppi-th point of the strip, the point-lanes of a channel quad meet in LDS.
// Stage 2: one workgroup per sample folds the strips.  max is exact in any order: deterministic.
constexpr int MP_TPB = 256;
constexpr int MP_STRIP = 256;

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

__global__ __launch_bounds__(MP_TPB) void maxpool_stage1_kernel(int n, int c, int nstrips, const float *__restrict__ x,
                                                                float *__restrict__ part) {
    __shared__ float4 red[MP_TPB];
    const int quads = c >> 2, ppi = MP_TPB / quads;
    const int cq = threadIdx.x % quads, pl = threadIdx.x / quads;
    const int bi = blockIdx.y, strip = blockIdx.x;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (pl < ppi) {
        const int n1 = min(n, (strip + 1) * MP_STRIP);
        for (int j = strip * MP_STRIP + pl; j < n1; j += ppi)
            m = max4(m, *(const float4 *)(x + ((size_t)bi * n + j) * c + cq * 4));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < ppi; k++) m = max4(m, red[k * quads + cq]);
        *(float4 *)(part + ((size_t)bi * nstrips + strip) * c + cq * 4) = m;
    }
}

__global__ __launch_bounds__(MP_TPB) void maxpool_stage2_kernel(int c, int nstrips, const float *__restrict__ part,
                                                                float *__restrict__ out) {
    const int bi = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += MP_TPB) {
        float m = -INFINITY;
        for (int s = 0; s < nstrips; s++) m = fmaxf(m, part[((size_t)bi * nstrips + s) * c + ch]);
        out[(size_t)bi * c + ch] = m;
    }
}

// The same with the position of the maximum (the lowest point index among ties; NaN entries are
// skipped, as fmaxf skips them): what the backward of the pooling needs.  part holds the strip maxima,
// parti (same shape) their point indices.
__device__ __forceinline__ void amax1(float v, int j, float &m, int &mj) {
    if (v > m || (v == m && j < mj)) {
        m = v;
        mj = j;
    }
}

__global__ __launch_bounds__(MP_TPB) void maxpool_idx_stage1_kernel(int n, int c, int nstrips,
                                                                    const float *__restrict__ x,
                                                                    float *__restrict__ part, int *__restrict__ parti) {
    __shared__ float4 red[MP_TPB];
    __shared__ int4 redi[MP_TPB];
    const int quads = c >> 2, ppi = MP_TPB / quads;
    const int cq = threadIdx.x % quads, pl = threadIdx.x / quads;
    const int bi = blockIdx.y, strip = blockIdx.x;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const int jfirst = strip * MP_STRIP;
    int4 mi = make_int4(jfirst, jfirst, jfirst, jfirst);
    if (pl < ppi) {
        const int n1 = min(n, (strip + 1) * MP_STRIP);
        for (int j = jfirst + pl; j < n1; j += ppi) {
            const float4 v = *(const float4 *)(x + ((size_t)bi * n + j) * c + cq * 4);
            amax1(v.x, j, m.x, mi.x);
            amax1(v.y, j, m.y, mi.y);
            amax1(v.z, j, m.z, mi.z);
            amax1(v.w, j, m.w, mi.w);
        }
    }
    red[threadIdx.x] = m;
    redi[threadIdx.x] = mi;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < ppi; k++) {
            const float4 v = red[k * quads + cq];
            const int4 vi = redi[k * quads + cq];
            amax1(v.x, vi.x, m.x, mi.x);
            amax1(v.y, vi.y, m.y, mi.y);
            amax1(v.z, vi.z, m.z, mi.z);
            amax1(v.w, vi.w, m.w, mi.w);
        }
        const size_t o = ((size_t)bi * nstrips + strip) * c + cq * 4;
        *(float4 *)(part + o) = m;
        *(int4 *)(parti + o) = mi;
    }
}

__global__ __launch_bounds__(MP_TPB) void maxpool_idx_stage2_kernel(int c, int nstrips, const float *__restrict__ part,
                                                                    const int *__restrict__ parti,
                                                                    float *__restrict__ out, int *__restrict__ idx) {
    const int bi = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += MP_TPB) {
        float m = -INFINITY;
        int mj = 0;
        for (int s = 0; s < nstrips; s++) {
            const size_t o = ((size_t)bi * nstrips + s) * c + ch;
            amax1(part[o], parti[o], m, mj);  // strips ascend: a tie keeps the earlier strip's index
        }
        out[(size_t)bi * c + ch] = m;
        idx[(size_t)bi * c + ch] = mj;
    }
}

// ---- backward of a layer tail: g = grad * act'(out), and the per-sample column sums of g (the bias /
// per-sample-row gradient) in the same pass.  As separate tensor ops that is threshold_backward (read
// grad, read out, write g) followed by a sum reduction that reads g again; here the sums ride along.
// Stage 1 as the pooling: a thread owns four channels of every ppi-th point of a strip, partial sums per
// strip to the workspace; stage 2 folds the strips of a sample in ascending order (deterministic).
// act: 0 none, 1 relu (out > 0), 2 tanh (1 - out^2), 3 leaky relu with slope 0.2 (sign of out).
template <int ACT>
__device__ __forceinline__ float act_grad(float g, float o) {
    if (ACT == 1) return o > 0.f ? g : 0.f;
    if (ACT == 2) return g * (1.f - o * o);
    if (ACT == 3) return o > 0.f ? g : g * 0.2f;
    return g;
}

template <int ACT>
__global__ __launch_bounds__(MP_TPB) void act_grad_colsum_kernel(int n, int c, int nstrips,
                                                                 const float *__restrict__ grad,
                                                                 const float *__restrict__ outv, float *__restrict__ g,
                                                                 float *__restrict__ part) {
    __shared__ float4 red[MP_TPB];
    const int quads = c >> 2, ppi = MP_TPB / quads;
    const int cq = threadIdx.x % quads, pl = threadIdx.x / quads;
    const int bi = blockIdx.y, strip = blockIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < ppi) {
        const int n1 = min(n, (strip + 1) * MP_STRIP);
        constexpr int U = 4;  // points in flight per thread
        for (int j0 = strip * MP_STRIP + pl; j0 < n1; j0 += ppi * U) {
            float4 gv[U], ov[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * ppi;
                const size_t e = ((size_t)bi * n + (j < n1 ? j : j0)) * c + cq * 4;
                gv[u] = *(const float4 *)(grad + e);
                if (ACT != 0) ov[u] = *(const float4 *)(outv + e);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * ppi;
                if (j < n1) {
                    float4 r;
                    r.x = act_grad<ACT>(gv[u].x, ACT ? ov[u].x : 0.f);
                    r.y = act_grad<ACT>(gv[u].y, ACT ? ov[u].y : 0.f);
                    r.z = act_grad<ACT>(gv[u].z, ACT ? ov[u].z : 0.f);
                    r.w = act_grad<ACT>(gv[u].w, ACT ? ov[u].w : 0.f);
                    if (g) *(float4 *)(g + ((size_t)bi * n + j) * c + cq * 4) = r;
                    acc.x += r.x;
                    acc.y += r.y;
                    acc.z += r.z;
                    acc.w += r.w;
                }
            }
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < ppi; k++) {
            const float4 v = red[k * quads + cq];
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
        *(float4 *)(part + ((size_t)bi * nstrips + strip) * c + cq * 4) = acc;
    }
}

__global__ __launch_bounds__(MP_TPB) void colsum_fold_kernel(int c, int nstrips, const float *__restrict__ part,
                                                             float *__restrict__ sums) {
    const int bi = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += MP_TPB) {
        float a = 0.f;
        for (int s = 0; s < nstrips; s++) a += part[((size_t)bi * nstrips + s) * c + ch];
        sums[(size_t)bi * c + ch] = a;
    }
}

}  // namespace

extern "C" {

size_t rf_maxpool_points_workspace_bytes(int b, int n, int c) {
    if (b <= 0 || n <= 0 || c <= 0) return 0;
    return (size_t)b * rf::ceil_div(n, MP_STRIP) * c * sizeof(float);
}

int rf_maxpool_points(int b, int n, int c, const float *x, float *out, void *workspace, size_t workspace_bytes,
                      rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0) return RF_EINVAL;
    if ((size_t)b * c == 0) return RF_OK;
    if (n == 0 || c % 4 != 0 || c / 4 > MP_TPB || b > 65535) return RF_EINVAL;
    if (!x || !out || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_maxpool_points_workspace_bytes(b, n, c)) return RF_EWORKSPACE;
    const int nstrips = rf::ceil_div(n, MP_STRIP);
    hipStream_t s = (hipStream_t)stream;
    RF_LAUNCH("maxpool_points", maxpool_stage1_kernel, dim3(nstrips, b), dim3(MP_TPB), 0, s, n, c, nstrips, x,
              (float *)workspace);
    RF_LAUNCH("maxpool_points_fold", maxpool_stage2_kernel, dim3(b), dim3(MP_TPB), 0, s, c, nstrips,
              (const float *)workspace, out);
    return RF_OK;
}

size_t rf_maxpool_points_idx_workspace_bytes(int b, int n, int c) {
    return 2 * rf_maxpool_points_workspace_bytes(b, n, c);
}

int rf_maxpool_points_idx(int b, int n, int c, const float *x, float *out, int *idx, void *workspace,
                          size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0) return RF_EINVAL;
    if ((size_t)b * c == 0) return RF_OK;
    if (n == 0 || c % 4 != 0 || c / 4 > MP_TPB || b > 65535) return RF_EINVAL;
    if (!x || !out || !idx || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_maxpool_points_idx_workspace_bytes(b, n, c)) return RF_EWORKSPACE;
    const int nstrips = rf::ceil_div(n, MP_STRIP);
    float *part = (float *)workspace;
    int *parti = (int *)(part + (size_t)b * nstrips * c);
    hipStream_t s = (hipStream_t)stream;
    RF_LAUNCH("maxpool_points_idx", maxpool_idx_stage1_kernel, dim3(nstrips, b), dim3(MP_TPB), 0, s, n, c, nstrips,
              x, part, parti);
    RF_LAUNCH("maxpool_points_idx_fold", maxpool_idx_stage2_kernel, dim3(b), dim3(MP_TPB), 0, s, c, nstrips,
              (const float *)part, (const int *)parti, out, idx);
    return RF_OK;
}

size_t rf_act_grad_colsum_workspace_bytes(int b, int n, int c) { return rf_maxpool_points_workspace_bytes(b, n, c); }

int rf_act_grad_colsum(int b, int n, int c, const float *grad, const float *out, int act, float *g, float *sums,
                       void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || act < 0 || act > 3) return RF_EINVAL;
    if ((size_t)b * c == 0) return RF_OK;
    if (c % 4 != 0 || c / 4 > MP_TPB || b > 65535) return RF_EINVAL;
    if (!sums) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        RF_ZERO(sums, sizeof(float) * (size_t)b * c, s);
        return RF_OK;
    }
    if (!grad || (act != 0 && !out) || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_act_grad_colsum_workspace_bytes(b, n, c)) return RF_EWORKSPACE;
    const int nstrips = rf::ceil_div(n, MP_STRIP);
    float *part = (float *)workspace;
    float *gw = (act == 0 && g == grad) ? nullptr : g;  // identity in place: nothing to write
    const dim3 grid(nstrips, b), blk(MP_TPB);
    switch (act) {
        case 0: RF_LAUNCH("act_grad_colsum", act_grad_colsum_kernel<0>, grid, blk, 0, s, n, c, nstrips, grad, out, gw, part); break;
        case 1: RF_LAUNCH("act_grad_colsum", act_grad_colsum_kernel<1>, grid, blk, 0, s, n, c, nstrips, grad, out, gw, part); break;
        case 2: RF_LAUNCH("act_grad_colsum", act_grad_colsum_kernel<2>, grid, blk, 0, s, n, c, nstrips, grad, out, gw, part); break;
        default: RF_LAUNCH("act_grad_colsum", act_grad_colsum_kernel<3>, grid, blk, 0, s, n, c, nstrips, grad, out, gw, part); break;
    }
    RF_LAUNCH("colsum_fold", colsum_fold_kernel, dim3(b), dim3(MP_TPB), 0, s, c, nstrips, (const float *)part, sums);
    return RF_OK;
}

int rf_point_affine_supported(int c, int kp) { return c > 0 && c % 4 == 0 && c / 4 <= PA_TPB && kp >= 0 && kp <= PA_KMAX; }

int rf_point_affine(int b, int n, int c, const float *y, const float *p, int kp, const float *w, const float *r,
                    int r_per_sample, int act, float *out, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || kp < 0 || act < 0 || act > 2) return RF_EINVAL;
    if ((size_t)b * n * c == 0) return RF_OK;
    if (!rf_point_affine_supported(c, kp) || b > 65535) return RF_EINVAL;
    if (!r || !out || (kp > 0 && (!p || !w))) return RF_EINVAL;
    const dim3 grid(rf::ceil_div(n, PA_STRIP), b);
    const long rs = r_per_sample ? c : 0;
    hipStream_t s = (hipStream_t)stream;
    const float *pp = kp > 0 ? p : nullptr;
    if (act == 0) {
        RF_LAUNCH("point_affine", point_affine_kernel<0>, grid, dim3(PA_TPB), 0, s, n, c, kp, rs, y, pp, w, r, out);
    } else if (act == 1) {
        RF_LAUNCH("point_affine", point_affine_kernel<1>, grid, dim3(PA_TPB), 0, s, n, c, kp, rs, y, pp, w, r, out);
    } else {
        RF_LAUNCH("point_affine", point_affine_kernel<2>, grid, dim3(PA_TPB), 0, s, n, c, kp, rs, y, pp, w, r, out);
    }
    return RF_OK;
}

}  // extern "C"
