// grouping.hip -- query_ball_point, group_point and its gradient for gfx950.
//
// Replaces query_ball_point_gpu / group_point_gpu / group_point_grad_gpu
// (tf_ops/grouping/tf_grouping_g.cu:3-78).  idx / pts_cnt are bit-exact with
// oracle/rfops_oracle.c: a dataset point k is in the ball iff
//     max(sqrt_rn(fma(dz,dz,fma(dx,dx,dy*dy))), 1e-20f) < radius      (distance domain)
// and the FIRST nsample hits in ascending k are kept, the rest of the row padded with the
// first hit; rows with no hit are not written.
//
// MI355X design: the reference gives each query to ONE thread that walks the dataset
// serially (divergent early exit, uncoalesced AoS loads).  Here one wave64 owns a query:
// the 64 lanes test 64 consecutive dataset points per step, a 64-bit ballot + popcount
// prefix gives every hit its output slot in ascending-k order, and the wave stops as soon as
// nsample hits are found.  Hits are written straight to their final slots; the padding
// [cnt, nsample) is written once at the end, so no slot is written twice.
#include "common.hpp"

namespace {

constexpr int QB_TPB = 256;  // 4 queries per workgroup

__global__ __launch_bounds__(QB_TPB) void query_ball_kernel(int n, int m, long nquery, float radius,
                                                            int nsample,
                                                            const float *__restrict__ xyz1,
                                                            const float *__restrict__ xyz2,
                                                            int *__restrict__ idx,
                                                            int *__restrict__ pts_cnt) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * (QB_TPB / 64) + (threadIdx.x >> 6);
    if (q >= nquery) return;  // wave-uniform
    const long bi = q / m;
    const float *D = xyz1 + bi * n * 3;
    const float x2 = xyz2[q * 3 + 0], y2 = xyz2[q * 3 + 1], z2 = xyz2[q * 3 + 2];
    int *I = idx + q * nsample;
    int cnt = 0;
    int first = -1;
    for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < n) {
            float d2 = rf::d2_fma(x2 - D[k * 3 + 0], y2 - D[k * 3 + 1], z2 - D[k * 3 + 2]);
            float d = fmaxf(sqrtf(d2), 1e-20f);
            hit = d < radius;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            if (first < 0) first = k0 + __builtin_ctzll(mask);
            const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < nsample) I[pos] = k;
            cnt = min(nsample, cnt + __builtin_popcountll(mask));
        }
    }
    if (cnt > 0)
        for (int l = cnt + lane; l < nsample; l += 64) I[l] = first;
    if (lane == 0) pts_cnt[q] = cnt;
}

__global__ void group_point_kernel(int n, int c, long per_batch /* m*nsample */, long total,
                                   const float *__restrict__ points, const int *__restrict__ idx,
                                   float *__restrict__ out) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long js = e / c;
    int ch = (int)(e - js * c);
    long bi = js / per_batch;
    int ii = idx[js];
    out[e] = points[(bi * n + ii) * c + ch];
}

__global__ void group_point_grad_kernel(int n, int c, long per_batch, long total,
                                        const float *__restrict__ grad_out,
                                        const int *__restrict__ idx, float *__restrict__ grad_points) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    long js = e / c;
    int ch = (int)(e - js * c);
    long bi = js / per_batch;
    int ii = idx[js];
    atomicAdd(&grad_points[(bi * n + ii) * c + ch], grad_out[e]);
}

}  // namespace

extern "C" {

int rf_queryballpoint(int b, int n, int m, float radius, int nsample, const float *xyz1,
                      const float *xyz2, int *idx, int *pts_cnt, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || nsample <= 0) return RF_EINVAL;
    long nquery = (long)b * m;
    if (nquery == 0) return RF_OK;
    if (!xyz2 || !idx || !pts_cnt || (n > 0 && !xyz1)) return RF_EINVAL;
    RF_LAUNCH("query_ball_point", query_ball_kernel, dim3(rf::ceil_div(nquery, QB_TPB / 64)),
              dim3(QB_TPB), 0, (hipStream_t)stream, n, m, nquery, radius, nsample, xyz1, xyz2, idx,
              pts_cnt);
    return RF_OK;
}

int rf_grouppoint(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                  float *out, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || m < 0 || nsample < 0) return RF_EINVAL;
    long per_batch = (long)m * nsample;
    long total = (long)b * per_batch * c;
    if (total == 0) return RF_OK;
    if (!points || !idx || !out) return RF_EINVAL;
    RF_LAUNCH("group_point", group_point_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, n, c, per_batch, total, points, idx, out);
    return RF_OK;
}

int rf_grouppoint_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                       const int *idx, float *grad_points, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || m < 0 || nsample < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n * c) {
        if (!grad_points) return RF_EINVAL;
        RF_HIP(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, s));
    }
    long per_batch = (long)m * nsample;
    long total = (long)b * per_batch * c;
    if (total == 0 || n == 0) return RF_OK;
    if (!grad_out || !idx) return RF_EINVAL;
    RF_LAUNCH("group_point_grad", group_point_grad_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              s, n, c, per_batch, total, grad_out, idx, grad_points);
    return RF_OK;
}

}  // extern "C"
