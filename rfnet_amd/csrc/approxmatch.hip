// approxmatch.hip -- approx_match / match_cost / match_cost_grad (EMD) for gfx950.
//
// Replaces approxmatch, matchcost, matchcostgrad1/2 (pc_distance/tf_approxmatch.cu:1-295).
// Algorithm = the reference CUDA schedule (levels {-4^7..-4^-1, 0}; P1/P2/P3 per level, see
// SURVEY.md 3.4), same fp32 expressions (d2 = fma chain, fma accumulates, exp as
// v_exp_f32(d2 * level*log2e) -- the analogue of __expf = ex2.approx(x*log2e)).
//
// MI355X design, not the reference's one-block-per-batch-element loop:
//   * every phase of every level is one launch over (row blocks) x (batch); a row's sweep
//     over the other set is split over the 4 waves of the workgroup (4 column segments,
//     combined in segment order), two rows per lane, columns staged as float4
//     {x,y,z,scalar} in LDS and read with one broadcast ds_read_b128 per column per wave;
//   * `match` is NOT read-modify-written once per level (the reference moves 21 x 4nm bytes
//     per sample).  The per-level ratio vectors (10 x (n+m) floats) are kept in the
//     workspace and match is produced ONCE at the end:
//         match[l][k] = fma(rl_9[k]*e_9, rr_9[l], ... fma(rl_0[k]*e_0, rr_0[l], 0))
//     which is the same fma chain, in the same level order, as the reference's 10 "+="
//     (tf_approxmatch.cu:152; App. A: the += is a fused multiply-add), so the bits do not
//     depend on this restructuring.  HBM traffic for match: one 4nm-byte write.
//   * row sums are accumulated per column segment and combined in segment order, so they
//     differ from the reference's strictly sequential order in the last bits (tolerance
//     stated in tests/test_emd_gpu.py).
#include "common.hpp"

namespace {

constexpr float kLog2e = 1.44269502f;  // 0x3FB8AA3B, the constant __expf multiplies by
constexpr int TPB = 256;
constexpr int RPT = 2;              // rows per lane
constexpr int ROWS = 64 * RPT;      // rows per workgroup
constexpr int NSEG = 4;             // column segments = waves per workgroup
constexpr int CT = 256;             // columns per segment tile
constexpr int LVG = 16;             // levels per group in the materialisation kernel
constexpr int MAX_LEVELS = 64;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__global__ void am_init_kernel(int n, int m, float multiL, float multiR, float *remainL,
                               float *remainR, size_t stride) {
    int bi = blockIdx.y;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) remainL[bi * stride + j] = multiL;
    if (j < m) remainR[bi * stride + j] = multiR;
}

// One phase of one level.  rows: the set owning the output vector; cols: the other set.
//  PHASE 1: rows=xyz1 (k), cols=xyz2 (l) with s=remainR[l];  acc from 1e-9: acc=fma(e,s,acc)
//           ratioL[k] = remainL[k] / acc
//  PHASE 2: rows=xyz2 (l), cols=xyz1 (k) with s=ratioL[k];   acc from 0:    acc=fma(e,s,acc)
//           t=acc*remainR[l]; cons=min(remainR[l]/(t+1e-9),1); ratioR[l]=remainR[l]*cons;
//           remainR[l]=max(0,remainR[l]-t)
//  PHASE 3: rows=xyz1 (k), cols=xyz2 (l) with s=ratioR[l];   acc from 0: acc=fma(ratioL[k]*e,s,acc)
//           remainL[k]=max(0,remainL[k]-acc)
template <int PHASE>
__global__ __launch_bounds__(TPB) void am_phase_kernel(int nr, int nc, const float *rows_xyz,
                                                       const float *cols_xyz, const float *col_s,
                                                       size_t s_stride, float *remain_row,
                                                       const float *ratioL_in, float *ratio_out,
                                                       float c_level) {
    __shared__ float4 tile[NSEG][CT];
    __shared__ float part[NSEG][ROWS];
    const int bi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int seg = threadIdx.x >> 6;
    rows_xyz += (size_t)bi * nr * 3;
    cols_xyz += (size_t)bi * nc * 3;
    col_s += (size_t)bi * s_stride;
    remain_row += (size_t)bi * s_stride;
    ratio_out += (size_t)bi * s_stride;
    if (PHASE == 3) ratioL_in += (size_t)bi * s_stride;

    float rx[RPT], ry[RPT], rz[RPT], acc[RPT], rl[RPT];
    int row[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        row[r] = blockIdx.x * ROWS + r * 64 + lane;
        int rr = min(row[r], nr - 1);
        rx[r] = rows_xyz[rr * 3 + 0];
        ry[r] = rows_xyz[rr * 3 + 1];
        rz[r] = rows_xyz[rr * 3 + 2];
        acc[r] = (PHASE == 1 && seg == 0) ? 1e-9f : 0.f;
        rl[r] = (PHASE == 3) ? ratioL_in[rr] : 1.f;
    }
    const int seglen = (nc + NSEG - 1) / NSEG;
    const int c0 = seg * seglen;
    const int c1 = min(nc, c0 + seglen);
    for (int t0 = 0; t0 < seglen; t0 += CT) {
        const int base = c0 + t0;
        const int cnt = max(0, min(CT, c1 - base));
        __syncthreads();
        for (int k = lane; k < cnt; k += 64) {
            const float *p = cols_xyz + (size_t)(base + k) * 3;
            tile[seg][k] = make_float4(p[0], p[1], p[2], col_s[base + k]);
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < cnt; k++) {
            const float4 c = tile[seg][k];
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                float d2 = rf::d2_fma(c.x - rx[r], c.y - ry[r], c.z - rz[r]);
                float e = fast_exp2(d2 * c_level);
                if (PHASE == 3) e = rl[r] * e;
                acc[r] = fmaf(e, c.w, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) part[seg][r * 64 + lane] = acc[r];
    __syncthreads();
    if (seg == 0) {
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            if (row[r] >= nr) continue;
            float s = part[0][r * 64 + lane];
#pragma unroll
            for (int g = 1; g < NSEG; g++) s += part[g][r * 64 + lane];
            if (PHASE == 1) {
                ratio_out[row[r]] = remain_row[row[r]] / s;
            } else if (PHASE == 2) {
                float rem = remain_row[row[r]];
                float t = s * rem;
                float cons = fminf(rem / (t + 1e-9f), 1.0f);
                ratio_out[row[r]] = rem * cons;
                remain_row[row[r]] = fmaxf(0.0f, rem - t);
            } else {
                remain_row[row[r]] = fmaxf(0.0f, remain_row[row[r]] - s);
            }
        }
    }
}

// match[l][k] = sum over levels (in order) of fma(ratioL_lv[k]*e_lv, ratioR_lv[l], acc).
// thread <-> k (coalesced 256-B stores per wave per l); workgroup = 256 k x LSEG l.
constexpr int LSEG = 64;
struct LevelConsts {
    float c[MAX_LEVELS];  // level * log2e
};

__global__ __launch_bounds__(TPB) void am_match_kernel(int n, int m, const float *xyz1,
                                                       const float *xyz2, const float *ratios,
                                                       size_t lv_stride, size_t b_stride,
                                                       int lv0, int nlv, LevelConsts lc,
                                                       float *match) {
    // ratios: [b][level][ (ratioL: n) (ratioR: m) ]
    __shared__ float cxyz[LSEG][4];
    __shared__ float crr[LSEG][LVG];
    const int bi = blockIdx.z;
    const int k = blockIdx.x * TPB + threadIdx.x;
    const int l0 = blockIdx.y * LSEG;
    const int lcnt = min(LSEG, m - l0);
    xyz1 += (size_t)bi * n * 3;
    xyz2 += (size_t)bi * m * 3;
    ratios += (size_t)bi * b_stride + (size_t)lv0 * lv_stride;
    match += (size_t)bi * n * m;
    for (int i = threadIdx.x; i < lcnt * LVG; i += TPB) {
        int l = i / LVG, v = i % LVG;
        crr[l][v] = v < nlv ? ratios[(size_t)v * lv_stride + n + l0 + l] : 0.f;
    }
    for (int i = threadIdx.x; i < lcnt; i += TPB) {
        cxyz[i][0] = xyz2[(size_t)(l0 + i) * 3 + 0];
        cxyz[i][1] = xyz2[(size_t)(l0 + i) * 3 + 1];
        cxyz[i][2] = xyz2[(size_t)(l0 + i) * 3 + 2];
    }
    __syncthreads();
    if (k >= n) return;
    const float x1 = xyz1[k * 3], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
    float rl[LVG];
#pragma unroll
    for (int v = 0; v < LVG; v++) rl[v] = v < nlv ? ratios[(size_t)v * lv_stride + k] : 0.f;
    for (int l = 0; l < lcnt; l++) {
        float d2 = rf::d2_fma(cxyz[l][0] - x1, cxyz[l][1] - y1, cxyz[l][2] - z1);
        float acc = lv0 == 0 ? 0.f : match[(size_t)(l0 + l) * n + k];
#pragma unroll
        for (int v = 0; v < LVG; v++) {
            if (v < nlv) {
                float p = rl[v] * fast_exp2(d2 * lc.c[lv0 + v]);
                acc = fmaf(p, crr[l][v], acc);
            }
        }
        match[(size_t)(l0 + l) * n + k] = acc;
    }
}

// ---- match_cost: cost[i] = sum_{l,k} match[l][k] * sqrt(d2(k,l)); HBM-bound stream of match.
// workgroup = 256 k x MC_L l; per-workgroup partial -> workspace; fixed-order final sum.
constexpr int MC_L = 32;
__global__ __launch_bounds__(TPB) void mc_partial_kernel(int n, int m, const float *xyz1,
                                                         const float *xyz2, const float *match,
                                                         float *partial) {
    __shared__ float cxyz[MC_L][4];
    __shared__ float wsum[TPB / 64];
    const int bi = blockIdx.z;
    const int k = blockIdx.x * TPB + threadIdx.x;
    const int l0 = blockIdx.y * MC_L;
    const int lcnt = min(MC_L, m - l0);
    xyz1 += (size_t)bi * n * 3;
    xyz2 += (size_t)bi * m * 3;
    match += (size_t)bi * n * m;
    if (threadIdx.x < lcnt) {
        cxyz[threadIdx.x][0] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 0];
        cxyz[threadIdx.x][1] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 1];
        cxyz[threadIdx.x][2] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 2];
    }
    __syncthreads();
    float sum = 0.f;
    if (k < n) {
        const float x1 = xyz1[k * 3], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
#pragma unroll 8
        for (int l = 0; l < lcnt; l++) {
            float d = sqrtf(rf::d2_fma(cxyz[l][0] - x1, cxyz[l][1] - y1, cxyz[l][2] - z1));
            sum = fmaf(d, match[(size_t)(l0 + l) * n + k], sum);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        partial[((size_t)bi * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
    }
}

__global__ void mc_final_kernel(const float *partial, int per_batch, float *cost) {
    __shared__ float red[256];
    const int bi = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < per_batch; i += 256) s += partial[(size_t)bi * per_batch + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) cost[bi] = red[0];
}

// ---- match_cost_grad, one pass over match for both gradients.
//  grad1[k] = sum_l match[l][k] (x1_k - x2_l) * rsqrt(max(d2,1e-20))     (tf_approxmatch.cu:270-291)
//  grad2[l] = sum_k match[l][k] (x2_l - x1_k) * rsqrt(max(d2,1e-20))     (:229-269)
// workgroup = 256 k x MG_L l.  grad1 partial per thread -> atomicAdd; grad2 partial per l via
// wave reduction -> atomicAdd.  Outputs are zero-filled first.
constexpr int MG_L = 64;
__global__ __launch_bounds__(TPB) void mcg_kernel(int n, int m, const float *xyz1,
                                                  const float *xyz2, const float *match,
                                                  float *grad1, float *grad2) {
    __shared__ float cxyz[MG_L][4];
    __shared__ float g2[MG_L][3];
    const int bi = blockIdx.z;
    const int k = blockIdx.x * TPB + threadIdx.x;
    const int l0 = blockIdx.y * MG_L;
    const int lcnt = min(MG_L, m - l0);
    xyz1 += (size_t)bi * n * 3;
    xyz2 += (size_t)bi * m * 3;
    match += (size_t)bi * n * m;
    grad1 += (size_t)bi * n * 3;
    grad2 += (size_t)bi * m * 3;
    if (threadIdx.x < lcnt) {
        cxyz[threadIdx.x][0] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 0];
        cxyz[threadIdx.x][1] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 1];
        cxyz[threadIdx.x][2] = xyz2[(size_t)(l0 + threadIdx.x) * 3 + 2];
    }
    if (threadIdx.x < MG_L * 3) (&g2[0][0])[threadIdx.x] = 0.f;
    __syncthreads();
    const bool live = k < n;
    const int kk = live ? k : n - 1;
    const float x1 = xyz1[kk * 3], y1 = xyz1[kk * 3 + 1], z1 = xyz1[kk * 3 + 2];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int l = 0; l < lcnt; l++) {
        float dx = x1 - cxyz[l][0], dy = y1 - cxyz[l][1], dz = z1 - cxyz[l][2];
        float mt = live ? match[(size_t)(l0 + l) * n + kk] : 0.f;
        float q = mt * __builtin_amdgcn_rsqf(fmaxf(rf::d2_fma(dx, dy, dz), 1e-20f));
        float vx = dx * q, vy = dy * q, vz = dz * q;
        ax += vx; ay += vy; az += vz;
        // grad2 wants (x2-x1)*q = -v summed over k: wave-reduce, one LDS atomic per wave
        float sx = -vx, sy = -vy, sz = -vz;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_down(sx, o, 64);
            sy += __shfl_down(sy, o, 64);
            sz += __shfl_down(sz, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g2[l][0], sx);
            atomicAdd(&g2[l][1], sy);
            atomicAdd(&g2[l][2], sz);
        }
    }
    if (live) {
        atomicAdd(&grad1[k * 3 + 0], ax);
        atomicAdd(&grad1[k * 3 + 1], ay);
        atomicAdd(&grad1[k * 3 + 2], az);
    }
    __syncthreads();
    if (threadIdx.x < lcnt * 3) {
        int l = threadIdx.x / 3, c = threadIdx.x % 3;
        atomicAdd(&grad2[(size_t)(l0 + l) * 3 + c], g2[l][c]);
    }
}

int default_levels(float *lv) {
    int c = 0;
    for (int j = 7; j >= -2; j--) lv[c++] = (j == -2) ? 0.0f : -ldexpf(1.0f, 2 * j);
    return c;
}

size_t am_ws_floats(int b, int n, int m, int nlevels) {
    // per batch element: remainL[n] remainR[m], then per level ratioL[n] ratioR[m]
    return (size_t)b * (size_t)(n + m) * (size_t)(1 + nlevels);
}

}  // namespace

extern "C" {

size_t rf_approxmatch_workspace_bytes(int b, int n, int m, int nlevels) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    if (nlevels <= 0) nlevels = 10;
    return am_ws_floats(b, n, m, nlevels) * sizeof(float);
}

int rf_approxmatch_levels(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                          const float *levels_host, int nlevels, void *workspace,
                          size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || nlevels <= 0 || nlevels > MAX_LEVELS || !levels_host)
        return RF_EINVAL;
    if (b == 0 || n == 0 || m == 0) return RF_OK;
    if (!xyz1 || !xyz2 || !match || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_approxmatch_workspace_bytes(b, n, m, nlevels)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // multiL / multiR: integer division as in tf_approxmatch.cu:4-10
    float multiL, multiR;
    if (n >= m) { multiL = 1.f; multiR = (float)(n / m); }
    else        { multiL = (float)(m / n); multiR = 1.f; }

    const size_t S = (size_t)(n + m);            // floats per vector pair
    const size_t bstride = S * (size_t)(1 + nlevels);
    float *w = (float *)workspace;
    float *remainL = w, *remainR = w + n;       // + bi*bstride
    float *ratios = w + S;                       // [level][ratioL n | ratioR m]
    {
        int mx = n > m ? n : m;
        RF_LAUNCH("am_init", am_init_kernel, dim3(rf::ceil_div(mx, 256), b), dim3(256), 0, s, n, m,
                  multiL, multiR, remainL, remainR, bstride);
    }
    LevelConsts lc;
    for (int v = 0; v < MAX_LEVELS; v++) lc.c[v] = v < nlevels ? levels_host[v] * kLog2e : 0.f;
    const dim3 g1(rf::ceil_div(n, ROWS), b), g2(rf::ceil_div(m, ROWS), b);
    for (int v = 0; v < nlevels; v++) {
        float *ratioL = ratios + (size_t)v * S, *ratioR = ratioL + n;
        RF_LAUNCH("am_phase1", am_phase_kernel<1>, g1, dim3(TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)remainR, bstride, remainL, (const float *)nullptr, ratioL, lc.c[v]);
        RF_LAUNCH("am_phase2", am_phase_kernel<2>, g2, dim3(TPB), 0, s, m, n, xyz2, xyz1,
                  (const float *)ratioL, bstride, remainR, (const float *)nullptr, ratioR, lc.c[v]);
        RF_LAUNCH("am_phase3", am_phase_kernel<3>, g1, dim3(TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)ratioR, bstride, remainL, (const float *)ratioL, ratioL /*unused*/,
                  lc.c[v]);
    }
    const dim3 gm(rf::ceil_div(n, TPB), rf::ceil_div(m, LSEG), b);
    for (int lv0 = 0; lv0 < nlevels; lv0 += LVG) {
        int nlv = nlevels - lv0 < LVG ? nlevels - lv0 : LVG;
        RF_LAUNCH("am_match", am_match_kernel, gm, dim3(TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)ratios, S, bstride, lv0, nlv, lc, match);
    }
    return RF_OK;
}

int rf_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                   void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    float lv[16];
    int nl = default_levels(lv);
    return rf_approxmatch_levels(b, n, m, xyz1, xyz2, match, lv, nl, workspace, workspace_bytes,
                                 stream);
}

size_t rf_matchcost_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return (size_t)b * rf::ceil_div(n, TPB) * rf::ceil_div(m, MC_L) * sizeof(float);
}

int rf_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                 float *cost, void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0) return RF_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0 || m == 0) {
        RF_HIP(hipMemsetAsync(cost, 0, sizeof(float) * b, s));
        return RF_OK;
    }
    if (!xyz1 || !xyz2 || !match || !cost || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_matchcost_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    dim3 g(rf::ceil_div(n, TPB), rf::ceil_div(m, MC_L), b);
    RF_LAUNCH("mc_partial", mc_partial_kernel, g, dim3(TPB), 0, s, n, m, xyz1, xyz2, match,
              (float *)workspace);
    RF_LAUNCH("mc_final", mc_final_kernel, dim3(b), dim3(256), 0, s, (const float *)workspace,
              (int)(g.x * g.y), cost);
    return RF_OK;
}

int rf_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *grad1, float *grad2, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) RF_HIP(hipMemsetAsync(grad1, 0, sizeof(float) * 3 * (size_t)b * n, s));
    if ((size_t)b * m) RF_HIP(hipMemsetAsync(grad2, 0, sizeof(float) * 3 * (size_t)b * m, s));
    if (b == 0 || n == 0 || m == 0) return RF_OK;
    dim3 g(rf::ceil_div(n, TPB), rf::ceil_div(m, MG_L), b);
    RF_LAUNCH("mc_grad", mcg_kernel, g, dim3(TPB), 0, s, n, m, xyz1, xyz2, match, grad1, grad2);
    return RF_OK;
}

}  // extern "C"
