// sampling.hip -- farthest_point_sample, gather_point and its gradient for gfx950.
//
// Replaces farthestpointsamplingKernel / gatherpointKernel / scatteraddpointKernel
// (tf_ops/sampling/tf_sampling_g.cu:105-192).  Indices are bit-exact with
// oracle/rfops_oracle.c, including the reference's tie order, which its 512-thread launch
// defines: largest running min-distance; among equals the smallest (k mod 512); among those
// the smallest k (tf_sampling_g.cu:146,158).
//
// MI355X design: FPS is a chain of m-1 dependent block-wide arg-max reductions -- latency
// bound, one workgroup per cloud.  The reference keeps the running min-distances in global
// memory and re-reads points beyond its 3072-point LDS cache from global every iteration.
// Here one 1024-thread workgroup holds the WHOLE cloud (up to 16384 points) in registers:
// 16 points x (x,y,z,running-min) per lane, so an iteration touches no memory except a
// 16-entry LDS exchange and one scalar load: per-lane scan -> wave maximum by DPP row
// rotations -> the lowest lane holding it publishes (d2,k) in the wave's LDS slot
// (double-buffered: ONE barrier per iteration) -> every 16-lane row re-reduces the slots by
// DPP (max d2, then min tie rank) -> the winner's xyz is re-read with a scalar load.
#include "common.hpp"

namespace {

constexpr int FPS_MAX_REG_POINTS = 16384;

// tie rank of point k under the reference's 512-thread layout: lower is preferred
__device__ __forceinline__ unsigned tie_rank(int k) { return ((unsigned)(k & 511) << 22) | (unsigned)(k >> 9); }

__device__ __forceinline__ unsigned long long make_key(float d2, int k) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(0xFFFFFFFFu - tie_rank(k));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), o, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

struct Slot {
    unsigned long long key;
    float x, y, z;
    int k;
};

// ---- single-instruction helpers.  hipcc wraps fminf/fmaxf in canonicalising v_max x,x and
// expands a float DPP reduction step into mov/nop/mov_dpp/max/max; on the serial path of FPS
// every instruction counts, so these are written as the one instruction they are.  (Inputs are
// never NaN here: distances of finite points, or the -1 / 1e38 sentinels.)
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v = max(v, v rotated by N lanes inside each 16-lane row).  The s_nop covers the
// VALU-write -> DPP-read hazard (the assembler/hazard recogniser does not look inside asm).
#define RF_DPP_STEP(OP, N)                                                                          \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float row_allmax(float v) {
    RF_DPP_STEP("v_max_f32_dpp", 8);
    RF_DPP_STEP("v_max_f32_dpp", 4);
    RF_DPP_STEP("v_max_f32_dpp", 2);
    RF_DPP_STEP("v_max_f32_dpp", 1);
    return v;
}
__device__ __forceinline__ unsigned row_allmin_u(unsigned v) {
    RF_DPP_STEP("v_min_u32_dpp", 8);
    RF_DPP_STEP("v_min_u32_dpp", 4);
    RF_DPP_STEP("v_min_u32_dpp", 2);
    RF_DPP_STEP("v_min_u32_dpp", 1);
    return v;
}
#undef RF_DPP_STEP
// maximum over the wave, uniform: row all-reduce, then the four row results
__device__ __forceinline__ float wave_allmax(float v) {
    v = row_allmax(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return vmax3(vmax3(r0, r1, r2), r3, r3);
}

// NT threads (multiple of 512), PPT points per thread, all register resident.
// Thread t owns k = (t & 511) + 512 * (s * (NT/512) + (t >> 9)), s = 0..PPT-1, so all its points
// share (k mod 512) and ascend with s; inside a wave a lower lane has a lower (k mod 512).
// One iteration:
//   scan      7 VALU per point (3 sub, mul, 2 fma, min) + half a v_max3 for the per-lane maximum
//             VALUE -- no per-point compare/select for the arg-max;
//   wave max  DPP row rotations + 4 v_readlane -> wm (uniform);
//   arg-max   per lane the lowest s with td[s] == the lane's maximum (compare + select per
//             point, overlapping the wave reduction); one v_cmp_eq(mx, wm) gives the lanes
//             holding the wave maximum, the lowest of them is the winner under the reference's
//             tie order and a v_readlane fetches its s;
//   exchange  lane 0 publishes (wm, k) in the wave's LDS slot (double-buffered: ONE barrier per
//             iteration); every 16-lane row re-reduces the <=16 slots by DPP (max d2, then
//             min tie rank); the winner's coordinates come back by a scalar load.
template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_reg_kernel(int n, int m, const float *__restrict__ inp,
                                                     int *__restrict__ out) {
    constexpr int NW = NT / 64;
    constexpr int HALVES = NT / 512;
    static_assert(NW <= 16, "slot reduction is one 16-lane DPP row");
    __shared__ float slot_d[2][16];
    __shared__ int slot_k[2][16];
    const int bi = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ P = inp + (size_t)bi * n * 3;
    int *__restrict__ O = out + (size_t)bi * m;

    float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        int k = (t & 511) + 512 * (s * HALVES + (t >> 9));
        if (k < n) {
            px[s] = P[k * 3 + 0];
            py[s] = P[k * 3 + 1];
            pz[s] = P[k * 3 + 2];
            td[s] = 1e38f;
        } else {
            px[s] = py[s] = pz[s] = 0.f;
            td[s] = -1.0f;  // min(d,-1) = -1: below every real distance, like the reference's best=-1
        }
    }
    if (t < 32) {  // unused slots never win
        slot_d[t >> 4][t & 15] = -2.0f;
        slot_k[t >> 4][t & 15] = 0;
    }
    if (t == 0) O[0] = 0;
    __syncthreads();
    float ox = P[0], oy = P[1], oz = P[2];  // old = 0
    for (int j = 1; j < m; j++) {
        float mx = -1.0f;
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            td[s] = vmin(rf::d2_fma(px[s] - ox, py[s] - oy, pz[s] - oz), td[s]);
            if (PPT == 1) {
                mx = td[0];
            } else if (s & 1) {
                mx = vmax3(mx, td[s - 1], td[s]);
            }
        }
        // The wave's arg-max under the tie order (lowest lane, then lowest s).  Per lane, the lowest
        // s attaining the lane's OWN maximum: 2 VALU per point, independent of the wave reduction
        // (so it overlaps the DPP latency); then ONE compare finds the lanes holding the wave
        // maximum and a v_readlane fetches the winner's s.  (16 v_cmp_eq + a 48-deep scalar
        // select chain per iteration cost 25 % of the kernel: ablation in DESIGN.md 5.3.)
        int sidx = 0;
#pragma unroll
        for (int s = PPT - 1; s >= 1; s--) sidx = (td[s] == mx) ? s : sidx;
        if (PPT > 1) sidx = (td[0] == mx) ? 0 : sidx;
        const float wm = wave_allmax(mx);
        const unsigned long long hl = __ballot(mx == wm);
        const int wl = hl ? __builtin_ctzll(hl) : 0;
        const int bs = __builtin_amdgcn_readlane(sidx, wl);
        const int wt = wave * 64 + wl;
        const int wk = (wt & 511) + 512 * (bs * HALVES + (wt >> 9));
        const int buf = j & 1;
        if (lane == 0) {
            slot_d[buf][wave] = wm;
            slot_k[buf][wave] = wm >= 0.f ? wk : 0;
        }
        __syncthreads();
        const float sd = slot_d[buf][lane & 15];
        const int sk = slot_k[buf][lane & 15];
        const float gm = row_allmax(sd);
        const unsigned rank = sd == gm ? tie_rank(sk) : 0xFFFFFFFFu;
        const unsigned gr = __builtin_amdgcn_readfirstlane(row_allmin_u(rank));
        const int gk = (int)(((gr & 0x3FFFFFu) << 9) | (gr >> 22));
        ox = P[gk * 3 + 0];  // uniform address: scalar loads
        oy = P[gk * 3 + 1];
        oz = P[gk * 3 + 2];
        if (t == 0) O[j] = gk;
    }
}

// Fallback for clouds beyond the register-resident limit: running min-distances in the
// caller's temp buffer (b*n floats), points re-read from global/L2.  Same selection rule.
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, const float *__restrict__ inp,
                                                       float *__restrict__ temp, int *__restrict__ out) {
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ Slot slots[2][NW];
    const int bi = blockIdx.x;
    const int t = threadIdx.x;
    const float *P = inp + (size_t)bi * n * 3;
    float *T = temp + (size_t)bi * n;
    int *O = out + (size_t)bi * m;
    for (int k = t; k < n; k += NT) T[k] = 1e38f;
    if (t == 0) O[0] = 0;
    float ox = P[0], oy = P[1], oz = P[2];
    for (int j = 1; j < m; j++) {
        float best = -1.0f;
        int bk = 0;
        for (int k = (t & 511) + 512 * (t >> 9); k < n; k += NT) {
            float d = rf::d2_fma(P[k * 3] - ox, P[k * 3 + 1] - oy, P[k * 3 + 2] - oz);
            float d2 = fminf(d, T[k]);
            T[k] = d2;
            if (d2 > best) {
                best = d2;
                bk = k;
            }
        }
        const unsigned long long key = best >= 0.f ? make_key(best, bk) : 0ull;
        const unsigned long long wmax = wave_max_u64(key);
        const int buf = j & 1;
        if (key == wmax && (key != 0ull || (t & 63) == 0)) {
            Slot sl;
            sl.key = key;
            int kk = key != 0ull ? bk : 0;
            sl.x = P[kk * 3]; sl.y = P[kk * 3 + 1]; sl.z = P[kk * 3 + 2];
            sl.k = kk;
            slots[buf][t >> 6] = sl;
        }
        __syncthreads();
        unsigned long long gk = slots[buf][0].key;
        int gw = 0;
        for (int w = 1; w < NW; w++) {
            unsigned long long kw = slots[buf][w].key;
            if (kw > gk) {
                gk = kw;
                gw = w;
            }
        }
        ox = slots[buf][gw].x;
        oy = slots[buf][gw].y;
        oz = slots[buf][gw].z;
        if (t == 0) O[j] = slots[buf][gw].k;
    }
}

__global__ void gather_kernel(int n, int m, long total, const float *__restrict__ inp,
                              const int *__restrict__ idx, float *__restrict__ out) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    long bi = g / m;
    int a = idx[g];
    const float *p = inp + (bi * n + a) * 3;
    out[g * 3 + 0] = p[0];
    out[g * 3 + 1] = p[1];
    out[g * 3 + 2] = p[2];
}

__global__ void scatteradd_kernel(int n, int m, long total, const float *__restrict__ out_g,
                                  const int *__restrict__ idx, float *__restrict__ inp_g) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    long bi = g / m;
    int a = idx[g];
    float *p = inp_g + (bi * n + a) * 3;
    atomicAdd(p + 0, out_g[g * 3 + 0]);
    atomicAdd(p + 1, out_g[g * 3 + 1]);
    atomicAdd(p + 2, out_g[g * 3 + 2]);
}

// ---- prob_sample (ProbSample op): cumsumKernel + binarysearchKernel, tf_sampling_g.cu:7-104.
// One workgroup per row; the scan keeps the reference's association order exactly (groups of
// 4, up-sweep / down-sweep over the group totals with the (i + i>>5) padded index, compensated
// carry between 8192-element chunks), so the cumulative sums -- and therefore the sampled
// indices -- are bit-exact with oracle/rfops_oracle.c::orc_cumsum.  Adds only: no FMA question.
constexpr int CS_BS = 2048;
__device__ __forceinline__ int pad5(int i) { return i + (i >> 5); }

__global__ __launch_bounds__(512) void cumsum_kernel(int n, const float *__restrict__ inp,
                                                     float *__restrict__ out) {
    __shared__ float buffer4[CS_BS * 4];
    __shared__ float buffer[CS_BS + (CS_BS >> 5) + 1];
    const int i = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
    float runningsum = 0.f, runningsum2 = 0.f;
    for (int j = 0; j < n; j += CS_BS * 4) {
        const float *in = inp + (size_t)i * n + j;
        const int n24_i = min(n - j, CS_BS * 4);
        const int n24 = (n24_i + 3) & ~3, n2 = n24 >> 2;
        for (int k = t * 4; k < n24_i; k += nt * 4) {
            if (k + 3 < n24_i) {
                float v1 = in[k], v2 = in[k + 1];
                v2 += v1;
                float v3 = in[k + 2], v4 = in[k + 3];
                v4 += v3;
                v3 += v2;
                v4 += v2;
                buffer4[k] = v1; buffer4[k + 1] = v2; buffer4[k + 2] = v3; buffer4[k + 3] = v4;
                buffer[pad5(k >> 2)] = v4;
            } else {
                float v = 0.f;
                for (int k2 = k; k2 < n24_i; k2++) { v += in[k2]; buffer4[k2] = v; }
                for (int k2 = n24_i; k2 < n24; k2++) buffer4[k2] = v;
                buffer[pad5(k >> 2)] = v;
            }
        }
        int u = 0;
        for (; (2 << u) <= n2; u++) {
            __syncthreads();
            for (int k = t; k < (n2 >> (u + 1)); k += nt)
                buffer[pad5((((k << 1) + 2) << u) - 1)] += buffer[pad5((((k << 1) + 1) << u) - 1)];
        }
        u--;
        for (; u >= 0; u--) {
            __syncthreads();
            for (int k = t; k < ((n2 - (1 << u)) >> (u + 1)); k += nt)
                buffer[pad5((((k << 1) + 3) << u) - 1)] += buffer[pad5((((k << 1) + 2) << u) - 1)];
        }
        __syncthreads();
        for (int k = t * 4; k < n24; k += nt * 4) {
            if (k != 0) {
                const float add = buffer[pad5((k >> 2) - 1)];
                buffer4[k] += add; buffer4[k + 1] += add; buffer4[k + 2] += add; buffer4[k + 3] += add;
            }
        }
        __syncthreads();
        for (int k = t; k < n24_i; k += nt) out[(size_t)i * n + j + k] = buffer4[k] + runningsum;
        const float tt = buffer[pad5(n2 - 1)] + runningsum2;
        const float r2 = runningsum + tt;
        runningsum2 = tt - (r2 - runningsum);
        runningsum = r2;
        __syncthreads();
    }
}

__global__ void binarysearch_kernel(int n, int m, long total, const float *__restrict__ dataset,
                                    const float *__restrict__ query, int *__restrict__ result) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const long i = g / m;
    const float *ds = dataset + i * n;
    int base = 1;
    while (base < n) base <<= 1;
    const float q = query[g] * ds[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && ds[r - k] >= q) r -= k;
    result[g] = r;
}

}  // namespace

extern "C" {

size_t rf_farthestpointsampling_temp_floats(int b, int n) {
    return n > FPS_MAX_REG_POINTS ? (size_t)b * n : 0;
}

int rf_farthestpointsampling(int b, int n, int m, const float *inp, float *temp, int *out,
                             rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (n == 0) return RF_EINVAL;  // cannot sample from an empty cloud
    if (!inp || !out) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
#define FPS_CASE(NT, PPT)                                                                      \
    RF_LAUNCH("fps_reg", (fps_reg_kernel<NT, PPT>), dim3(b), dim3(NT), 0, s, n, m, inp, out); \
    return RF_OK
    if (n <= 512) { FPS_CASE(512, 1); }
    if (n <= 1024) { FPS_CASE(1024, 1); }
    if (n <= 2048) { FPS_CASE(1024, 2); }
    if (n <= 4096) { FPS_CASE(1024, 4); }
    if (n <= 8192) { FPS_CASE(1024, 8); }
    if (n <= 16384) { FPS_CASE(512, 32); }
#undef FPS_CASE
    if (!temp) return RF_EINVAL;
    RF_LAUNCH("fps_mem", fps_mem_kernel, dim3(b), dim3(1024), 0, s, n, m, inp, temp, out);
    return RF_OK;
}

int rf_gatherpoint(int b, int n, int m, const float *inp, const int *idx, float *out,
                   rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    long total = (long)b * m;
    if (total == 0) return RF_OK;
    if (!inp || !idx || !out) return RF_EINVAL;
    RF_LAUNCH("gather_point", gather_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, n, m, total, inp, idx, out);
    return RF_OK;
}

int rf_scatteraddpoint(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                       rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) {
        if (!inp_g) return RF_EINVAL;
        RF_HIP(hipMemsetAsync(inp_g, 0, sizeof(float) * 3 * (size_t)b * n, s));
    }
    long total = (long)b * m;
    if (total == 0 || n == 0) return RF_OK;
    if (!out_g || !idx) return RF_EINVAL;
    RF_LAUNCH("scatteradd_point", scatteradd_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0, s, n,
              m, total, out_g, idx, inp_g);
    return RF_OK;
}

int rf_probsample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                  rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (n == 0) return RF_EINVAL;
    if (!inp_p || !inp_r || !temp || !out) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RF_LAUNCH("cumsum", cumsum_kernel, dim3(b), dim3(512), 0, s, n, inp_p, temp);
    const long total = (long)b * m;
    RF_LAUNCH("binarysearch", binarysearch_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0, s, n, m,
              total, (const float *)temp, inp_r, out);
    return RF_OK;
}

}  // extern "C"
