// nn_pruned.hip -- Chamfer nearest-neighbour distance with exact spatial culling, for gfx950.
//
// Same results as nn_distance.hip (and the reference's NmDistanceKernel,
// tf_ops/CD/tf_nndistance_g.cu:4-130): d2 = fma(dz,dz,fma(dx,dx,dy*dy)) of the surviving pairs is
// evaluated with the very same instruction sequence, the minimum is the minimum over ALL
// candidates and ties go to the lowest ORIGINAL index -- only pairs that provably cannot attain
// or equal a point's minimum are skipped.  The dense sweep is at the VALU issue limit of its
// instruction mix; the remaining lever is not to evaluate most pairs at all.
//
//   nnp_sort_kernel   one workgroup per cloud.  Per-axis histogram equalisation (256 bins ->
//                     32 cells of equal marginal population: far outliers cannot flatten the
//                     grid), a 15-bit Hilbert key per point, counting sort in LDS (32768 bins =
//                     128 KiB of the CU's 160 KiB).  Output: the cloud as float4 records
//                     (x, y, z, original index) in key order, padded to a multiple of 64, and the
//                     axis-aligned boxes of every 16-record block and 64-record superblock.
//                     The ORDER only steers how much gets culled; any permutation is correct.
//   nnp_sweep_kernel  one wave per 64 consecutive sorted queries (one per lane).  Candidate
//                     superblocks are visited in ascending order of the box-to-box lower bound
//                     until that bound exceeds every lane's current minimum (strictly); inside
//                     a superblock each 16-candidate block is skipped when its box is strictly
//                     farther than the current minimum of every lane.  Surviving blocks are
//                     streamed through SGPRs by scalar loads (the dense sweep's trick: VALU
//                     ops take the candidate coordinates as SGPR operands) at 6.5 VALU per pair.
//                     When the query set is small (few groups), 4 waves share a group, each
//                     taking every 4th superblock, with the running minima shared through LDS.
//
// Why the bounds are safe in fp32: the lower bound of a box is evaluated with the SAME
// sequence (sub, mul, fma, fma) on per-axis gaps g = max(lo - q, q - hi, 0).  For any candidate c
// inside the box, |c - q| >= g holds exactly per axis; fp32 subtraction, multiplication and fma
// are monotone in each argument under round-to-nearest, so the computed bound is <= the computed
// d2 of every candidate in the box.  Boxes are culled only on bound > minimum (strict), so a
// candidate that would TIE the minimum is always evaluated.
//
// Lowest original index on ties (the reference's strict '<' scan order) with candidates visited
// out of order: a lane tracks the running minimum VALUE, the first visited block that attained
// it, and a flag raised when a later block's minimum equals it bit for bit.  Without the flag all
// candidates attaining the minimum sit in that one block: a re-scan of its 16 records takes the
// lowest original index among the exact matches.  With the flag (duplicated points, symmetric
// configurations) the wave re-scans, for that one query, every superblock whose bound does not
// exceed the minimum, lanes across candidates, and reduces the lowest matching index.
#include <stdlib.h>

#include "common.hpp"
#include "nn_pruned.hpp"

namespace {

constexpr int BS = 16;             // candidates per block
constexpr int SBB = 4;             // blocks per superblock
constexpr int SB = BS * SBB;       // 64 records: one superblock = one query group = one wave
constexpr int KEYBITS = 15;        // 5 bits per axis
constexpr int NBINS = 1 << KEYBITS;
constexpr int STPB = 1024;         // sort kernel threads
constexpr int HB = 256;            // equalisation histogram bins per axis
constexpr int MAXSB = rfp::kMaxPoints / SB;  // 1024: superblock id fits the key's low 10 bits
constexpr unsigned IDMASK = 0x3FFu;

struct SortArgs {
    int b;
    int n[2], npad[2];
    const float *src[2];  // (b, n, 3)
    float4 *sorted[2];    // (b, npad)
    float4 *box16[2];     // (b, npad/16, 2): lo, hi
    float4 *box64[2];     // (b, npad/64, 2)
};

// Skilling's axes-to-transpose Hilbert mapping, 5 bits per axis -> 15-bit index.
__device__ __forceinline__ unsigned hilbert15(unsigned x, unsigned y, unsigned z) {
    unsigned X[3] = {x, y, z};
#pragma unroll
    for (unsigned Q = 16; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const unsigned t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned t = 0;
#pragma unroll
    for (unsigned Q = 16; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
    unsigned key = 0;
#pragma unroll
    for (int bit = 0; bit < 5; bit++)
#pragma unroll
        for (int i = 0; i < 3; i++) key |= ((X[i] >> bit) & 1u) << (3 * bit + 2 - i);
    return key & (NBINS - 1);
}

__device__ __forceinline__ int axis_bin(float v, float lo, float scale) {
    if (!isfinite(v)) return HB - 1;
    const float f = fminf(fmaxf((v - lo) * scale, 0.f), (float)(HB - 1));
    return (int)f;  // NaN products (inf * 0) fall through fmaxf as 0
}

__global__ __launch_bounds__(STPB) void nnp_sort_kernel(SortArgs a) {
    __shared__ unsigned hist[NBINS];
    __shared__ unsigned ahist[3][HB];
    __shared__ unsigned char cellmap[3][HB];
    __shared__ float red[STPB / 64][6];
    __shared__ unsigned wsum[STPB / 64];
    __shared__ float frame[6];  // lo[3], scale[3]

    const int set = (int)blockIdx.x >= a.b;
    const int bi = blockIdx.x - (set ? a.b : 0);
    const int n = a.n[set], npad = a.npad[set];
    const float *__restrict__ src = a.src[set] + (size_t)bi * n * 3;
    float4 *__restrict__ out = a.sorted[set] + (size_t)bi * npad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < NBINS; i += STPB) hist[i] = 0;
    if (tid < 3 * HB) (&ahist[0][0])[tid] = 0;

    // 1. bounding box of the finite coordinates
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += STPB) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = src[(size_t)i * 3 + c];
            if (isfinite(v)) {
                lo[c] = fminf(lo[c], v);
                hi[c] = fmaxf(hi[c], v);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            red[wave][c] = lo[c];
            red[wave][3 + c] = hi[c];
        }
    }
    __syncthreads();
    if (tid < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < STPB / 64; w++) {
            l = fminf(l, red[w][tid]);
            h = fmaxf(h, red[w][3 + tid]);
        }
        const float ext = h - l;
        const bool ok = isfinite(ext) && ext > 0.f;
        frame[tid] = ok ? l : 0.f;
        frame[3 + tid] = ok ? (float)HB / ext : 0.f;
    }
    __syncthreads();
    const float fl[3] = {frame[0], frame[1], frame[2]};
    const float fs[3] = {frame[3], frame[4], frame[5]};

    // 2. per-axis histograms -> equal-population cells
    for (int i = tid; i < n; i += STPB) {
#pragma unroll
        for (int c = 0; c < 3; c++) atomicAdd(&ahist[c][axis_bin(src[(size_t)i * 3 + c], fl[c], fs[c])], 1u);
    }
    __syncthreads();
    if (wave < 3) {
        unsigned c4[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c4[k] = ahist[wave][lane * 4 + k];
            s += c4[k];
        }
        unsigned incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        unsigned run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // cell of a bin = the 1/32-quantile its first point falls into
            const unsigned cell = (unsigned)(((unsigned long long)run * 32u) / (unsigned)n);
            cellmap[wave][lane * 4 + k] = (unsigned char)(cell > 31u ? 31u : cell);
            run += c4[k];
        }
    }
    __syncthreads();

    auto key_of = [&](int i) {
        const unsigned cx = cellmap[0][axis_bin(src[(size_t)i * 3 + 0], fl[0], fs[0])];
        const unsigned cy = cellmap[1][axis_bin(src[(size_t)i * 3 + 1], fl[1], fs[1])];
        const unsigned cz = cellmap[2][axis_bin(src[(size_t)i * 3 + 2], fl[2], fs[2])];
        return hilbert15(cx, cy, cz);
    };

    // 3. key histogram
    for (int i = tid; i < n; i += STPB) atomicAdd(&hist[key_of(i)], 1u);
    __syncthreads();

    // 4. exclusive scan of the 32768 bins: 32 consecutive bins per thread
    {
        constexpr int PER = NBINS / STPB;
        unsigned s = 0;
#pragma unroll 8
        for (int k = 0; k < PER; k++) s += hist[tid * PER + k];
        unsigned incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; w++) base += wsum[w];
        unsigned run = base + incl - s;
#pragma unroll 8
        for (int k = 0; k < PER; k++) {
            const unsigned c = hist[tid * PER + k];
            hist[tid * PER + k] = run;
            run += c;
        }
    }
    __syncthreads();

    // 5. scatter (the order inside a key is whatever the atomics give: results do not depend on it)
    for (int i = tid; i < n; i += STPB) {
        const unsigned pos = atomicAdd(&hist[key_of(i)], 1u);
        out[pos] = make_float4(src[(size_t)i * 3 + 0], src[(size_t)i * 3 + 1], src[(size_t)i * 3 + 2],
                               __int_as_float(i));
    }
    for (int i = n + tid; i < npad; i += STPB) out[i] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
    __threadfence();
    __syncthreads();

    // 6. boxes of the 16-record blocks and the 64-record superblocks (padding and NaN excluded)
    float4 *__restrict__ b16 = a.box16[set] + (size_t)bi * (npad / BS) * 2;
    float4 *__restrict__ b64 = a.box64[set] + (size_t)bi * (npad / SB) * 2;
    // a quad of lanes per superblock, one block each; quads stay inside a wave
    for (int blk = tid; blk < npad / BS; blk += STPB) {
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        const float4 *p = out + (size_t)blk * BS;
        for (int u = 0; u < BS; u++) {
            const float4 r = p[u];
            if (__float_as_int(r.w) >= 0) {
                l[0] = fminf(l[0], r.x); h[0] = fmaxf(h[0], r.x);
                l[1] = fminf(l[1], r.y); h[1] = fmaxf(h[1], r.y);
                l[2] = fminf(l[2], r.z); h[2] = fmaxf(h[2], r.z);
            }
        }
        b16[(size_t)blk * 2 + 0] = make_float4(l[0], l[1], l[2], 0.f);
        b16[(size_t)blk * 2 + 1] = make_float4(h[0], h[1], h[2], 0.f);
        // npad/BS is a multiple of 4 and STPB too, so all 4 lanes of a quad are in this iteration
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                l[c] = fminf(l[c], __shfl_xor(l[c], o, 64));
                h[c] = fmaxf(h[c], __shfl_xor(h[c], o, 64));
            }
        }
        if ((blk & 3) == 0) {
            b64[(size_t)(blk >> 2) * 2 + 0] = make_float4(l[0], l[1], l[2], 0.f);
            b64[(size_t)(blk >> 2) * 2 + 1] = make_float4(h[0], h[1], h[2], 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------

// The record / box arrays are separate `const __restrict__` kernel parameters (not struct members):
// only then does the compiler know they are never written during the kernel and turn the
// wave-uniform loads of candidates and boxes into scalar loads.
struct SweepArgs {
    int b;
    int n[2], npad[2];
    int groups[2];   // npad[d] / 64
    int nw[2];       // waves per query group: 1 or 4
    int wg0;         // workgroups of direction 0
};

__device__ __forceinline__ float min3_acc(float acc, float a, float b) {
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

#define RFP_DPP(OP, N) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
// uniform minimum / maximum over the wave: DPP row rotations, then the four row results
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    RFP_DPP("v_min_u32_dpp", 8);
    RFP_DPP("v_min_u32_dpp", 4);
    RFP_DPP("v_min_u32_dpp", 2);
    RFP_DPP("v_min_u32_dpp", 1);
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}
// (inputs are +-inf or non-negative distances, never NaN; non-negative floats order as integers)
__device__ __forceinline__ float wave_max_nonneg(float f) {
    // -inf (lanes that do not take part) maps to 0, which never exceeds a participant's value
    unsigned v = f < 0.f ? 0u : __float_as_uint(f);
    RFP_DPP("v_max_u32_dpp", 8);
    RFP_DPP("v_max_u32_dpp", 4);
    RFP_DPP("v_max_u32_dpp", 2);
    RFP_DPP("v_max_u32_dpp", 1);
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return __uint_as_float(max(max(r0, r1), max(r2, r3)));
}
#undef RFP_DPP

// lower bound of d2 between point (qx,qy,qz) and the box [lo,hi], same instruction sequence as d2
__device__ __forceinline__ float box_bound(float qx, float qy, float qz, const float4 lo, const float4 hi) {
    const float gx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f);
    const float gy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f);
    const float gz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
    return rf::d2_fma(gx, gy, gz);
}
// box to box: gap per axis between [alo,ahi] and [blo,bhi]
__device__ __forceinline__ float boxbox_bound(const float4 alo, const float4 ahi, const float4 blo,
                                              const float4 bhi) {
    const float gx = fmaxf(fmaxf(blo.x - ahi.x, alo.x - bhi.x), 0.f);
    const float gy = fmaxf(fmaxf(blo.y - ahi.y, alo.y - bhi.y), 0.f);
    const float gz = fmaxf(fmaxf(blo.z - ahi.z, alo.z - bhi.z), 0.f);
    return rf::d2_fma(gx, gy, gz);
}

// direction d: nearest neighbour of every point of set d among set 1-d -> dist_d, idx_d (b, n[d]).
// stats (optional): [dir][4] = waves, superblock steps, block tests, block scans.
__global__ __launch_bounds__(256) void nnp_sweep_kernel(
    SweepArgs a, const float4 *__restrict__ sorted0, const float4 *__restrict__ sorted1,
    const float4 *__restrict__ b16_0, const float4 *__restrict__ b16_1, const float4 *__restrict__ b64_0,
    const float4 *__restrict__ b64_1, float *__restrict__ dist0, float *__restrict__ dist1,
    int *__restrict__ idx0, int *__restrict__ idx1, unsigned long long *__restrict__ stats) {
    __shared__ unsigned keys[4][MAXSB];
    __shared__ int shbest[64];
    __shared__ float md[4][64];
    __shared__ unsigned mi[4][64];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int wg = blockIdx.x;
    const int dir = wg >= a.wg0;
    if (dir) wg -= a.wg0;
    const int cd = 1 - dir;
    const bool shared4 = a.nw[dir] == 4;
    const int G = a.groups[dir];
    const int gid = shared4 ? wg : wg * 4 + wib;
    const int sub = shared4 ? wib : 0, nsub = shared4 ? 4 : 1;
    if (gid >= a.b * G) return;  // only in the 1-wave-per-group shape (no barriers there)
    const int bi = gid / G, g = gid - bi * G;

    const float4 *__restrict__ Q = (dir ? sorted1 : sorted0) + (size_t)bi * a.npad[dir];
    const float4 *__restrict__ C = (dir ? sorted0 : sorted1) + (size_t)bi * a.npad[cd];
    const float4 *__restrict__ CB16 = (dir ? b16_0 : b16_1) + (size_t)bi * (a.npad[cd] / BS) * 2;
    const float4 *__restrict__ CB64 = (dir ? b64_0 : b64_1) + (size_t)bi * (a.npad[cd] / SB) * 2;
    const int nsb = a.npad[cd] / SB;

    const float4 q = Q[(size_t)g * SB + lane];
    const int qorig = __float_as_int(q.w);
    const bool valid = qorig >= 0;
    const float4 *gb = (dir ? b64_1 : b64_0) + ((size_t)bi * G + g) * 2;  // uniform
    const float4 glo = gb[0], ghi = gb[1];

    // lower bound group box <-> candidate superblock, truncated (downwards) into the high 22 bits
    // of a key whose low 10 bits are the superblock id: the wave minimum of the keys is the next
    // superblock in ascending bound order.  Entry e of this wave's list is superblock sub + nsub*e;
    // lane e % 64 owns it (writes it, consumes it, keeps the minimum of its entries in `lmin`).
    const int nmine = (nsb - sub + nsub - 1) / nsub;
    unsigned lmin = 0xFFFFFFFFu;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int e = lane; e < nmine; e += 64) {
        const int s = sub + nsub * e;
        const float lb = boxbox_bound(glo, ghi, CB64[(size_t)s * 2], CB64[(size_t)s * 2 + 1]);
        const unsigned key = (__float_as_uint(lb) & ~IDMASK) | (unsigned)s;
        keys[wib][e] = key;
        lmin = min(lmin, key);
    }
    if (shared4) {
        if (wib == 0) shbest[lane] = 0x7F800000;  // +inf
        __syncthreads();
    }

    float best = INFINITY;  // this wave's own minimum, the block that first attained it, tie flag
    int bblk = 0;
    bool tie = false;
    float cull = INFINITY;  // <= best: also what the other waves of the group have found
    unsigned n_step = 0, n_test = 0, n_scan = 0;

    for (;;) {
        const unsigned kmin = wave_min_u32(lmin);
        if (kmin == 0xFFFFFFFFu) break;
        if (shared4) cull = fminf(cull, __int_as_float(shbest[lane]));
        const float bound = __uint_as_float(kmin & ~IDMASK);
        const float worst = wave_max_nonneg(valid ? cull : -INFINITY);
        if (bound > worst) break;  // every remaining superblock is strictly farther than every lane's minimum
        const int s = (int)(kmin & IDMASK);
        const int e = (s - sub) / nsub;
        if (lane == (e & 63)) {
            keys[wib][e] = 0xFFFFFFFFu;
            lmin = 0xFFFFFFFFu;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (int ee = lane; ee < nmine; ee += 64) lmin = min(lmin, keys[wib][ee]);
        }
        n_step++;
#pragma unroll 1
        for (int j = 0; j < SBB; j++) {
            const int blk = s * SBB + j;
            const float lb = box_bound(q.x, q.y, q.z, CB16[(size_t)blk * 2], CB16[(size_t)blk * 2 + 1]);
            n_test++;
            if (__ballot(valid && lb <= cull) == 0ull) continue;
            n_scan++;
            const float4 *cp = C + (size_t)blk * BS;  // uniform: scalar loads
            float cm = INFINITY;
#pragma unroll
            for (int u = 0; u < BS; u += 2) {
                const float4 c0 = cp[u], c1 = cp[u + 1];
                const float d0 = rf::d2_fma(c0.x - q.x, c0.y - q.y, c0.z - q.z);
                const float d1 = rf::d2_fma(c1.x - q.x, c1.y - q.y, c1.z - q.z);
                cm = min3_acc(cm, d0, d1);
            }
            if (cm < best) {
                best = cm;
                bblk = blk;
                tie = false;
            } else if (cm == best) {
                tie = true;
            }
        }
        cull = fminf(cull, best);
        if (shared4) atomicMin(&shbest[lane], __float_as_int(cull));
    }

    // lowest original index among the exact matches of the winning block
    unsigned besti = 0xFFFFFFFFu;
    {
        const float4 *cp = C + (size_t)bblk * BS;  // per lane
#pragma unroll 4
        for (int u = 0; u < BS; u++) {
            const float4 c = cp[u];
            const float d = rf::d2_fma(c.x - q.x, c.y - q.y, c.z - q.z);
            if (d == best) besti = min(besti, __float_as_uint(c.w));  // padding carries 0xFFFFFFFF
        }
    }
    // queries whose minimum was attained in more than one visited block: exact re-scan, one query
    // at a time, lanes across the candidates of every superblock that can hold a match
    unsigned long long tm = __ballot(tie && valid);
    while (tm) {
        const int L = __builtin_ctzll(tm);
        tm &= tm - 1;
        const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), L));
        const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), L));
        const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), L));
        const float bL = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), L));
        unsigned cand = 0xFFFFFFFFu;
        for (int s0 = 0; s0 < nsb; s0 += 64) {
            const int s = s0 + lane;
            bool need = false;
            if (s < nsb) need = box_bound(qx, qy, qz, CB64[(size_t)s * 2], CB64[(size_t)s * 2 + 1]) <= bL;
            unsigned long long m2 = __ballot(need);
            while (m2) {
                const int k = __builtin_ctzll(m2);
                m2 &= m2 - 1;
                const float4 c = C[(size_t)(s0 + k) * SB + lane];
                const float d = rf::d2_fma(c.x - qx, c.y - qy, c.z - qz);
                if (d == bL) cand = min(cand, __float_as_uint(c.w));
            }
        }
        cand = wave_min_u32(cand);
        if (lane == L) besti = cand;
    }

    if (stats && lane == 0) {
        atomicAdd(&stats[dir * 4 + 0], 1ull);
        atomicAdd(&stats[dir * 4 + 1], (unsigned long long)n_step);
        atomicAdd(&stats[dir * 4 + 2], (unsigned long long)n_test);
        atomicAdd(&stats[dir * 4 + 3], (unsigned long long)n_scan);
    }

    if (shared4) {
        md[wib][lane] = best;
        mi[wib][lane] = besti;
        __syncthreads();
        if (wib != 0) return;
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const float d = md[w][lane];
            const unsigned i = mi[w][lane];
            if (d < best || (d == best && i < besti)) {
                best = d;
                besti = i;
            }
        }
    }
    if (valid) {
        (dir ? dist1 : dist0)[(size_t)bi * a.n[dir] + qorig] = best;
        (dir ? idx1 : idx0)[(size_t)bi * a.n[dir] + qorig] = besti == 0xFFFFFFFFu ? 0 : (int)besti;
    }
}

int round_up(long v, int q) { return (int)((v + q - 1) / q * q); }

struct PPlan {
    int npad[2];
    size_t off_sorted[2], off_b16[2], off_b64[2], off_stats, bytes;
};

PPlan make_pplan(int b, int n, int m) {
    PPlan p;
    const int nn[2] = {n, m};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    for (int s = 0; s < 2; s++) {
        p.npad[s] = round_up(nn[s], SB);
        p.off_sorted[s] = take((size_t)b * p.npad[s] * sizeof(float4));
        p.off_b16[s] = take((size_t)b * (p.npad[s] / BS) * 2 * sizeof(float4));
        p.off_b64[s] = take((size_t)b * (p.npad[s] / SB) * 2 * sizeof(float4));
    }
    p.off_stats = take(8 * sizeof(unsigned long long));
    p.bytes = off;
    return p;
}

}  // namespace

namespace rfp {

bool pruned_supported(int b, int n, int m) {
    return b > 0 && n > 0 && m > 0 && n <= kMaxPoints && m <= kMaxPoints;
}

size_t pruned_workspace_bytes(int b, int n, int m) {
    if (!pruned_supported(b, n, m)) return 0;
    return make_pplan(b, n, m).bytes;
}

int pruned_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                       float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s,
                       unsigned long long *stats_out) {
    if (!pruned_supported(b, n, m)) return RF_EINVAL;
    const PPlan p = make_pplan(b, n, m);
    if (workspace_bytes < p.bytes) return RF_EWORKSPACE;
    char *w = (char *)workspace;
    SortArgs sa;
    SweepArgs wa;
    sa.b = wa.b = b;
    const int nn[2] = {n, m};
    const float *src[2] = {xyz1, xyz2};
    for (int k = 0; k < 2; k++) {
        sa.n[k] = wa.n[k] = nn[k];
        sa.npad[k] = wa.npad[k] = p.npad[k];
        sa.src[k] = src[k];
        sa.sorted[k] = (float4 *)(w + p.off_sorted[k]);
        sa.box16[k] = (float4 *)(w + p.off_b16[k]);
        sa.box64[k] = (float4 *)(w + p.off_b64[k]);
        wa.groups[k] = p.npad[k] / SB;
        // a set with few groups cannot fill the chip with one wave per group: 4 waves share a group
        static const long split_below = getenv("RF_NNP_SPLIT") ? atol(getenv("RF_NNP_SPLIT")) : 4096;
        wa.nw[k] = ((long)b * wa.groups[k] < split_below) ? 4 : 1;
    }
    unsigned long long *stats = nullptr;
    if (stats_out) {
        stats = (unsigned long long *)(w + p.off_stats);
        RF_HIP(hipMemsetAsync(stats, 0, 8 * sizeof(unsigned long long), s));
    }
    RF_LAUNCH("nnp_sort", nnp_sort_kernel, dim3(2 * b), dim3(STPB), 0, s, sa);
    const long g0 = (long)b * wa.groups[0], g1 = (long)b * wa.groups[1];
    wa.wg0 = wa.nw[0] == 4 ? (int)g0 : rf::ceil_div(g0, 4);
    const int wg1 = wa.nw[1] == 4 ? (int)g1 : rf::ceil_div(g1, 4);
    RF_LAUNCH("nnp_sweep", nnp_sweep_kernel, dim3(wa.wg0 + wg1), dim3(256), 0, s, wa,
              (const float4 *)sa.sorted[0], (const float4 *)sa.sorted[1], (const float4 *)sa.box16[0],
              (const float4 *)sa.box16[1], (const float4 *)sa.box64[0], (const float4 *)sa.box64[1], dist1, dist2,
              idx1, idx2, stats);
    if (stats_out) {
        RF_HIP(hipMemcpyAsync(stats_out, stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        RF_HIP(hipStreamSynchronize(s));
    }
    return RF_OK;
}

}  // namespace rfp
