// approxmatch.hip -- approx_match / match_cost / match_cost_grad (EMD) for gfx950.
//
// Replaces approxmatch, matchcost, matchcostgrad1/2 (pc_distance/tf_approxmatch.cu:1-295).
// Algorithm = the reference CUDA schedule (levels {-4^7..-4^-1, 0}; P1/P2/P3 per level, see
// SURVEY.md 3.4), same fp32 expressions (d2 = fma chain, fma accumulates, exp as
// v_exp_f32(d2 * level*log2e) -- the analogue of __expf = ex2.approx(x*log2e)).
//
// MI355X design, not the reference's one-block-per-batch-element loop:
//   * every sweep is one launch over (64-row blocks) x (batch); one row per lane; the row's
//     sweep over the other set is split over the NSEG waves of the workgroup (column segments,
//     combined in segment order).  Columns are wave-uniform, so they are STREAMED THROUGH SGPRs
//     by scalar loads one sub-chunk ahead (no LDS tile, no barrier in the loop; VALU ops take
//     the SGPR operand directly) -- same scheme as the Chamfer sweep;
//   * P3 of level v-1 and P1 of level v sweep the same rows over the same columns and P1 only
//     needs the row's own updated remainL, so they are FUSED: one distance evaluation feeds both
//     exponentials (20 + 1 launches instead of 30, 6 of 19 ops per pair saved);
//   * `match` is NOT read-modify-written once per level (the reference moves 21 x 4nm bytes
//     per sample).  The per-level ratio vectors (10 x (n+m) floats) are kept in the
//     workspace and match is produced ONCE at the end:
//         match[l][k] = fma(rl_9[k]*e_9, rr_9[l], ... fma(rl_0[k]*e_0, rr_0[l], 0))
//     which is the same fma chain, in the same level order, as the reference's 10 "+="
//     (tf_approxmatch.cu:152; App. A: the += is a fused multiply-add), so the bits do not
//     depend on this restructuring.  HBM traffic for match: one 4nm-byte write.
//   * row sums are accumulated per column segment and combined in segment order, so they
//     differ from the reference's strictly sequential order in the last bits (tolerance
//     stated in tests/test_gpu_emd.py).
#include <stdlib.h>

#include "common.hpp"
#include "nn_pruned.hpp"

namespace {

#ifndef RFA_TARGET_WAVES
#define RFA_TARGET_WAVES 4096  // waves a level sweep is cut into at least (column segments per row block)
#endif
constexpr float kSkipArg = 161.f;  // d2 * |c| >= 161 => fl(d2 * c) <= -160 => v_exp_f32 = +0 (with the product's rounding covered)
constexpr float kLog2e = 1.44269502f;  // 0x3FB8AA3B, the constant __expf multiplies by
// am_match's stores of `match` are non-temporal: the tensor (512 MiB at C4) is twice the memory-side cache and is read next by
// another launch; written through the caches it leaves that launch competing with the write-back of its own input
// (approx_match + match_cost 0.987 -> 0.956 ms same-device, am_match itself 139 -> 135.5 us)
// match_cost reads `match` with non-temporal loads too: 88.4 -> 78.6 us alone, 87.7 -> 79.7 inside the sequence (a pure read of
// 512 MiB: 83 us plain, 76 non-temporal -- tools/ubench/stream_rate.hip); the gradient pass behind it then finds less of the tensor's
// head in the memory-side cache (85.8 -> 91.2 us): -3 us for the three ops together, -8 for approx_match + match_cost
typedef float am_v2f __attribute__((ext_vector_type(2)));
constexpr int TPB = 256;
constexpr int LVG = 16;             // levels per group in the materialisation kernel
constexpr int MAX_LEVELS = 64;
constexpr int MAX_BATCH = 65535;  // the batch index is a grid y/z dimension (the reference: 32 blocks)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

typedef __attribute__((address_space(4))) float cfloat;  // wave-uniform, launch-constant operands: s_load
constexpr int SUB = 8;     // columns per scalar-load sub-chunk
constexpr int CPAD = 128;  // column counts are padded to a multiple of this (zero scalars)

// pack xyz (b,npts,3) -> (b,npad,3) zero padded, and initialise the state vectors: remain = the cloud's multiplier, and the
// PADDED entries of every vector slot (remain and each level's ratio) = 0, so that padded columns contribute e*0 = 0 to every sum
// (e <= 1 is always finite) -- the entries of real points are all written by their level's sweep before anything reads them, so
// nothing else of the state needs clearing (a memset of the whole region was 8 us per call at C4).  Both clouds in one launch.
struct AmInit {
    int npts[2], npad[2];
    float fill[2];
    const float *xyz[2];
    float *xyzp[2];
    size_t xyzp_stride[2];
    float *vec;        // the vector region: per batch element `stride` floats = nslots slots of V = npad[0] + npad[1] floats
    size_t stride, V;  // slot s: [cloud 0's vector (npad[0]) | cloud 1's (npad[1])]
    int nslots, b;
};
constexpr int AI_TPB = 1024;
__global__ __launch_bounds__(AI_TPB) void am_init_kernel(AmInit a) {
    const int bi = blockIdx.y, c = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int npts = c ? a.npts[1] : a.npts[0], npad = c ? a.npad[1] : a.npad[0];
    if (bi == 0 && c == 0 && j < 64) a.vec[(size_t)a.b * a.stride + j] = 0.f;  // the slack behind the last element's vectors
    if (j >= npad) return;
    float *v = a.vec + (size_t)bi * a.stride + (c ? a.npad[0] : 0) + j;
    if (j < npts) {
        const float *p = (c ? a.xyz[1] : a.xyz[0]) + ((size_t)bi * npts + j) * 3;
        float *q = (c ? a.xyzp[1] : a.xyzp[0]) + (size_t)bi * (c ? a.xyzp_stride[1] : a.xyzp_stride[0]) + (size_t)j * 3;
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
        v[0] = c ? a.fill[1] : a.fill[0];
    } else {
        float *q = (c ? a.xyzp[1] : a.xyzp[0]) + (size_t)bi * (c ? a.xyzp_stride[1] : a.xyzp_stride[0]) + (size_t)j * 3;
        q[0] = q[1] = q[2] = 0.f;
        for (int s = 0; s < a.nslots; s++) v[(size_t)s * a.V] = 0.f;
    }
}

// Rows = xyz1 points k (one per lane), columns = xyz2 points l streamed through SGPRs.
//   HAS_P3: acc3 = sum_l fma(ratioL_prev[k]*e(c_prev), ratioR_prev[l], .)   (P3 of the previous level)
//           remainL[k] = max(0, remainL[k] - acc3)
//   HAS_P1: acc1 = 1e-9 + sum_l fma(e(c_cur), remainR[l], .)                 (P1 of this level)
//           ratioL_out[k] = remainL[k] / acc1
// P1: 0 = absent, 1 = present, 2 = present at a level whose multiplier is 0 (the reference's last
// level): e = exp2(d2*0) = 1.0 exactly, so the exponential (and, without P3, the distance) is not
// evaluated; fma(1.0, s, acc) rounds exactly as before -> the same bits for fewer instructions;
// 3 = present at the SAME multiplier as the P3 it is fused with (a schedule that repeats a level, e.g. the
// 50-level schedule of BASELINE configs[3]): the two exponentials have the same argument, one is evaluated -- same bits.
// SKIP (round 5, the sharp levels): the lanes' rows are taken in the SPATIAL order of their cloud (`perm`: position in the
// sort-tile-recursive order -> original row, -1 for padding: rfp::Sorted::orig), so a wave's 64 * RPT rows sit in a small box,
// and a column whose d2 to every one of them is >= tskip is skipped after the distance: its weights would be exactly 0
// (tskip is the d2 beyond which v_exp_f32 returns +0 for the LESS sharp of the launch's levels), so every sum keeps its
// bits -- the row order does not enter any sum, the column order is untouched.  At C4 a wave skips the exponentials of
// 91 / 81 / 58 % of its columns at levels -4^7 / -4^6 / -4^5 (28 / 84 / 100 % would pass with rows in input order).
// (Finite inputs: a NaN distance is not "near", so where the plain sweep would spread a NaN coordinate's NaN over a sum, this one
// may leave the sum finite -- EMD of non-finite clouds is NaN garbage in the reference too and no caller's contract; the `<`
// form of the test is 3 % of the launch cheaper than the NaN-keeping `!(>=)`, measured.)
// SKIP = 1: a column is dropped when it is beyond `tskip` (this level's cut-off) of every row; the fused P3 of the previous,
//           sharper level is evaluated only for columns within its own cut-off `tskip_prev` (<= tskip) of some row;
// SKIP = 2: this level is too broad to drop columns (every weight is evaluated) but the fused P3's level is not: only its part
//           is conditional.  Both tests are wave-uniform branches on a ballot.
// MASK (with SKIP = 1): 1 = this launch also LISTS, per wave, the columns that are within `tmask` -- the NEXT level's cut-off -- of
//           some row of the wave (their indices in column order, and the count); 2 = this launch visits only the columns an earlier
//           launch of the same grid listed (the geometry does not change between the phases of a call): at level -4^6 a wave keeps a
//           fifth of its columns, and computing and testing the distances of all of them again was what its skipping sweep still
//           cost.  The listed columns' operands are gathered by the lanes (one column per lane, one memory round trip for 64
//           columns), parked in LDS and read back as broadcasts: same operations per kept column, same order -- same bits.
// CMP (round 6): the columns are the sample's LIVE ones only -- am_compact_kernel's packed copy of the columns whose scalars are
// not both exactly 0 (xyz2p / ratioR_prev / remainR then point at that copy, `cstride` floats per sample; counts[bi * 4] = how
// many, a multiple of 16 with zero-scalar padding).  A column whose scalars are +0 adds fma(e, 0, acc) = acc: leaving it out
// changes no sum.  The live columns are cut into the workgroup's segments evenly (a device-side count: no host round trip).
template <bool HAS_P3, int P1, int RPT, int SKIP = 0, int MASK = 0, int CMP = 0>
__global__ __launch_bounds__(1024) void am_rowk_kernel(
    int n, int seglen, const float *__restrict__ xyz1, const float *__restrict__ xyz2p,
    size_t xyz2p_stride, const float *__restrict__ ratioR_prev, const float *__restrict__ remainR,
    const float *__restrict__ ratioL_prev, float *__restrict__ remainL,
    float *__restrict__ ratioL_out, size_t stride, float c_prev, float c_cur, const int *__restrict__ perm,
    int perm_stride, float tskip, float tskip_prev, const int *__restrict__ guard, int guard_want,
    unsigned short *__restrict__ mask = nullptr, float tmask = 0.f, const int *__restrict__ counts = nullptr,
    size_t cstride = 0, const float *__restrict__ rowbox = nullptr) {
    static_assert(MASK == 0 || SKIP == 1, "column lists belong to the skipping sweeps");
    static_assert(CMP == 0 || (SKIP == 0 && MASK == 0), "the live-column form is a dense sweep over a shorter column set");
    __shared__ float part3[16][64 * RPT], part1[16][64 * RPT];
    // (guard: an optional device word that switches the launch off -- unused by the current routes)
    if (guard && (*guard != 0) != (guard_want != 0)) return;
    static_assert(SKIP == 0 || P1 == 1, "the skipping sweeps evaluate this level's own exponential");
    // (a sample's workgroups on ONE XCD, rf::xcd_contiguous: its columns are then read from HBM by one L2 instead of eight)
    const unsigned lgc = rf::xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bi = lgc / gridDim.x, bx = lgc - bi * gridDim.x;
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool HAS_P1 = P1 != 0;
    const int nseg = blockDim.x >> 6;
    const float *__restrict__ A = xyz1 + (size_t)bi * n * 3;
    float x1[RPT], y1[RPT], z1[RPT], rl[RPT], rem0[RPT], acc3[RPT], acc1[RPT];
    int krow[RPT];  // the lane's rows (original indices; < 0: none)
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const int pos = bx * 64 * RPT + r * 64 + lane;
        krow[r] = SKIP != 0 ? (pos < perm_stride ? perm[(size_t)bi * perm_stride + pos] : -1) : (pos < n ? pos : -1);
        const int kk = krow[r] >= 0 ? krow[r] : n - 1;
        x1[r] = A[kk * 3]; y1[r] = A[kk * 3 + 1]; z1[r] = A[kk * 3 + 2];
        if (SKIP != 0 && krow[r] < 0) x1[r] = INFINITY;  // a lane without a row never keeps a column alive (d2 = inf)
        rl[r] = HAS_P3 ? ratioL_prev[(size_t)bi * stride + kk] : 0.f;
        rem0[r] = remainL[(size_t)bi * stride + kk];  // (asked for here, used behind the sweep: read behind the barrier it was a round trip to memory at the launch's very end)
        acc3[r] = 0.f;
        acc1[r] = (seg == 0) ? 1e-9f : 0.f;
    }
    const float *__restrict__ C = xyz2p + (size_t)bi * xyz2p_stride;
    const float *__restrict__ S3 = ratioR_prev + (size_t)bi * (CMP ? cstride : stride);
    const float *__restrict__ S1 = remainR + (size_t)bi * (CMP ? cstride : stride);
    int c0 = seg * seglen, c1 = c0 + seglen;  // multiples of SUB, inside the padded range
    if (CMP) {  // (uniform) this wave's share of the live columns
        const int cnt = ((const __attribute__((address_space(4))) int *)counts)[bi * 4];  // a multiple of 2 * SUB
        const int slen = ((cnt + nseg - 1) / nseg + 2 * SUB - 1) / (2 * SUB) * (2 * SUB);
        c0 = min(seg * slen, cnt);
        c1 = min(c0 + slen, cnt);
    }
    // one column (its coordinates and scalars wave-uniform) against the lane's RPT rows
    // (PK: the lane's two rows as the halves of packed fp32 operations, as in am_rowl_kernel -- bit-identical sums -- together
    // with the column operands through two scalar register sets in turn: the dense fused P3 + P1 sweep 61.8 -> 47.5 us at C4, P3
    // alone 41.6 -> 33; packed alone the fused sweep got 6 % SLOWER, the two register sets alone 3 %.  The skipping sweeps keep
    // the scalar form: most of their columns end after the distance and its test, packed 41.6 / 53.4 for 35.9 / 47.1 us.)
    constexpr bool PK = RPT == 2 && SKIP == 0;
    am_v2f X1 = {x1[0], x1[RPT - 1]}, Y1 = {y1[0], y1[RPT - 1]}, Z1 = {z1[0], z1[RPT - 1]}, RL = {rl[0], rl[RPT - 1]};
    am_v2f ACC3 = {acc3[0], acc3[RPT - 1]}, ACC1 = {acc1[0], acc1[RPT - 1]};
    int nlisted = 0;                       // (MASK == 1, uniform) columns listed so far
    unsigned short *__restrict__ lst = nullptr;  // (MASK != 0) this wave's list: [0] = count, then the columns
    auto column = [&](float cx, float cy, float cz, float s3u, float s1u, int cidx = 0) {
        if constexpr (PK) {
            const am_v2f dx = cx - X1, dy = cy - Y1, dz = cz - Z1;
            const am_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
            auto ex2 = [](am_v2f a) { return am_v2f{fast_exp2(a.x), fast_exp2(a.y)}; };
            if (SKIP == 1) {
                if (__ballot(d2.x < tskip || d2.y < tskip) == 0ull) return;  // (uniform) every weight of this column is exactly 0 in this wave
                asm volatile("; column kept");
            }
            if (SKIP != 0 && HAS_P3) {  // the two parts under their own tests
                ACC1 = __builtin_elementwise_fma(ex2(d2 * c_cur), am_v2f{s1u, s1u}, ACC1);
                if (__ballot(d2.x < tskip_prev || d2.y < tskip_prev) != 0ull) {  // (uniform)
                    asm volatile("; previous level kept");
                    ACC3 = __builtin_elementwise_fma(RL * ex2(d2 * c_prev), am_v2f{s3u, s3u}, ACC3);
                }
                return;
            }
            am_v2f e3 = {0.f, 0.f};
            if (HAS_P3) {
                e3 = ex2(d2 * c_prev);
                ACC3 = __builtin_elementwise_fma(RL * e3, am_v2f{s3u, s3u}, ACC3);
            }
            if (HAS_P1) ACC1 = __builtin_elementwise_fma(P1 == 2 ? am_v2f{1.0f, 1.0f} : (P1 == 3 ? e3 : ex2(d2 * c_cur)), am_v2f{s1u, s1u}, ACC1);
            return;
        }
        float d2[RPT];
        bool near = false, near_prev = false;
        {
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                d2[r] = rf::d2_fma(cx - x1[r], cy - y1[r], cz - z1[r]);
                near = near || d2[r] < tskip;
                near_prev = near_prev || d2[r] < tskip_prev;
            }
        }
        if (MASK == 1) {
            // the broader test first: a column beyond the NEXT level's cut-off of every row is beyond this level's too
            bool near_mask = false;
#pragma unroll
            for (int r = 0; r < RPT; r++) near_mask = near_mask || d2[r] < tmask;
            if (__ballot(near_mask) == 0ull) return;  // (uniform)
            asm volatile("; column listed");
            if (lane == 0) lst[1 + nlisted] = (unsigned short)cidx;
            nlisted++;
        }
        if (SKIP == 1) {
            if (__ballot(near) == 0ull) return;  // (uniform) every weight of this column is exactly 0 in this wave
            asm volatile("; column kept");        // (keeps the branch a branch: see grouping.hip's hit path)
        }
        if (SKIP != 0 && HAS_P3) {  // the two parts under their own tests (static_assert above: P1 == 1 here)
#pragma unroll
            for (int r = 0; r < RPT; r++) acc1[r] = fmaf(fast_exp2(d2[r] * c_cur), s1u, acc1[r]);
            if (__ballot(near_prev) != 0ull) {  // (uniform)
                asm volatile("; previous level kept");
#pragma unroll
                for (int r = 0; r < RPT; r++) acc3[r] = fmaf(rl[r] * fast_exp2(d2[r] * c_prev), s3u, acc3[r]);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            float e3 = 0.f;
            if (HAS_P3) {
                e3 = fast_exp2(d2[r] * c_prev);
                acc3[r] = fmaf(rl[r] * e3, s3u, acc3[r]);
            }
            if (HAS_P1) acc1[r] = fmaf(P1 == 2 ? 1.0f : (P1 == 3 ? e3 : fast_exp2(d2[r] * c_cur)), s1u, acc1[r]);
        }
    };
    if constexpr (SKIP != 0 || PK) {
        // The skipping sweeps issue 14 VALU per column and lane pair against ~10 scalar instructions -- and a CU has ONE scalar
        // unit for its 16 waves: they are bound by IT (an x-only pre-test that removed 4 VALU from 50 % of the columns changed
        // nothing).  So here the column operands go through TWO scalar register sets used in turn instead of load-then-copy (5
        // s_mov per column gone).  For the plain sweeps, which are VALU-bound, the same form measured 5 % slower
        // (tools/experiments/emd_p3p1_square_pingpong.patch.txt) and they keep the copy.  The operands are cast to the constant
        // address space -- nothing in this launch writes them -- which keeps their loads on s_load in this loop shape.
        const cfloat *Cc = (const cfloat *)C, *S3c = (const cfloat *)S3, *S1c = (const cfloat *)S1;
        float xa[3 * SUB], a3[SUB], a1[SUB], xb[3 * SUB], b3[SUB], b1[SUB];
#define RFA_FETCH_K(xs, t3, t1, c)                                                           \
    do {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 3 * SUB; i++) xs[i] = Cc[(size_t)(c) * 3 + i]; \
        _Pragma("unroll") for (int i = 0; i < SUB; i++) {                                    \
            t3[i] = HAS_P3 ? S3c[(c) + i] : 0.f;                                             \
            t1[i] = HAS_P1 ? S1c[(c) + i] : 0.f;                                             \
        }                                                                                    \
    } while (0)
        // the wave's list: ((batch element, row group), column segment) -> 1 + seglen entries
        if (MASK != 0) lst = mask + (((size_t)bi * gridDim.x + bx) * nseg + seg) * (size_t)(seglen + 1);
        if constexpr (MASK == 2) {
            __shared__ float colbuf[16][64][8];  // per wave: 64 listed columns' operands (x y z s3 s1)
            float(*cb)[8] = colbuf[seg];
            const int cnt = ((const __attribute__((address_space(4))) unsigned short *)lst)[0];
            for (int base = 0; base < cnt; base += 64) {
                const int j = base + lane;
                if (j < cnt) {
                    const int c = c0 + lst[1 + j];
                    cb[lane][0] = C[(size_t)c * 3], cb[lane][1] = C[(size_t)c * 3 + 1], cb[lane][2] = C[(size_t)c * 3 + 2];
                    cb[lane][3] = HAS_P3 ? S3[c] : 0.f;
                    cb[lane][4] = HAS_P1 ? S1[c] : 0.f;
                }
                // (a wave's own LDS writes are visible to it without a barrier: in-order LDS queue)
                const int nb = min(64, cnt - base);
                int u = 0;
                for (; u + 4 <= nb; u += 4) {  // four records in flight: one LDS round trip per four columns
                    float4 v[4];
                    float w[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = *(const float4 *)cb[u + q], w[q] = cb[u + q][4];
#pragma unroll
                    for (int q = 0; q < 4; q++) column(v[q].x, v[q].y, v[q].z, v[q].w, w[q]);
                }
                for (; u < nb; u++) {
                    const float4 v = *(const float4 *)cb[u];
                    column(v.x, v.y, v.z, v.w, cb[u][4]);
                }
            }
        } else if constexpr (MASK == 1) {
            // The LISTING launch (round 6, late): 64 columns at a time, one per lane, against the BOX of the wave's rows first -- the
            // two superblocks of the sorted set the rows come from (rfp::Sorted::box64; the same operations on the per-axis gaps as
            // d2 on the differences: never above the d2 of any row in the box, fps_sorted_kernel's argument) -- and only the columns
            // whose bound is inside the listing cut-off go through the exact per-row test, in column order, their operands parked
            // in LDS and read back as broadcasts (MASK == 2's gather).  At C4 a wave's box passes 2 of 5 columns; testing every
            // column against every row was 31 us of a 0.56 ms call, twice (here and in am_rowl_kernel).
            __shared__ float colbuf1[16][64][8];
            float(*cb)[8] = colbuf1[seg];
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            {
                const int nsb = perm_stride >> 6;
                const cfloat *BX = (const cfloat *)(rowbox + (size_t)bi * nsb * 8);
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int sb = bx * RPT + r;
                    if (sb < nsb) {  // (uniform)
#pragma unroll
                        for (int a = 0; a < 3; a++) blo[a] = fminf(blo[a], BX[sb * 8 + a]), bhi[a] = fmaxf(bhi[a], BX[sb * 8 + 4 + a]);
                    }
                }
            }
            for (int base = c0; base < c1; base += 64) {
                const int j = base + lane;
                const bool in = j < c1;
                const int jj = in ? j : c0;
                const float cx = C[(size_t)jj * 3], cy = C[(size_t)jj * 3 + 1], cz = C[(size_t)jj * 3 + 2];
                const float s3v = HAS_P3 ? S3[jj] : 0.f, s1v = HAS_P1 ? S1[jj] : 0.f;
                const float gx = fmaxf(fmaxf(blo[0] - cx, cx - bhi[0]), 0.f), gy = fmaxf(fmaxf(blo[1] - cy, cy - bhi[1]), 0.f),
                            gz = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.f);
                unsigned long long M = __ballot(in && rf::d2_fma(gx, gy, gz) < tmask);
                if (M == 0ull) continue;  // (uniform)
                *(float4 *)cb[lane] = make_float4(cx, cy, cz, s3v);
                cb[lane][4] = s1v;
                // (a wave's own LDS writes are visible to it without a barrier: in-order LDS queue)
                while (M != 0ull) {  // four records in flight: one LDS round trip per four columns
                    int uu[4], nb = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (M != 0ull) {
                            uu[q] = __builtin_ctzll(M);
                            M &= M - 1ull;
                            nb = q + 1;
                        } else {
                            uu[q] = uu[0];
                        }
                    }
                    float4 v[4];
                    float w[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = *(const float4 *)cb[uu[q]], w[q] = cb[uu[q]][4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < nb) column(v[q].x, v[q].y, v[q].z, v[q].w, w[q], base - c0 + uu[q]);
                }
            }
            if (lane == 0) lst[0] = (unsigned short)nlisted;
        } else {
        RFA_FETCH_K(xa, a3, a1, c0);
        for (int c = c0; c < c1; c += 2 * SUB) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set a has arrived
            __builtin_amdgcn_sched_barrier(0);
            RFA_FETCH_K(xb, b3, b1, c + SUB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(xa[u * 3], xa[u * 3 + 1], xa[u * 3 + 2], a3[u], a1[u], c - c0 + u);
            if (c + SUB >= c1) break;
            __builtin_amdgcn_s_waitcnt(0xC07F);  // set b has arrived
            __builtin_amdgcn_sched_barrier(0);
            RFA_FETCH_K(xa, a3, a1, c + 2 * SUB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(xb[u * 3], xb[u * 3 + 1], xb[u * 3 + 2], b3[u], b1[u], c + SUB - c0 + u);
        }
        }
#undef RFA_FETCH_K
    } else {
        float nb[3 * SUB], n3[SUB], n1[SUB];
#pragma unroll
        for (int i = 0; i < 3 * SUB; i++) nb[i] = C[(size_t)c0 * 3 + i];
#pragma unroll
        for (int i = 0; i < SUB; i++) {
            n3[i] = HAS_P3 ? S3[c0 + i] : 0.f;
            n1[i] = HAS_P1 ? S1[c0 + i] : 0.f;
        }
        for (int c = c0; c < c1; c += SUB) {
            float cb[3 * SUB], s3[SUB], s1[SUB];
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): retire the previous prefetch first
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3 * SUB; i++) cb[i] = nb[i];
#pragma unroll
            for (int i = 0; i < SUB; i++) { s3[i] = n3[i]; s1[i] = n1[i]; }
#pragma unroll
            for (int i = 0; i < 3 * SUB; i++) nb[i] = C[(size_t)(c + SUB) * 3 + i];
#pragma unroll
            for (int i = 0; i < SUB; i++) {
                n3[i] = HAS_P3 ? S3[c + SUB + i] : 0.f;
                n1[i] = HAS_P1 ? S1[c + SUB + i] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(cb[u * 3], cb[u * 3 + 1], cb[u * 3 + 2], s3[u], s1[u]);
        }
    }
    if constexpr (PK) {
        acc3[0] = ACC3.x, acc3[RPT - 1] = ACC3.y;
        acc1[0] = ACC1.x, acc1[RPT - 1] = ACC1.y;
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        part3[seg][r * 64 + lane] = acc3[r];
        part1[seg][r * 64 + lane] = acc1[r];
    }
    __syncthreads();
    if (seg == 0) {
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int k = krow[r];
            if (k < 0) continue;
            float t3 = part3[0][r * 64 + lane], t1 = part1[0][r * 64 + lane];
            for (int g = 1; g < nseg; g++) {
                t3 += part3[g][r * 64 + lane];
                t1 += part1[g][r * 64 + lane];
            }
            float rem = rem0[r];
            if (HAS_P3) {
                rem = fmaxf(0.0f, rem - t3);
                remainL[(size_t)bi * stride + k] = rem;
            }
            if (HAS_P1) ratioL_out[(size_t)bi * stride + k] = rem / t1;
        }
    }
}

// P2: rows = xyz2 points l, columns = xyz1 points k with scalar ratioL[k].
//   sumr = sum_k fma(e, ratioL[k], .);  t = sumr*remainR[l];  cons = min(remainR[l]/(t+1e-9), 1)
//   ratioR[l] = remainR[l]*cons;  remainR[l] = max(0, remainR[l]-t)
// ZERO: the level's multiplier is 0 -> e = 1.0 exactly, no distance and no exponential (see am_rowk).
// LIST (round 6): the rows are the sample's LIVE ones only -- `perm` is am_compact_kernel's list of the rows whose remainR is not
// exactly 0 (in index order, -1 behind the last; counts[bi * 4 + 1] of them).  A row with remainR = +0 has t = sumr * 0 = 0,
// cons = min(0 / 1e-9, 1) = 0, ratioR = 0 and stays at 0 whatever its sum: the compaction writes those zeros, this launch skips
// the rows.  Workgroups beyond the list leave at once.
template <int RPT, bool ZERO, bool SKIP = false, int MASK = 0, bool LIST = false>
__global__ __launch_bounds__(1024) void am_rowl_kernel(
    int m, int seglen, const float *__restrict__ xyz2, const float *__restrict__ xyz1p,
    size_t xyz1p_stride, const float *__restrict__ ratioL, float *__restrict__ remainR,
    float *__restrict__ ratioR_out, size_t stride, float c_cur, const int *__restrict__ perm, int perm_stride,
    float tskip, const int *__restrict__ guard, int guard_want, unsigned short *__restrict__ mask = nullptr,
    float tmask = 0.f, const int *__restrict__ counts = nullptr, const float *__restrict__ rowbox = nullptr) {
    static_assert(MASK == 0 || SKIP, "column lists belong to the skipping sweeps (am_rowk_kernel MASK)");
    static_assert(!LIST || (!SKIP && MASK == 0), "the live-row form is a dense sweep over fewer rows");
    if (guard && (*guard != 0) != (guard_want != 0)) return;  // (see am_rowk_kernel)
    __shared__ float part[16][64 * RPT];
    // (a sample's workgroups on ONE XCD, rf::xcd_contiguous: its columns are then read from HBM by one L2 instead of eight)
    const unsigned lgc = rf::xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bi = lgc / gridDim.x, bx = lgc - bi * gridDim.x;
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nseg = blockDim.x >> 6;
    if (LIST && bx * 64 * RPT >= ((const __attribute__((address_space(4))) int *)counts)[bi * 4 + 1]) return;  // (uniform)
    const float *__restrict__ B = xyz2 + (size_t)bi * m * 3;
    float x2[RPT], y2[RPT], z2[RPT], rem0[RPT], acc[RPT];
    int lrow[RPT];  // the lane's rows (original indices; < 0: none); SKIP: in the cloud's spatial order (see am_rowk_kernel)
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const int pos = bx * 64 * RPT + r * 64 + lane;
        lrow[r] = (SKIP || LIST) ? (pos < perm_stride ? perm[(size_t)bi * perm_stride + pos] : -1) : (pos < m ? pos : -1);
        const int ll = lrow[r] >= 0 ? lrow[r] : m - 1;
        x2[r] = B[ll * 3]; y2[r] = B[ll * 3 + 1]; z2[r] = B[ll * 3 + 2];
        if (SKIP && lrow[r] < 0) x2[r] = INFINITY;
        rem0[r] = remainR[(size_t)bi * stride + ll];  // (asked for here, used behind the sweep: am_rowk_kernel)
        acc[r] = 0.f;
    }
    const float *__restrict__ C = xyz1p + (size_t)bi * xyz1p_stride;
    const float *__restrict__ S = ratioL + (size_t)bi * stride;
    const int c0 = seg * seglen, c1 = c0 + seglen;
    // PKL: the lane's two rows as the two halves of packed fp32 operations (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE
    // operations in the same order -- bit-identical sums): 6 instead of 16 vector instructions per column beside the exponentials
    constexpr bool PKL = RPT == 2 && !SKIP;
    am_v2f X2 = {x2[0], x2[RPT - 1]}, Y2 = {y2[0], y2[RPT - 1]}, Z2 = {z2[0], z2[RPT - 1]}, ACC = {0.f, 0.f};
    int nlisted = 0;                            // (MASK == 1, uniform)
    unsigned short *__restrict__ lst = nullptr;  // (MASK != 0) this wave's list: [0] = count, then the columns
    auto column = [&](float cx, float cy, float cz, float su, int cidx = 0) {
        if constexpr (PKL) {
            const am_v2f dx = X2 - cx, dy = Y2 - cy, dz = Z2 - cz;
            const am_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
            if (SKIP) {
                if (__ballot(d2.x < tskip || d2.y < tskip) == 0ull) return;  // (uniform)
                asm volatile("; column kept");
            }
            am_v2f e = {1.0f, 1.0f};
            if (!ZERO) {
                const am_v2f a = d2 * c_cur;
                e = am_v2f{fast_exp2(a.x), fast_exp2(a.y)};
            }
            ACC = __builtin_elementwise_fma(e, am_v2f{su, su}, ACC);
            return;
        }
        float d2[RPT];
        bool near = false;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            d2[r] = rf::d2_fma(x2[r] - cx, y2[r] - cy, z2[r] - cz);
            near = near || d2[r] < tskip;
        }
        if (MASK == 1) {  // (the broader test first: am_rowk_kernel)
            bool near_mask = false;
#pragma unroll
            for (int r = 0; r < RPT; r++) near_mask = near_mask || d2[r] < tmask;
            if (__ballot(near_mask) == 0ull) return;  // (uniform)
            asm volatile("; column listed");
            if (lane == 0) lst[1 + nlisted] = (unsigned short)cidx;
            nlisted++;
        }
        if (SKIP) {
            if (__ballot(near) == 0ull) return;  // (uniform)
            asm volatile("; column kept");
        }
#pragma unroll
        for (int r = 0; r < RPT; r++) acc[r] = fmaf(ZERO ? 1.0f : fast_exp2(d2[r] * c_cur), su, acc[r]);
    };
    if constexpr (SKIP || PKL) {  // two scalar register sets in turn (am_rowk_kernel)
        const cfloat *Cc = (const cfloat *)C, *Sc = (const cfloat *)S;
        float xa[3 * SUB], sa[SUB], xb[3 * SUB], sb[SUB];
#define RFA_FETCH_L(xs, ts, c)                                                               \
    do {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 3 * SUB; i++) xs[i] = Cc[(size_t)(c) * 3 + i]; \
        _Pragma("unroll") for (int i = 0; i < SUB; i++) ts[i] = Sc[(c) + i];                 \
    } while (0)
        if (MASK != 0) lst = mask + (((size_t)bi * gridDim.x + bx) * nseg + seg) * (size_t)(seglen + 1);
        if constexpr (MASK == 2) {  // only the listed columns, gathered 64 at a time through LDS (am_rowk_kernel)
            __shared__ float colbuf[16][64][4];
            float(*cb)[4] = colbuf[seg];
            const int cnt = ((const __attribute__((address_space(4))) unsigned short *)lst)[0];
            for (int base = 0; base < cnt; base += 64) {
                const int j = base + lane;
                if (j < cnt) {
                    const int c = c0 + lst[1 + j];
                    *(float4 *)cb[lane] = make_float4(C[(size_t)c * 3], C[(size_t)c * 3 + 1], C[(size_t)c * 3 + 2], S[c]);
                }
                const int nb = min(64, cnt - base);
                int u = 0;
                for (; u + 4 <= nb; u += 4) {
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = *(const float4 *)cb[u + q];
#pragma unroll
                    for (int q = 0; q < 4; q++) column(v[q].x, v[q].y, v[q].z, v[q].w);
                }
                for (; u < nb; u++) {
                    const float4 v = *(const float4 *)cb[u];
                    column(v.x, v.y, v.z, v.w);
                }
            }
        } else if constexpr (MASK == 1) {  // the listing launch: the columns against the box of the wave's rows first (am_rowk_kernel)
            __shared__ float colbuf1[16][64][4];
            float(*cb)[4] = colbuf1[seg];
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            {
                const int nsb = perm_stride >> 6;
                const cfloat *BX = (const cfloat *)(rowbox + (size_t)bi * nsb * 8);
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int sb = bx * RPT + r;
                    if (sb < nsb) {  // (uniform)
#pragma unroll
                        for (int a = 0; a < 3; a++) blo[a] = fminf(blo[a], BX[sb * 8 + a]), bhi[a] = fmaxf(bhi[a], BX[sb * 8 + 4 + a]);
                    }
                }
            }
            for (int base = c0; base < c1; base += 64) {
                const int j = base + lane;
                const bool in = j < c1;
                const int jj = in ? j : c0;
                const float cx = C[(size_t)jj * 3], cy = C[(size_t)jj * 3 + 1], cz = C[(size_t)jj * 3 + 2], sv = S[jj];
                const float gx = fmaxf(fmaxf(blo[0] - cx, cx - bhi[0]), 0.f), gy = fmaxf(fmaxf(blo[1] - cy, cy - bhi[1]), 0.f),
                            gz = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.f);
                unsigned long long M = __ballot(in && rf::d2_fma(gx, gy, gz) < tmask);
                if (M == 0ull) continue;  // (uniform)
                *(float4 *)cb[lane] = make_float4(cx, cy, cz, sv);
                while (M != 0ull) {
                    int uu[4], nb = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (M != 0ull) {
                            uu[q] = __builtin_ctzll(M);
                            M &= M - 1ull;
                            nb = q + 1;
                        } else {
                            uu[q] = uu[0];
                        }
                    }
                    float4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = *(const float4 *)cb[uu[q]];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (q < nb) column(v[q].x, v[q].y, v[q].z, v[q].w, base - c0 + uu[q]);
                }
            }
            if (lane == 0) lst[0] = (unsigned short)nlisted;
        } else {
        RFA_FETCH_L(xa, sa, c0);
        for (int c = c0; c < c1; c += 2 * SUB) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            RFA_FETCH_L(xb, sb, c + SUB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(xa[u * 3], xa[u * 3 + 1], xa[u * 3 + 2], sa[u], c - c0 + u);
            if (c + SUB >= c1) break;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            RFA_FETCH_L(xa, sa, c + 2 * SUB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(xb[u * 3], xb[u * 3 + 1], xb[u * 3 + 2], sb[u], c + SUB - c0 + u);
        }
        }
#undef RFA_FETCH_L
    } else {
        float nb[3 * SUB], ns[SUB];
#pragma unroll
        for (int i = 0; i < 3 * SUB; i++) nb[i] = C[(size_t)c0 * 3 + i];
#pragma unroll
        for (int i = 0; i < SUB; i++) ns[i] = S[c0 + i];
        for (int c = c0; c < c1; c += SUB) {
            float cb[3 * SUB], sc[SUB];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3 * SUB; i++) cb[i] = nb[i];
#pragma unroll
            for (int i = 0; i < SUB; i++) sc[i] = ns[i];
#pragma unroll
            for (int i = 0; i < 3 * SUB; i++) nb[i] = C[(size_t)(c + SUB) * 3 + i];
#pragma unroll
            for (int i = 0; i < SUB; i++) ns[i] = S[c + SUB + i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < SUB; u++) column(cb[u * 3], cb[u * 3 + 1], cb[u * 3 + 2], sc[u]);
        }
    }
    if constexpr (PKL) {
        acc[0] = ACC.x;
        acc[RPT - 1] = ACC.y;
    }
#pragma unroll
    for (int r = 0; r < RPT; r++) part[seg][r * 64 + lane] = acc[r];
    __syncthreads();
    if (seg == 0) {
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int l = lrow[r];
            if (l < 0) continue;
            float sumr = part[0][r * 64 + lane];
            for (int g = 1; g < nseg; g++) sumr += part[g][r * 64 + lane];
            const float rem = rem0[r];
            const float t = sumr * rem;
            const float cons = fminf(rem / (t + 1e-9f), 1.0f);
            ratioR_out[(size_t)bi * stride + l] = rem * cons;
            remainR[(size_t)bi * stride + l] = fmaxf(0.0f, rem - t);
        }
    }
}

// The LIVE columns and rows of set 2 (round 6).  P2 sets remainR[l] = max(0, remainR[l] - t): a column whose demand reached its
// supply is EXACTLY +0 from then on -- 44 % of C4's columns after two levels, 68 / 80 / 87 / 92 / 97 % after three .. seven -- and,
// with it, ratioR of every later level.  What the schedule still computes for such a column is fma(e, +0, acc) = acc in P1 / P3
// and a row of zeros in P2.  From level vC on the sweeps therefore run over PACKED SETS of the live points of set 2, kept in
// index order (so every sum is over the same terms in the same order whatever the batch around the sample):
//   struct of arrays per sample: cc (x, y, z), s3 = ratioR of the level, s1 = remainR, rows = original index, cnt[0] = entries
//   padded to a multiple of 16 with zero-scalar entries (+ one sub-chunk of prefetch slack), cnt[1] = entries.
// am_compact_kernel makes the first two sets after P2 of level vC - 1 (the last level swept the old way):
//   set `cols`: the columns with ratioR != 0, for the fused P3(vC - 1) + P1(vC) sweep (am_rowk_kernel CMP);
//   set `live`: the rows with remainR != 0 (a subset), which am_p2_live_kernel of level vC takes its rows from;
// and writes ratioR = 0 at every later level for the rows that are dead already.  From then on am_p2_live_kernel maintains
// the sets itself (each level's P2 packs its own rows, with their new scalars, into the other buffer): no launch in between.
struct LiveSet {
    float *cc, *s3, *s1;  // per sample: cstride * 3 / cstride / cstride floats
    int *rows;            // per sample: cstride ints
    int *cnt;             // per sample: 4 ints
};
constexpr int CK_TPB = 1024;
__global__ __launch_bounds__(CK_TPB) void am_compact_kernel(int m, const float *__restrict__ xyz2p, size_t xyz2p_stride,
                                                            const float *__restrict__ ratioR, const float *__restrict__ remainR,
                                                            float *__restrict__ ratioR_later, size_t lv_stride, int nlater,
                                                            size_t stride, LiveSet cols, LiveSet live, size_t cstride) {
    // The sample in ROWS of CK_TPB consecutive entries, an entry per thread: loads and -- the survivors of a wave land side by
    // side -- stores of whole cache lines.  (A thread owning consecutive entries, as this kernel first had it, scatters every store
    // of a wave over 64 lines: 9 us at 2048 points, 110 us at 16384 where four workgroups do all of it.)  An entry's place in a
    // set = the set's count before this row + the waves before this one (LDS, double-buffered: one barrier per row) + the lanes
    // before this one (ballot).
    __shared__ unsigned wsc[2][CK_TPB / 64], wsr[2][CK_TPB / 64];
    const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *__restrict__ C = xyz2p + (size_t)bi * xyz2p_stride;
    const float *__restrict__ RR = ratioR + (size_t)bi * stride;
    const float *__restrict__ RM = remainR + (size_t)bi * stride;
    float *__restrict__ CC = cols.cc + (size_t)bi * cstride * 3, *__restrict__ CS3 = cols.s3 + (size_t)bi * cstride,
                        *__restrict__ CS1 = cols.s1 + (size_t)bi * cstride;
    float *__restrict__ LC = live.cc + (size_t)bi * cstride * 3, *__restrict__ LS1 = live.s1 + (size_t)bi * cstride;
    int *__restrict__ LR = live.rows + (size_t)bi * cstride;
    unsigned tc = 0, tr = 0;  // (uniform) the sets' counts so far
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int row0 = 0, it = 0; row0 < m; row0 += CK_TPB, it++) {
        const int l = row0 + tid;
        const bool in = l < m;
        const int ll = in ? l : 0;
        const float rr = in ? RR[ll] : 0.f, rm = in ? RM[ll] : 0.f;
        const float x = C[(size_t)ll * 3], y = C[(size_t)ll * 3 + 1], z = C[(size_t)ll * 3 + 2];
        const unsigned long long bc = __ballot(rr != 0.f), br = __ballot(rm != 0.f);
        const int buf = it & 1;
        if (lane == 0) wsc[buf][wave] = (unsigned)__builtin_popcountll(bc), wsr[buf][wave] = (unsigned)__builtin_popcountll(br);
        __syncthreads();
        unsigned pc = tc + (unsigned)__builtin_popcountll(bc & below), pr = tr + (unsigned)__builtin_popcountll(br & below);
#pragma unroll
        for (int w = 0; w < CK_TPB / 64; w++) {
            const unsigned c = wsc[buf][w], r = wsr[buf][w];
            pc += w < wave ? c : 0u, pr += w < wave ? r : 0u;
            tc += c, tr += r;
        }
        if (rr != 0.f) {
            CC[pc * 3] = x, CC[pc * 3 + 1] = y, CC[pc * 3 + 2] = z;
            CS3[pc] = rr;
            CS1[pc] = rm;
        }
        if (rm != 0.f) {
            LC[pr * 3] = x, LC[pr * 3 + 1] = y, LC[pr * 3 + 2] = z;
            LS1[pr] = rm;
            LR[pr] = l;
        } else if (in) {
            for (int u = 0; u < nlater; u++) ratioR_later[(size_t)u * lv_stride + (size_t)bi * stride + l] = 0.f;
        }
    }
    const unsigned tcp = (tc + 2 * SUB - 1) / (2 * SUB) * (2 * SUB);
    for (unsigned e = tc + tid; e < tcp + SUB; e += CK_TPB) {  // zero-scalar columns: the padding and the sweep's prefetch slack
        CC[e * 3] = CC[e * 3 + 1] = CC[e * 3 + 2] = 0.f;
        CS3[e] = CS1[e] = 0.f;
    }
    if (tid == 0) {
        cols.cnt[bi * 4] = (int)tcp;
        cols.cnt[bi * 4 + 1] = (int)tc;
        live.cnt[bi * 4] = (int)((tr + 2 * SUB - 1) / (2 * SUB) * (2 * SUB));
        live.cnt[bi * 4 + 1] = (int)tr;
    }
}

// P2 of a level over the LIVE rows of set 2 (round 6; formulas: am_rowl_kernel).  Few rows are left at these levels (56 % of C4's
// at level 2, 8 % at level 6, 0.3 % at the last): with a row per lane and the columns through scalar registers (am_rowl_kernel)
// what is left is a handful of long waves, each a chain of scalar-load round trips nothing covers -- 18 us per launch however few
// the rows.  Here the roles are turned round: a work item is PL_ROWS = 32 consecutive live rows, parked in LDS (few rows per
// item: a launch with 70 rows left per sample still spreads over 96 CUs); each of its waves owns a
// slice of the COLUMNS (set 1), PL_CPL per lane at a time in registers, walks the workgroup's PL_ROWS rows over them, reduces each row's terms
// across the wave (DPP) and keeps row j's running sum in lane j.  Work ~ live rows x columns, in waves of equal length.
//   prologue   the workgroup finds its rows itself: they are the survivors (s1 != 0) no. [32 bx, 32 bx + 32) of the previous
//              level's packed set `in` (a scan of its s1 array: <= m floats) -- no compaction launch between the levels;
//   epilogue   ratioR_v / remainR to the full vectors, and the rows with their new scalars into the other packed set `out`,
//              which the fused P3(v) + P1(v + 1) sweep (am_rowk_kernel CMP) and the next level's P2 read; a row whose remainR
//              reaches +0 here writes ratioR = 0 for every later level itself.
// Sums: per lane over its columns in index order, then across the lanes, then across the waves in order -- another order than
// am_rowl_kernel's (tolerance, not bits: tests/test_gpu_emd.py), the same for a sample whatever its batch.
// (same device, C4: 16 / 32 / 64 rows per item 0.642 / 0.614 / 0.634 ms per approx_match + match_cost -- an item's prologue is two
// round trips to memory whatever its rows; 512 threads x 4 columns per lane, unpacked: 0.671; packed 0.597-0.604 against 0.574-0.580;
// the kernel's 134 registers (three waves per SIMD) held to 128 (four): +-0; one row's chain at a time: am_p2 184 against 175 us)
constexpr int PL_TPB = 256, PL_NW = PL_TPB / 64, PL_ROWS = 32, PL_CPL = 8, PL_RU = 2;  // (PL_RU rows' chains side by side)
// An item walks its rows over ALL columns: at 16384 columns an item of 32 rows is 46 us long however few items a late level has
// left (`tools/experiments/trace_emd.sh big`) -- clouds of more than PL_BIG_N points take their rows in items of PL_ROWS_BIG.
constexpr int PL_ROWS_BIG = 8, PL_BIG_N = 4096;
static_assert(PL_CPL % 2 == 0, "the columns of a lane go through packed fp32 operations in pairs");
static_assert(PL_ROWS % PL_RU == 0 && PL_ROWS <= 64 && PL_ROWS_BIG % PL_RU == 0, "a row per lane of the running sums");
template <bool ZERO, int ROWS = PL_ROWS>
__global__ __launch_bounds__(PL_TPB) void am_p2_live_kernel(int npad, const float *__restrict__ xyz1p, size_t xyz1p_stride,
                                                            const float *__restrict__ ratioL, float *__restrict__ remainR,
                                                            float *__restrict__ ratioR_out, size_t lv_stride, int nlater,
                                                            size_t stride, float c_cur, LiveSet in, LiveSet out, size_t cstride, int cap, int b) {
    __shared__ unsigned wsum[PL_NW];
    __shared__ int rpos[ROWS];            // this workgroup's rows: position in `in`
    __shared__ float4 rowbuf[ROWS];       // (x, y, z, remainR)
    __shared__ float part[PL_NW][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // A FIXED grid walks the (chunk of rows, sample) items, chunk-major: how many chunks a sample still has is known on the device
    // only, and one workgroup per possible chunk -- 4096 at C4, nearly all of which read their sample's count and leave -- cost a
    // round trip to memory each, four residency rounds of them: an 8 us floor under a launch with 70 rows per sample left.  The
    // counts of the items a workgroup walks past come out of the scalar cache after its first.
    const int nitems = ((cap + ROWS - 1) / ROWS) * b;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int bx = item / b, bi = item - bx * b;
    // Everything that does not depend on anything else is asked for FIRST and together -- the set's count, its s1 array (a thread's
    // entries by the set's capacity, not by the count that is still on its way), the wave's first tile of columns: the launches of
    // a level follow each other across XCDs, every one of these comes from memory (~2 us), and asked for one after the other they
    // were a 19 us floor under this kernel however few its rows.
    const int cin = ((const __attribute__((address_space(4))) int *)in.cnt)[bi * 4 + 1];  // entries of the previous set
    if (bx != 0 && bx * ROWS >= cin) continue;  // (uniform; survivors <= entries; chunk 0 always writes the new set's counts)
    // what does not depend on anything else is asked for together: the set's s1 array (a thread's entries by the set's capacity)
    // and the wave's first tile of columns -- every one of these comes from memory (~2 us: the launches of a level follow each
    // other across XCDs)
    const float *__restrict__ IS1 = in.s1 + (size_t)bi * cstride;
    const float *__restrict__ C = xyz1p + (size_t)bi * xyz1p_stride;
    const float *__restrict__ S = ratioL + (size_t)bi * stride;
    // (uniform) entries per thread: by the set's CAPACITY (mpad) while that is one batch of eight loads -- they are then asked for
    // before the count has arrived --, by the set's count beyond (clouds of more than 2048 points: with 64 entries per thread by
    // capacity, a late level's few hundred entries sat in a handful of threads that walked them in eight dependent rounds, twice:
    // a 60 us floor under every launch at 16384 points)
    const int per = cap <= 8 * PL_TPB ? (cap + PL_TPB - 1) / PL_TPB : max(1, (cin + PL_TPB - 1) / PL_TPB);
    const int e0 = tid * per;
    float sv[8];
    if (per <= 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) sv[q] = (q < per && e0 + q < cin) ? IS1[e0 + q] : 0.f;
    }
    const int slice = (npad / PL_NW + 63) / 64 * 64;  // (npad is a multiple of 128; the slices cover it, the last may be short)
    const int k0 = wave * slice, k1 = min(npad, k0 + slice);
    // (a lane's columns in PAIRS, the halves of packed fp32 operations -- v_pk_add / v_pk_mul / v_pk_fma, as am_rowl_kernel's two
    // rows: per pair of terms 8 packed instructions and two exponentials instead of 18 scalar ones)
    am_v2f cx[PL_CPL / 2], cy[PL_CPL / 2], cz[PL_CPL / 2], cs[PL_CPL / 2];
#define RFA_PL_COLS(kb)                                                                  \
    _Pragma("unroll") for (int q = 0; q < PL_CPL; q++) {                                 \
        const int k = (kb) + q * 64 + lane;                                              \
        const bool in_ = k < k1;                                                         \
        const int kk = in_ ? k : 0;                                                      \
        const float vx = C[(size_t)kk * 3], vy = C[(size_t)kk * 3 + 1], vz = C[(size_t)kk * 3 + 2]; \
        const float vs = in_ ? S[kk] : 0.f; /* (padded columns of the set read 0 by themselves: am_init) */ \
        if (q & 1) { cx[q / 2].y = vx, cy[q / 2].y = vy, cz[q / 2].y = vz, cs[q / 2].y = vs; }   \
        else       { cx[q / 2].x = vx, cy[q / 2].x = vy, cz[q / 2].x = vz, cs[q / 2].x = vs; }   \
    }
    RFA_PL_COLS(k0)
    // ---- the survivors' ranks: thread t owns `per` consecutive entries
    unsigned mine = 0, alive = 0;  // (alive: bit q = entry e0 + q survives, for per <= 8 -- sets of up to 4096 entries)
    if (per <= 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) alive |= (e0 + q < cin && sv[q] != 0.f) ? 1u << q : 0u;
        mine = (unsigned)__builtin_popcount(alive);
    } else {  // (larger sets: eight entries in flight at a time)
        for (int q0 = 0; q0 < per; q0 += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = (q0 + q < per && e0 + q0 + q < cin) ? IS1[e0 + q0 + q] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) mine += v[q] != 0.f ? 1u : 0u;
        }
    }
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    if (tid < ROWS) rpos[tid] = -1;
    __syncthreads();
    unsigned before = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < PL_NW; w++) {
        if (w < wave) before += wsum[w];
        total += wsum[w];
    }
    const int lo = bx * ROWS;
    if (lo >= (int)total) {  // (uniform) no survivor left for this workgroup ...
        if (bx == 0 && tid == 0) {  // ... (none at all: total == 0) the next sweeps see an empty set
            out.cnt[bi * 4] = 0;
            out.cnt[bi * 4 + 1] = 0;
        }
        if (bx == 0 && tid < SUB) {
            float *oc = out.cc + (size_t)bi * cstride * 3;
            oc[tid * 3] = oc[tid * 3 + 1] = oc[tid * 3 + 2] = 0.f;
            out.s3[(size_t)bi * cstride + tid] = out.s1[(size_t)bi * cstride + tid] = 0.f;
        }
        __syncthreads();  // (wsum / rpos are rewritten by the next item)
        continue;
    }
    {
        unsigned r = before;
        if (per <= 8) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (alive >> q & 1u) {
                    if ((int)r >= lo && (int)r < lo + ROWS) rpos[r - lo] = e0 + q;
                    r++;
                }
        } else if (before < (unsigned)(lo + ROWS) && before + mine > (unsigned)lo) {  // (only the threads whose entries reach into this chunk look again)
            for (int q0 = 0; q0 < per; q0 += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = (q0 + q < per && e0 + q0 + q < cin) ? IS1[e0 + q0 + q] : 0.f;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (v[q] != 0.f) {
                        if ((int)r >= lo && (int)r < lo + ROWS) rpos[r - lo] = e0 + q0 + q;
                        r++;
                    }
            }
        }
    }
    __syncthreads();
    const int nrows = min(ROWS, (int)total - lo);
    int orig = -1;
    if (tid < ROWS) {
        const int e = rpos[tid];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e >= 0) {
            const float *ic = in.cc + (size_t)bi * cstride * 3 + (size_t)e * 3;
            v = make_float4(ic[0], ic[1], ic[2], IS1[e]);
            orig = in.rows[(size_t)bi * cstride + e];
        }
        rowbuf[tid] = v;
    }
    __syncthreads();
    // ---- the sums: this wave's slice of the columns, PL_CPL per lane at a time (the first tile is in registers already)
    float racc = 0.f;  // lane j: row j's sum over this wave's columns
    for (int kb = k0; kb < k1; kb += 64 * PL_CPL) {
        if (kb != k0) {
            RFA_PL_COLS(kb)
        }
        for (int j0 = 0; j0 < nrows; j0 += PL_RU) {  // (rows behind the last hold zeros and are not kept)
            float4 rw[PL_RU];
            float a[PL_RU];
#pragma unroll
            for (int u = 0; u < PL_RU; u++) rw[u] = rowbuf[j0 + u];  // (broadcast reads)
#pragma unroll
            for (int u = 0; u < PL_RU; u++) {
                am_v2f acc = {0.f, 0.f};
                const am_v2f rx = {rw[u].x, rw[u].x}, ry = {rw[u].y, rw[u].y}, rz = {rw[u].z, rw[u].z};
#pragma unroll
                for (int q = 0; q < PL_CPL / 2; q++) {
                    const am_v2f dx = rx - cx[q], dy = ry - cy[q], dz = rz - cz[q];
                    const am_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
                    am_v2f e = {1.0f, 1.0f};
                    if (!ZERO) {
                        const am_v2f t = d2 * c_cur;
                        e = am_v2f{fast_exp2(t.x), fast_exp2(t.y)};
                    }
                    acc = __builtin_elementwise_fma(e, cs[q], acc);
                }
                a[u] = acc.x + acc.y;
            }
            // the wave's sums: row shifts, then the two row broadcasts (lane 63 holds them)
#define RFA_DPP_ADD(ctrl, rmask)                                                                                             \
    _Pragma("unroll") for (int u = 0; u < PL_RU; u++)                                                                        \
        a[u] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[u]), ctrl, rmask, 0xf, false))
            RFA_DPP_ADD(0x111, 0xf);
            RFA_DPP_ADD(0x112, 0xf);
            RFA_DPP_ADD(0x114, 0xf);
            RFA_DPP_ADD(0x118, 0xf);
            RFA_DPP_ADD(0x142, 0xa);
            RFA_DPP_ADD(0x143, 0xc);
#undef RFA_DPP_ADD
#pragma unroll
            for (int u = 0; u < PL_RU; u++) {
                const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[u]), 63));
                racc = lane == j0 + u ? racc + tot : racc;
            }
        }
    }
#undef RFA_PL_COLS
    if (lane < ROWS) part[wave][lane] = racc;
    __syncthreads();
    if (tid < ROWS && tid < nrows) {
        float sumr = part[0][tid];
#pragma unroll
        for (int w = 1; w < PL_NW; w++) sumr += part[w][tid];
        const float4 rw = rowbuf[tid];
        const float rem = rw.w;
        const float t = sumr * rem;
        const float cons = fminf(rem / (t + 1e-9f), 1.0f);
        const float ro = rem * cons, rn = fmaxf(0.0f, rem - t);
        const size_t o = (size_t)bi * stride + orig;
        ratioR_out[o] = ro;
        remainR[o] = rn;
        if (rn == 0.f)
            for (int u = 1; u <= nlater; u++) ratioR_out[o + (size_t)u * lv_stride] = 0.f;  // dead from here on
        const size_t p = (size_t)bi * cstride + lo + tid;
        float *oc = out.cc + (size_t)bi * cstride * 3 + (size_t)(lo + tid) * 3;
        oc[0] = rw.x, oc[1] = rw.y, oc[2] = rw.z;
        out.s3[p] = ro;
        out.s1[p] = rn;
        out.rows[p] = orig;
    }
    // the set's padding (zero-scalar entries up to a multiple of 16, + one sub-chunk) and its counts: the workgroup that holds its end
    if (lo + ROWS >= (int)total) {
        const int tp = ((int)total + 2 * SUB - 1) / (2 * SUB) * (2 * SUB);
        for (int e = (int)total + tid; e < tp + SUB; e += PL_TPB) {
            float *oc = out.cc + (size_t)bi * cstride * 3 + (size_t)e * 3;
            oc[0] = oc[1] = oc[2] = 0.f;
            out.s3[(size_t)bi * cstride + e] = out.s1[(size_t)bi * cstride + e] = 0.f;
        }
        if (tid == 0) {
            out.cnt[bi * 4] = tp;
            out.cnt[bi * 4 + 1] = (int)total;
        }
    }
    __syncthreads();  // (the LDS arrays are rewritten by the next item)
    }  // items
}

// match[l][k] = sum over levels (in order) of fma(ratioL_lv[k]*e_lv, ratioR_lv[l], acc).
// thread <-> k (coalesced 256-B stores per wave per l); workgroup = AMM_TPB k x LSEG l.
// The kernel sits on its STORES (512 MiB at C4; the same stores of a value that costs nothing take as long), and how long they
// take depends on the rows a thread walks, not on the workgroup's width: same device, C4, 256 threads x 64 rows (rounds 1-5) 117 us,
// x 32 rows 95, x 16 rows 106, 24 / 40 / 48 rows 109 / 101 / 108; 512 or 1024 threads x 32 rows 94-96 (a linear fill of the tensor:
// 89-95, profiles/r05_stream_rate.txt); two or four entries of a row per thread 110-121 (tools/experiments/
// am_match_entries_per_thread.patch.txt).  approx_match + match_cost 0.577 -> 0.557 ms with 1024 x 32 (profiles/r06_ab_am_match.txt).
#ifndef RFA_MATCH_TPB
#define RFA_MATCH_TPB 1024
#endif
#ifndef RFA_MATCH_LSEG
#define RFA_MATCH_LSEG 32
#endif
constexpr int LSEG = RFA_MATCH_LSEG, AMM_TPB = RFA_MATCH_TPB;
struct LevelConsts {
    float c[MAX_LEVELS];  // level * log2e
};

// NLV > 0: exactly NLV levels starting at level 0, fully unrolled with no per-level branch (the
// reference schedule is NLV = 10).
// LASTZERO: level NLV-1 has multiplier 0 (e = 1.0 exactly): its exponential is not evaluated.
// SQ: every odd level below the last has exactly four times the multiplier of the level after it (the reference
// schedule: levels -4^7 ... -4^-1, 0: tf_approxmatch.cu:36), so d2*c[v] == 4*(d2*c[v+1]) exactly and its weight is the
// next level's squared twice -- two multiplies (4.9 issue cycles) instead of a multiply and a v_exp_f32 (10.6).  Never
// a chain of two such steps: a weight is at most 5 ulp from v_exp_f32's, 3e-7 relative in a match entry that is a sum of
// such terms (the op's tolerance: rel 1e-4).  ONLY here and in emd_fused_kernel, where the weights go straight into the
// output: in the phase sweeps the same trick is 9 % faster and NOT taken -- there the weights feed remainL -= sum, whose
// cancellation amplifies 5 ulp past the tolerance (tools/experiments/emd_p3p1_square_pingpong.patch.txt).
template <int NLV, bool LASTZERO, bool SQ, int V0 = 0>
__device__ __forceinline__ void level_weights(float d2, const float (&cl)[NLV > 0 ? NLV : 1], float (&e)[NLV > 0 ? NLV : 1]) {
#pragma unroll
    for (int v = NLV - 1; v >= V0; v--) {
        if (LASTZERO && v == NLV - 1) {
            e[v] = 1.0f;
        } else if (SQ && (v & 1) && v + 1 < NLV - (LASTZERO ? 1 : 0)) {
            const float q = e[v + 1] * e[v + 1];
            e[v] = q * q;
        } else {
            e[v] = fast_exp2(d2 * cl[v]);
        }
    }
}
// the same for two pairs at once (the halves of packed fp32 operations: the same IEEE operations per half)
template <int NLV, bool LASTZERO, bool SQ, int V0 = 0>
__device__ __forceinline__ void level_weights2(am_v2f d2, const float (&cl)[NLV > 0 ? NLV : 1], am_v2f (&e)[NLV > 0 ? NLV : 1]) {
#pragma unroll
    for (int v = NLV - 1; v >= V0; v--) {
        if (LASTZERO && v == NLV - 1) {
            e[v] = am_v2f{1.0f, 1.0f};
        } else if (SQ && (v & 1) && v + 1 < NLV - (LASTZERO ? 1 : 0)) {
            const am_v2f q = e[v + 1] * e[v + 1];
            e[v] = q * q;
        } else {
            const am_v2f a = d2 * cl[v];
            e[v] = am_v2f{fast_exp2(a.x), fast_exp2(a.y)};
        }
    }
}
// does the schedule allow SQ?  (host)
inline bool quarter_chain(const float *c, int nlv, bool lastzero) {
    for (int v = 1; v + 1 < nlv - (lastzero ? 1 : 0); v += 2)
        if (c[v] != 4.0f * c[v + 1] || c[v + 1] == 0.0f) return false;
    return true;
}

template <int NLV, bool LASTZERO = false, bool SQ = false>
__global__ __launch_bounds__(AMM_TPB) void am_match_kernel(int n, int m, const float *xyz1,
                                                       const float *xyz2, const float *ratios,
                                                       size_t lv_stride, size_t b_stride, int roff,
                                                       int lv0, int nlv, LevelConsts lc,
                                                       float *match) {
    // ratios: [b][level][ (ratioL: npad) (ratioR: mpad) ]; roff = npad
    __shared__ float cxyz[LSEG][4];
    __shared__ float crr[LSEG][LVG];
    __shared__ int cdl[LSEG];  // per row: its last level with ratioR != 0
    // (a sample's workgroups on ONE XCD, rf::xcd_contiguous: the sweeps' order)
    const unsigned per = gridDim.x * gridDim.y;
    const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned lgc = rf::xcd_contiguous(lin, per * gridDim.z);
    const int bi = lgc / per;
    const unsigned rem = lgc - bi * per;
    const int by = rem / gridDim.x, bx = rem - by * gridDim.x;
    const int k = bx * AMM_TPB + threadIdx.x;
    const int l0 = by * LSEG;
    const int lcnt = min(LSEG, m - l0);
    xyz1 += (size_t)bi * n * 3;
    xyz2 += (size_t)bi * m * 3;
    ratios += (size_t)bi * b_stride + (size_t)lv0 * lv_stride;
    match += (size_t)bi * n * m;
    for (int i = threadIdx.x; i < lcnt * LVG; i += AMM_TPB) {
        int l = i / LVG, v = i % LVG;
        crr[l][v] = v < nlv ? ratios[(size_t)v * lv_stride + roff + l0 + l] : 0.f;
    }
    for (int i = threadIdx.x; i < lcnt; i += AMM_TPB) {
        cxyz[i][0] = xyz2[(size_t)(l0 + i) * 3 + 0];
        cxyz[i][1] = xyz2[(size_t)(l0 + i) * 3 + 1];
        cxyz[i][2] = xyz2[(size_t)(l0 + i) * 3 + 2];
    }
    __syncthreads();
    if (threadIdx.x < lcnt) {
        int dl = 0;
#pragma unroll
        for (int v = 1; v < LVG; v++) dl = crr[threadIdx.x][v] != 0.f ? v : dl;
        cdl[threadIdx.x] = dl;
    }
    __syncthreads();
    if (k >= n) return;
    const float x1 = xyz1[k * 3], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
    if (NLV > 0) {
        float rl[NLV > 0 ? NLV : 1], cl[NLV > 0 ? NLV : 1];
#pragma unroll
        for (int v = 0; v < NLV; v++) {
            rl[v] = ratios[(size_t)v * lv_stride + k];
            cl[v] = lc.c[v];
        }
        const float t0 = cl[0] < 0.f ? kSkipArg / -cl[0] : INFINITY;  // (uniform)
        for (int l = 0; l < lcnt; l++) {
            const float d2 = rf::d2_fma(cxyz[l][0] - x1, cxyz[l][1] - y1, cxyz[l][2] - z1);
            float e[NLV > 0 ? NLV : 1];
            float acc = 0.f;
            if (SQ && NLV == 10 && LASTZERO) {
                // (round 6) Row l's ratioR is exactly +0 from the level after its LAST LIVE one on (remainR reached +0 there:
                // am_compact_kernel) -- 44 % of C4's rows are dead after level 1, 68 % after level 2 -- and a dead level's term is
                // fma(p, +0, acc) = acc.  The weights come in pairs (an odd level's from the next even one's by two squarings), so
                // pair (v, v + 1) is formed and added only while the row lives at v: wave-uniform branches on a per-row word,
                // the same fma chain over the live levels -- same bits, 1.5 exponentials per entry instead of 4.
                const int dl = cdl[l];  // (uniform) the row's last live level
                const bool g1 = dl >= 1, g3 = dl >= 3, g5 = dl >= 5, g7 = dl >= 7;
                if (g7) { e[8] = fast_exp2(d2 * cl[8]); const float q = e[8] * e[8]; e[7] = q * q; }
                if (g5) { e[6] = fast_exp2(d2 * cl[6]); const float q = e[6] * e[6]; e[5] = q * q; }
                if (g3) { e[4] = fast_exp2(d2 * cl[4]); const float q = e[4] * e[4]; e[3] = q * q; }
                if (g1) { e[2] = fast_exp2(d2 * cl[2]); const float q = e[2] * e[2]; e[1] = q * q; }
                if (__ballot(d2 < t0) != 0ull) {
                    asm volatile("; level 0 kept");
                    acc = fmaf(rl[0] * fast_exp2(d2 * cl[0]), crr[l][0], 0.f);
                }
                if (g1) { acc = fmaf(rl[1] * e[1], crr[l][1], acc); acc = fmaf(rl[2] * e[2], crr[l][2], acc); }
                if (g3) { acc = fmaf(rl[3] * e[3], crr[l][3], acc); acc = fmaf(rl[4] * e[4], crr[l][4], acc); }
                if (g5) { acc = fmaf(rl[5] * e[5], crr[l][5], acc); acc = fmaf(rl[6] * e[6], crr[l][6], acc); }
                if (g7) {
                    acc = fmaf(rl[7] * e[7], crr[l][7], acc);
                    acc = fmaf(rl[8] * e[8], crr[l][8], acc);
                    acc = fmaf(rl[9] * 1.0f, crr[l][9], acc);
                }
            } else if (SQ) {
                // the sharpest level on its own: beyond t0 its weight is exactly +0 (v_exp_f32 returns +0 below -160), and a
                // row l is beyond t0 of ALL 64 columns of the wave in 86 % of the cases at C4 (the cut-off is 0.082) -- then
                // fma(rl * 0, rr, 0) = +0 = the accumulator's start: skipped by a wave-uniform branch, same bits
                level_weights<NLV, LASTZERO, SQ, 1>(d2, cl, e);
                if (__ballot(d2 < t0) != 0ull) {
                    asm volatile("; level 0 kept");
                    acc = fmaf(rl[0] * fast_exp2(d2 * cl[0]), crr[l][0], 0.f);
                }
#pragma unroll
                for (int v = 1; v < NLV; v++) acc = fmaf(rl[v] * e[v], crr[l][v], acc);
            } else {
                level_weights<NLV, LASTZERO, SQ>(d2, cl, e);
#pragma unroll
                for (int v = 0; v < NLV; v++) acc = fmaf(rl[v] * e[v], crr[l][v], acc);
            }
            __builtin_nontemporal_store(acc, &match[(size_t)(l0 + l) * n + k]);
        }
        return;
    }
}

// Any other schedule (up to MAX_LEVELS levels; BASELINE configs[3]: the ten reference levels five times each = 50): ALL its levels
// in ONE pass over `match` -- NG groups of LVG levels, a thread's ratioL of every level in registers (LVG * NG of them), the
// rows' ratioR in LDS.  (Rounds 1-5 took LVG levels per launch and read the tensor back for the next group: four launches and
// 3.5 GiB of traffic for the 50-level schedule at C4, 1.6 ms of its 2.9.)  The same fma chain in level order, so the same bits:
//   * a level with the multiplier of the one before it reuses its weight (the same argument: the same v_exp_f32 result) -- which
//     levels need a new one is a 64-bit word from the host;
//   * a row's levels behind its LAST LIVE one are not formed (ratioR is exactly +0 there and fma(p, +0, acc) = acc -- p is finite);
//     the guard is per group of four levels, wave-uniform.
// REP > 0: a schedule that takes every multiplier REP times in a row (configs[3]: REP = 5) -- one guard, one exponential and REP
// terms per run instead of a test per level: the branches were what the general form spent its time on (0.40 -> see below).
template <int NG, int REP = 0>
__global__ __launch_bounds__(AMM_TPB) void am_match_any_kernel(int n, int m, const float *xyz1, const float *xyz2,
                                                               const float *ratios, size_t lv_stride, size_t b_stride,
                                                               int roff, int nlv, LevelConsts lc, unsigned long long fresh,
                                                               float *match) {
    constexpr int NL = LVG * NG;
    __shared__ float cxyz[LSEG][4];
    __shared__ __attribute__((aligned(16))) float crr[LSEG][NL];
    __shared__ int cdl[LSEG];  // per row: its last level with ratioR != 0
    const unsigned per = gridDim.x * gridDim.y;
    const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned lgc = rf::xcd_contiguous(lin, per * gridDim.z);
    const int bi = lgc / per;
    const unsigned rem = lgc - bi * per;
    const int by = rem / gridDim.x, bx = rem - by * gridDim.x;
    const int k = bx * AMM_TPB + threadIdx.x;
    const int l0 = by * LSEG;
    const int lcnt = min(LSEG, m - l0);
    xyz1 += (size_t)bi * n * 3;
    xyz2 += (size_t)bi * m * 3;
    ratios += (size_t)bi * b_stride;
    match += (size_t)bi * n * m;
    for (int i = threadIdx.x; i < lcnt * NL; i += AMM_TPB) {
        const int l = i / NL, v = i % NL;
        crr[l][v] = v < nlv ? ratios[(size_t)v * lv_stride + roff + l0 + l] : 0.f;
    }
    for (int i = threadIdx.x; i < lcnt; i += AMM_TPB) {
        cxyz[i][0] = xyz2[(size_t)(l0 + i) * 3 + 0];
        cxyz[i][1] = xyz2[(size_t)(l0 + i) * 3 + 1];
        cxyz[i][2] = xyz2[(size_t)(l0 + i) * 3 + 2];
    }
    __syncthreads();
    if (threadIdx.x < lcnt) {
        int dl = 0;
        for (int v = 1; v < NL; v++) dl = crr[threadIdx.x][v] != 0.f ? v : dl;
        cdl[threadIdx.x] = dl;
    }
    __syncthreads();
    if (k >= n) return;
    const float x1 = xyz1[k * 3], y1 = xyz1[k * 3 + 1], z1 = xyz1[k * 3 + 2];
    float rl[NL];
#pragma unroll
    for (int v = 0; v < NL; v++) rl[v] = v < nlv ? ratios[(size_t)v * lv_stride + k] : 0.f;
    for (int l = 0; l < lcnt; l++) {
        const float d2 = rf::d2_fma(cxyz[l][0] - x1, cxyz[l][1] - y1, cxyz[l][2] - z1);
        const int last = min(cdl[l], nlv - 1);  // (uniform)
        float acc = 0.f, e = 0.f;
        if constexpr (REP > 0) {
            // every multiplier REP times in a row (the host checked): a run is one exponential, one guard, REP terms
#pragma unroll
            for (int g = 0; g < NL / REP; g++) {
                if (REP * g <= last) {  // (uniform) the row lives at this run's first level
                    e = fast_exp2(d2 * lc.c[REP * g]);
#pragma unroll
                    for (int u = 0; u < REP; u++) acc = fmaf(rl[REP * g + u] * e, crr[l][REP * g + u], acc);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NL / 4; g++) {
                if (4 * g <= last) {  // (uniform) the row lives at this group's first level
                    const float4 rr = *(const float4 *)&crr[l][4 * g];  // (one broadcast read for the group's four levels)
                    const float r4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int v = 4 * g + u;
                        // (uniform) a NEW weight only where the multiplier changes (`fresh`: bit v, from the host); otherwise the level
                        // before it left the same argument's v_exp_f32 in `e` -- across groups too: last >= 4 g, so group g - 1 ran
                        if (fresh >> v & 1ull) e = fast_exp2(d2 * lc.c[v]);
                        acc = fmaf(rl[v] * e, r4[u], acc);
                    }
                }
            }
        }
        __builtin_nontemporal_store(acc, &match[(size_t)(l0 + l) * n + k]);
    }
}

// ---- small clouds (max(n,m) <= AM_SMALL, e.g. the 64 x 64 EMD term of the training loss,
// vv_recon.py:489): the multi-launch pipeline above is launch-bound there (21 launches for a few
// thousand pairs), so ONE workgroup per batch element runs the whole schedule out of LDS, like
// the reference's block (tf_approxmatch.cu:13-178) -- and, summing each row strictly in index
// order, with exactly the reference's summation order.
constexpr int AM_SMALL = 256;
__global__ __launch_bounds__(AM_SMALL) void am_small_kernel(int n, int m, int nlevels, LevelConsts lc,
                                                            float multiL, float multiR,
                                                            const float *__restrict__ xyz1,
                                                            const float *__restrict__ xyz2,
                                                            float *__restrict__ match) {
    // columns are float4 {x,y,z,scalar}: one broadcast ds_read_b128 per column; the scalar slot is
    // rewritten by the owning thread before each phase (remainR / ratioL / ratioR of the level)
    __shared__ float4 c1[AM_SMALL], c2[AM_SMALL];
    __shared__ float ratL[MAX_LEVELS][AM_SMALL], ratR[MAX_LEVELS][AM_SMALL];  // 128 KiB
    const int bi = blockIdx.x, t = threadIdx.x;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f, x2 = 0.f, y2 = 0.f, z2 = 0.f;
    float remL = multiL, remR = multiR;
    if (t < n) {
        const float *p = xyz1 + ((size_t)bi * n + t) * 3;
        x1 = p[0]; y1 = p[1]; z1 = p[2];
        c1[t] = make_float4(x1, y1, z1, 0.f);
    }
    if (t < m) {
        const float *p = xyz2 + ((size_t)bi * m + t) * 3;
        x2 = p[0]; y2 = p[1]; z2 = p[2];
        c2[t] = make_float4(x2, y2, z2, remR);
    }
    __syncthreads();
    for (int v = 0; v < nlevels; v++) {
        const float c = lc.c[v];
        float rl = 0.f;
        if (t < n) {  // P1: columns l carry remainR
            float suml = 1e-9f;
#pragma unroll 4
            for (int l = 0; l < m; l++) {
                const float4 q = c2[l];
                suml = fmaf(fast_exp2(rf::d2_fma(q.x - x1, q.y - y1, q.z - z1) * c), q.w, suml);
            }
            rl = remL / suml;
            ratL[v][t] = rl;
            c1[t].w = rl;
        }
        __syncthreads();
        if (t < m) {  // P2: columns k carry ratioL
            float sumr = 0.f;
#pragma unroll 4
            for (int k = 0; k < n; k++) {
                const float4 q = c1[k];
                sumr = fmaf(fast_exp2(rf::d2_fma(x2 - q.x, y2 - q.y, z2 - q.z) * c), q.w, sumr);
            }
            const float tt = sumr * remR;
            const float cons = fminf(remR / (tt + 1e-9f), 1.0f);
            const float rr = remR * cons;
            ratR[v][t] = rr;
            c2[t].w = rr;
            remR = fmaxf(0.0f, remR - tt);
        }
        __syncthreads();
        if (v + 1 < nlevels) {  // P3 (its only effect is remainL, unused after the last level)
            if (t < n) {
                float suml = 0.f;
#pragma unroll 4
                for (int l = 0; l < m; l++) {
                    const float4 q = c2[l];
                    suml = fmaf(rl * fast_exp2(rf::d2_fma(q.x - x1, q.y - y1, q.z - z1) * c), q.w, suml);
                }
                remL = fmaxf(0.0f, remL - suml);
            }
            __syncthreads();
            if (t < m) c2[t].w = remR;  // next level's P1 scalar
            __syncthreads();
        }
    }
    if (t < n) {  // match[l][k] = level-ordered fma chain, as am_match_kernel
        float *M = match + (size_t)bi * n * m;
        for (int l = 0; l < m; l++) {
            const float4 q = c2[l];
            const float d2 = rf::d2_fma(q.x - x1, q.y - y1, q.z - z1);
            float acc = 0.f;
            for (int v = 0; v < nlevels; v++) acc = fmaf(ratL[v][t] * fast_exp2(d2 * lc.c[v]), ratR[v][l], acc);
            M[(size_t)l * n + t] = acc;
        }
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
    // The batch elements LAST first, and of each its rows last first: match_cost follows approx_match, whose am_match wrote `match`
    // (512 MiB at C4) in dispatch order, and what the 256 MB memory-side cache still holds is the END of the tensor.  Inside the
    // sequence approx_match -> match_cost this kernel takes 112 us walking backwards and 131 us walking forwards (alone in a loop:
    // 90 either way; tools/experiments/match_order.py); it ends at the head of the tensor, where match_cost_grad starts.
    const int bi = gridDim.z - 1 - blockIdx.z;
    const int l0 = (gridDim.y - 1 - blockIdx.y) * MC_L;
    const int k = blockIdx.x * TPB + threadIdx.x;
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
            sum = fmaf(d, __builtin_nontemporal_load(&match[(size_t)(l0 + l) * n + k]), sum);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        partial[((size_t)bi * gridDim.y + l0 / MC_L) * gridDim.x + blockIdx.x] = s;
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

// ---- match_cost_grad: ONE pass over match (HBM-bound: 4*B*n*m bytes) for both gradients.
//  grad1[k] = sum_l match[l][k] (x1_k - x2_l) * rsqrt(max(d2,1e-20))     (tf_approxmatch.cu:270-291)
//  grad2[l] = sum_k match[l][k] (x2_l - x1_k) * rsqrt(max(d2,1e-20))     (:229-269)
// The reference reads match twice and tree-reduces 256 threads per l.  Here a workgroup owns
// 256 k x an l-range and walks it in tiles of 32 l:
//   phase A (thread <-> k): the tile's 32 match rows are ALREADY in registers (below); q = match*rsq goes to an LDS
//            tile [32][257] (stride 257: conflict-free by rows and by columns), grad1 accumulates in registers over
//            the whole l-range;
//   phase B (thread <-> (l, 32-k slice)): re-reads q column-wise from LDS, accumulates (x2_l - x1_k)*q;
//   phase C: the 8 slices summed through LDS, one coalesced atomicAdd per (l, c).
// grad1: 3 atomics per thread at the end (LSPLIT-way contention only).  Outputs zero-filled first.
// Round 4 (profiles/r04_opbench.txt; every step measured same-device, tools/ab_mcg.py):
//   * ROLLING prefetch: a row's register is reloaded with the next tile's row as soon as phase A has consumed it (uniform
//     row base + one 32-bit lane offset, 8 rows per scheduling group): the loads are in flight through phases A, B and C
//     -- the 40 KB LDS tile holds the kernel to 4 waves per SIMD whatever the register count, so 32 rows per lane cost
//     no residency -- and the rows requested behind the last tile are the workgroup's own last row (cache hits; the
//     first prefetch form read the next workgroup's first tile: +5 % FETCH_SIZE by the counters).
//   * whole tiles (all but the last of a range whose length is no multiple of 32) take no row mask and fetch the columns'
//     coordinates by scalar loads, 8 rows = 24 dwords one group ahead; a partial tile masks rows by a multiplication and
//     reads the columns from lanes 0..31 of a clamped vector load by v_readlane.  A dead lane (k >= n) runs with
//     x1 = +inf: rsq(inf) = 0 zeroes its q.
//   * TWO barriers per tile: the next phase A only writes qs, last read before this tile's second barrier; ps is
//     rewritten behind the next tile's first barrier, which no thread passes before its own phase C is done.
// The knob version this came from (non-prefetch form, columns through LDS or v_readlane only, three barriers, timing
// ablations) is tools/experiments/mcg_knobs.patch.txt.
#ifndef RFA_MG_LSPLIT
#define RFA_MG_LSPLIT 4
#endif
#define RFA_MG_LOAD(p) __builtin_nontemporal_load(p)  // (`match` is read once: see mr_load_row)
constexpr int MG_TL = 32;
constexpr int MG_LSPLIT = RFA_MG_LSPLIT;
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void mcg_kernel(int n, int m, int lspan,
                                                  const float *__restrict__ xyz1,
                                                  const float *__restrict__ xyz2,
                                                  const float *__restrict__ match,
                                                  float *__restrict__ grad1,
                                                  float *__restrict__ grad2) {
    __shared__ float qs[MG_TL][TPB + 1];
    __shared__ float4 sx1[TPB];
    __shared__ float ps[TPB / MG_TL][MG_TL][3];
    const int bi = blockIdx.z;
    const int t = threadIdx.x;
    const int k0 = blockIdx.x * TPB;
    const int k = k0 + t;
    const bool live = k < n;
    const int kk = live ? k : n - 1;
    const float *__restrict__ A = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ B = xyz2 + (size_t)bi * m * 3;
    const float *__restrict__ M = match + (size_t)bi * n * m;
    const float x1 = A[kk * 3], y1 = A[kk * 3 + 1], z1 = A[kk * 3 + 2];
    sx1[t] = make_float4(x1, y1, z1, 0.f);
    const float x1a = live ? x1 : INFINITY;  // (phase A only: a dead lane's q = match * rsq(inf) = 0)
    float ax = 0.f, ay = 0.f, az = 0.f;
    const int lbeg = blockIdx.y * lspan;
    const int lend = min(m, lbeg + lspan);
    const int bl = t & (MG_TL - 1);  // phase-B row (l) of this thread
    const int br = t / MG_TL;        // phase-B k slice: [br*32, br*32+32)
    const unsigned koff = (unsigned)kk * 4u;
    // this thread's phase-B row is l0 + (t & 31): the column record its lane holds (before the rows: the oldest load in
    // flight, as inside the loop -- the wait at the loop head then lets the rows stay in flight)
    float cxv, cyv, czv;
    {
        const int ll = min(lbeg + (t & (MG_TL - 1)), m - 1);
        cxv = B[ll * 3];
        cyv = B[ll * 3 + 1];
        czv = B[ll * 3 + 2];
    }
    __builtin_amdgcn_sched_barrier(0);
    float mv[MG_TL];  // raw rows; unconditional loads from clamped rows (finite values)
#pragma unroll
    for (int l = 0; l < MG_TL; l++) mv[l] = RFA_MG_LOAD((const float *)((const char *)(M + (size_t)min(lbeg + l, lend - 1) * n) + koff));
    float scb[2][24];  // column records of 8 rows (whole tiles), ping-pong
    for (int l0 = lbeg; l0 < lend; l0 += MG_TL) {
        const int lc = min(MG_TL, lend - l0);
        // the next tile's column records first: they are the oldest loads in flight when the next phase A needs them
        float cxn, cyn, czn;
        {
            const int ll = min(l0 + MG_TL + (t & (MG_TL - 1)), m - 1);
            cxn = B[ll * 3];
            cyn = B[ll * 3 + 1];
            czn = B[ll * 3 + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
#define RFA_MCG_PHASE_A(WHOLE)                                                                                           \
    _Pragma("unroll") for (int g = 0; g < MG_TL; g += 8) {                                                               \
        if (WHOLE) {                                                                                                     \
            /* uniform addresses: scalar loads (the tile is whole: in bounds), one group ahead; scalar loads return out  \
               of order, so the wait for this group's records is lgkmcnt(0) and comes BEFORE the next group's issue */   \
            __builtin_amdgcn_s_waitcnt(0xC07F);                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            if (g + 8 < MG_TL) {                                                                                         \
                const float *bp = B + (size_t)(l0 + g + 8) * 3;                                                          \
                _Pragma("unroll") for (int i = 0; i < 24; i++) scb[((g >> 3) + 1) & 1][i] = bp[i];                       \
            }                                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }                                                                                                                \
        const float(&sc)[24] = scb[(g >> 3) & 1];                                                                        \
        _Pragma("unroll") for (int l = g; l < g + 8; l++) {                                                              \
            const float dx = x1a - (WHOLE ? sc[(l - g) * 3 + 0] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cxv), l))), \
                        dy = y1 - (WHOLE ? sc[(l - g) * 3 + 1] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cyv), l))),  \
                        dz = z1 - (WHOLE ? sc[(l - g) * 3 + 2] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(czv), l)));  \
            const float mvl = WHOLE ? mv[l] : mv[l] * ((l0 + l < lend) ? 1.f : 0.f);                                     \
            const float q = mvl * __builtin_amdgcn_rsqf(fmaxf(rf::d2_fma(dx, dy, dz), 1e-20f));                          \
            ax = fmaf(dx, q, ax);                                                                                        \
            ay = fmaf(dy, q, ay);                                                                                        \
            az = fmaf(dz, q, az);                                                                                        \
            qs[l][t] = q;                                                                                                \
        }                                                                                                                \
        _Pragma("unroll") for (int l = g; l < g + 8; l++)                                                                \
            mv[l] = RFA_MG_LOAD((const float *)((const char *)(M + (size_t)min(l0 + MG_TL + l, lend - 1) * n) + koff)); \
        __builtin_amdgcn_sched_barrier(0); /* 8 rows at a time: the scheduler otherwise hoists all 32 rows' work */      \
    }
        if (lc == MG_TL) {  // (uniform)
            const float *bp = B + (size_t)l0 * 3;
#pragma unroll
            for (int i = 0; i < 24; i++) scb[0][i] = bp[i];
            RFA_MCG_PHASE_A(true)
        } else {
            RFA_MCG_PHASE_A(false)
        }
#undef RFA_MCG_PHASE_A
        __syncthreads();
        {
            const float x2 = cxv, y2 = cyv, z2 = czv;
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll 8
            for (int j = 0; j < MG_TL; j++) {
                const int kq = br * MG_TL + j;
                const float q = qs[bl][kq];
                const float4 p = sx1[kq];
                sx = fmaf(x2 - p.x, q, sx);
                sy = fmaf(y2 - p.y, q, sy);
                sz = fmaf(z2 - p.z, q, sz);
            }
            ps[br][bl][0] = sx;
            ps[br][bl][1] = sy;
            ps[br][bl][2] = sz;
        }
        __syncthreads();
        if (t < MG_TL * 3) {
            const int l = t / 3, c = t - l * 3;
            if (l < lc) {
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < TPB / MG_TL; r++) v += ps[r][l][c];
                atomicAdd(&grad2[((size_t)bi * m + l0 + l) * 3 + c], v);
            }
        }
        cxv = cxn;
        cyv = cyn;
        czv = czn;
    }
    if (live) {
        float *g = grad1 + ((size_t)bi * n + k) * 3;
        atomicAdd(g + 0, ax);
        atomicAdd(g + 1, ay);
        atomicAdd(g + 2, az);
    }
}

// ---- match_cost_grad, second form (round 5): whole rows per workgroup, no LDS tile, no barrier.
// A lane owns FOUR consecutive k (one 16-byte load per match row: a wave reads 1 KB, a workgroup 4 KB of a row in one piece) and
// walks an l-range; grad1 of its four k accumulates in registers over the range, as in mcg_kernel.  grad2[l] = -sum_k (x1_k - x2_l) q
// is a sum ACROSS lanes: every lane adds its four products per row into three per-row registers, and eight rows' 24 partial sums
// are reduced over the 64 lanes by a reduce-scatter -- v_permlane32_swap + add halves the values held per lane (24 -> 12), then
// v_permlane16_swap + add (12 -> 6), then four DPP row rotations per value leave every lane of a 16-lane row with the row's total of
// its six values; lane j < 6 of each row picks value j: 24 lanes hold the 24 sums and add them to grad2 with one atomic each.
// 65 VALU per 8 rows x 4 k instead of a 6-VALU-per-entry column phase behind an LDS transpose: 18 VALU per match entry (13 for
// q and grad1, 3 for the row sums, 2 for the reduction) against mcg_kernel's 25.  The rows a lane will need are in flight 8 deep
// (rolling: a row's register is reloaded when its group is done); the rows' x2 are wave-uniform scalar loads.
// Needs n % 4 == 0 and 16-byte aligned xyz1 / match (else mcg_kernel).
constexpr int MR_KPL = 4;     // k per lane
constexpr int MR_LSPAN_MAX = 1024;  // rows per workgroup at most (its grad2 sums live in LDS)
constexpr int MR_G = 4;      // rows per reduce-scatter group (8: +-2 %)
constexpr int MR_DEPTH = 8;  // rows in flight per lane, 2 or 4 groups (16: +-2 %, and 128 registers)
// one group: MR_G rows x 4 k.  xs: the rows' x2 records (wave-uniform).  TAIL: rows at or beyond `lend` are masked by a
// multiplication (their registers hold the range's last row).
// `match` is read once, front to back: non-temporal loads (C4 same-device: mcg_rows 104.5 -> 94.3 us, 5.1 -> 5.7 TB/s; inside
// the sequence approx_match -> match_cost -> match_cost_grad 97 -> 90 us)
typedef float mr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 mr_load_row(const void *p) {
    const mr_v4f v = __builtin_nontemporal_load((const mr_v4f *)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool FULL, bool TAIL, int G0>
__device__ __forceinline__ void mr_group(const float4 (&mv)[MR_DEPTH], const float (&xs)[3 * MR_G], int lg, int lend, bool live,
                                         float (&x1)[MR_KPL], float (&y1)[MR_KPL], float (&z1)[MR_KPL],
                                         float (&ax)[MR_KPL], float (&ay)[MR_KPL], float (&az)[MR_KPL], float (&S)[3 * MR_G]) {
#pragma unroll
    for (int r = 0; r < MR_G; r++) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        const float x2 = xs[r * 3], y2 = xs[r * 3 + 1], z2 = xs[r * 3 + 2];
        const float4 v4 = mv[G0 + r];
        const float vv[MR_KPL] = {v4.x, v4.y, v4.z, v4.w};
        const float keep = (!TAIL || lg + r < lend) ? 1.f : 0.f;  // (uniform)
#pragma unroll
        for (int e = 0; e < MR_KPL; e++) {
            const float dx = x1[e] - x2, dy = y1[e] - y2, dz = z1[e] - z2;
            float w = FULL ? vv[e] : (live ? vv[e] : 0.f);
            if (TAIL) w *= keep;
            const float q = w * __builtin_amdgcn_rsqf(fmaxf(rf::d2_fma(dx, dy, dz), 1e-20f));
            ax[e] = fmaf(dx, q, ax[e]);
            ay[e] = fmaf(dy, q, ay[e]);
            az[e] = fmaf(dz, q, az[e]);
            sx = fmaf(dx, q, sx);
            sy = fmaf(dy, q, sy);
            sz = fmaf(dz, q, sz);
        }
        // a row at a time: left alone the compiler opens all 16 entries of a group at once and spills (sched_barrier does not
        // bind the order instructions are selected in); the next row's differences are made to depend on this empty statement
#pragma unroll
        for (int e = 0; e < MR_KPL; e++)
            asm volatile("" : "+v"(x1[e]), "+v"(y1[e]), "+v"(z1[e]), "+v"(ax[e]), "+v"(ay[e]), "+v"(az[e]), "+v"(sx), "+v"(sy), "+v"(sz));
        S[r * 3 + 0] = sx;
        S[r * 3 + 1] = sy;
        S[r * 3 + 2] = sz;
    }
}

// the group's 3 * MR_G per-lane sums reduced over the wave's 64 lanes and added to the workgroup's grad2 sums (the rows from lg on, below lend)
typedef double mr_sum_t;
__device__ __forceinline__ void mr_reduce_emit(const float (&S)[3 * MR_G], int lane, int lg, int lend, int lbeg, mr_sum_t *g2s) {
    // reduce-scatter: 3G -> 3G/2 (lanes >= 32 keep the upper half) -> 3G/4 (odd 16-lane rows keep the upper half)
    constexpr int NH = 3 * MR_G / 2, NQ = 3 * MR_G / 4;
    float W[NH], U[NQ];
#pragma unroll
    for (int i = 0; i < NH; i++) {
        const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(S[i]), __float_as_uint(S[i + NH]), false, false);
        W[i] = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(W[i]), __float_as_uint(W[i + NQ]), false, false);
        U[i] = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
#pragma unroll
    for (int i = 0; i < NQ; i++) {  // every lane of a 16-lane row ends with the row's total (row_ror 8, 4, 2, 1)
        U[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(U[i]), 0x128, 0xf, 0xf, false));
        U[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(U[i]), 0x124, 0xf, 0xf, false));
        U[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(U[i]), 0x122, 0xf, 0xf, false));
        U[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(U[i]), 0x121, 0xf, 0xf, false));
    }
    const int j = lane & 15;
    float out = U[0];
#pragma unroll
    for (int i = 1; i < NQ; i++) out = j == i ? U[i] : out;
    // lane -> index into the group's 3G sums (row * 3 + component)
    const int vi = j + ((lane & 16) ? NQ : 0) + ((lane & 32) ? NH : 0);
    // (into the workgroup's own sums in LDS: its four waves hold different k of the same rows; one global atomic per sum at the end)
    if (j < NQ && lg + vi / 3 < lend) __hip_atomic_fetch_add(&g2s[(lg - lbeg) * 3 + vi], (mr_sum_t)-out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool FULL>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void mcg_rows_kernel(int n, int m, int lspan,
                                                  const float *__restrict__ xyz1,
                                                  const float *__restrict__ xyz2,
                                                  const float *__restrict__ match,
                                                  float *__restrict__ grad1,
                                                  float *__restrict__ grad2) {
    const int bi = blockIdx.z;
    const int t = threadIdx.x;
    const int k = (blockIdx.x * TPB + t) * MR_KPL;
    const bool live = FULL || k < n;  // (n % 4 == 0: a lane's four k are all inside or all outside)
    const int kk = live ? k : 0;
    float x1[MR_KPL], y1[MR_KPL], z1[MR_KPL];
    {
        const float4 *A = (const float4 *)(xyz1 + ((size_t)bi * n + kk) * 3);
        const float4 a0 = A[0], a1 = A[1], a2 = A[2];
        x1[0] = a0.x, y1[0] = a0.y, z1[0] = a0.z;
        x1[1] = a0.w, y1[1] = a1.x, z1[1] = a1.y;
        x1[2] = a1.z, y1[2] = a1.w, z1[2] = a2.x;
        x1[3] = a2.y, y1[3] = a2.z, z1[3] = a2.w;
    }
    const cfloat *B = (const cfloat *)(xyz2 + (size_t)bi * m * 3);
    const float *__restrict__ M = match + (size_t)bi * n * m;  // (uniform: a row's base is scalar, the lane adds koff bytes)
    const unsigned koff = (unsigned)kk * 4u;
    float *__restrict__ g2 = grad2 + (size_t)bi * m * 3;
    const int lbeg = blockIdx.y * lspan;
    const int lend = min(m, lbeg + lspan);
    float ax[MR_KPL], ay[MR_KPL], az[MR_KPL];
#pragma unroll
    for (int e = 0; e < MR_KPL; e++) ax[e] = ay[e] = az[e] = 0.f;
    __shared__ mr_sum_t g2s[MR_LSPAN_MAX * 3];  // grad2 of the workgroup's rows, summed over its four waves
    for (int i = t; i < (lend - lbeg) * 3; i += TPB) g2s[i] = (mr_sum_t)0;
    __syncthreads();
    // the x2 records of a group of rows: 3 * MR_G consecutive dwords by scalar loads (the base row clamped into the cloud: a
    // prefetch behind the range loads records nobody reads)
#define RFA_MR_LDX(d, row)                                                                 \
    {                                                                                      \
        const cfloat *bp_ = B + (size_t)max(0, min((row), m - MR_G)) * 3;                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 3 * MR_G; i_++) d[i_] = bp_[i_];           \
    }
    float xs0[3 * MR_G], xs1[3 * MR_G];
    RFA_MR_LDX(xs0, lbeg)
    float4 mv[MR_DEPTH];
#pragma unroll
    for (int i = 0; i < MR_DEPTH; i++) mv[i] = mr_load_row((const char *)(M + (size_t)min(lbeg + i, lend - 1) * n) + koff);
    const int lane = t & 63;
    int l0 = lbeg;
    // one group of the whole-block path.  Scalar loads return out of order: the wait for this group's records is "all", and
    // comes before the next group's issue.  The group's rows consumed, their registers take the rows MR_DEPTH further on
    // (behind the range: its last row).
#define RFA_MR_BODY(G0, CUR, NXT)                                                                                          \
    {                                                                                                                      \
        const int lg = l0 + G0;                                                                                            \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        RFA_MR_LDX(NXT, lg + MR_G)                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        float S[3 * MR_G];                                                                                                 \
        mr_group<FULL, false, G0>(mv, CUR, lg, lend, live, x1, y1, z1, ax, ay, az, S);                                     \
        _Pragma("unroll") for (int r = 0; r < MR_G; r++)                                                                   \
            mv[G0 + r] = mr_load_row((const char *)(M + (size_t)min(lg + MR_DEPTH + r, lend - 1) * n) + koff);              \
        mr_reduce_emit(S, lane, lg, lend, lbeg, g2s);                                                                      \
    }
#define RFA_MR_TAIL(G0)                                                                                                    \
    if (l0 + G0 < lend) {                                                                                                  \
        const int lg = l0 + G0;                                                                                            \
        float xt[3 * MR_G];                                                                                                \
        _Pragma("unroll") for (int r = 0; r < MR_G; r++) {                                                                 \
            const cfloat *bp = B + (size_t)min(lg + r, m - 1) * 3;                                                         \
            xt[r * 3] = bp[0], xt[r * 3 + 1] = bp[1], xt[r * 3 + 2] = bp[2];                                               \
        }                                                                                                                  \
        float4 tv[MR_DEPTH]; /* (its own loads: the prefetch registers end their life with the loop above) */             \
        _Pragma("unroll") for (int r = 0; r < MR_G; r++)                                                                   \
            tv[r] = mr_load_row((const char *)(M + (size_t)min(lg + r, lend - 1) * n) + koff);                             \
        float S[3 * MR_G];                                                                                                 \
        mr_group<FULL, true, 0>(tv, xt, lg, lend, live, x1, y1, z1, ax, ay, az, S);                                        \
        mr_reduce_emit(S, lane, lg, lend, lbeg, g2s);                                                                      \
    }
    static_assert(MR_DEPTH == 4 * MR_G || MR_DEPTH == 2 * MR_G, "two or four groups per block below");
    for (; l0 + MR_DEPTH <= lend; l0 += MR_DEPTH) {
        RFA_MR_BODY(0, xs0, xs1)
        RFA_MR_BODY(MR_G, xs1, xs0)
        if (MR_DEPTH == 4 * MR_G) {
            RFA_MR_BODY((2 * MR_G) % MR_DEPTH, xs0, xs1)
            RFA_MR_BODY((3 * MR_G) % MR_DEPTH, xs1, xs0)
        }
    }
    if (l0 < lend) {  // (uniform) the range's last, partial block: rows masked, records fetched row by row
        RFA_MR_TAIL(0)
        RFA_MR_TAIL(MR_G)
        if (MR_DEPTH == 4 * MR_G) {
            RFA_MR_TAIL(2 * MR_G)
            RFA_MR_TAIL(3 * MR_G)
        }
    }
#undef RFA_MR_LDX
#undef RFA_MR_BODY
#undef RFA_MR_TAIL
    // grad1 of the workgroup's 1024 k: through LDS into memory order, so that a wave's atomic covers 64 consecutive floats
    // (added straight from the lanes -- twelve atomics of stride 48 bytes each -- these were HALF of the kernel's time: 16 l-ranges
    // add into the same 3072 addresses and every instruction touched 48 cache lines)
    __shared__ float gs[TPB * MR_KPL * 3];
#pragma unroll
    for (int e = 0; e < MR_KPL; e++) {
        gs[(t * MR_KPL + e) * 3 + 0] = ax[e];
        gs[(t * MR_KPL + e) * 3 + 1] = ay[e];
        gs[(t * MR_KPL + e) * 3 + 2] = az[e];
    }
    __syncthreads();
    for (int i = t; i < (lend - lbeg) * 3; i += TPB) atomicAdd(&g2[(size_t)lbeg * 3 + i], (float)g2s[i]);
    {
        const int k0 = blockIdx.x * TPB * MR_KPL;
        const int cnt = min(TPB * MR_KPL, n - k0) * 3;  // (>= 0: the grid covers n)
        float *gp = grad1 + ((size_t)bi * n + k0) * 3;
#pragma unroll
        for (int i = 0; i < MR_KPL * 3; i++) {
            const int o = i * TPB + t;
            if (o < cnt) atomicAdd(gp + o, gs[o]);
        }
    }
}

// ---- earth_mover fused (row f1): cost and its gradients straight from the per-level ratio vectors;
// `match` (4*B*n*m bytes: 512 MiB at C4, 1 GiB per sample at 16384^2) is never materialised.
// The match entry is recomputed in registers with the same level-ordered fma chain as
// am_match_kernel, then used at once the way matchcost / matchcostgrad1/2 use it
// (tf_approxmatch.cu:183-295): cost += sqrt(d2)*match, q = match*rsq(max(d2,1e-20)),
// grad1[k] += (x1-x2)q, grad2[l] += (x2-x1)q.  Compute-bound (10 exp per pair) instead of three
// HBM passes over match.  Layout as mcg_kernel: thread <-> k, the l-range walked in tiles of 32;
// the column operands (x2_l and the 10 ratioR values of l) are wave-uniform and come from one
// 64-byte record per l by scalar loads.
constexpr int EF_REC = 16;  // floats per column record: x y z 0 | ratioR[0..NLV) | 0 ..
__global__ void emd_pack_cols_kernel(int m, int mpad, int nlv, const float *__restrict__ xyz2,
                                     const float *__restrict__ ratios, size_t lv_stride,
                                     size_t b_stride, int roff, float *__restrict__ rec, int pairs) {
    const int bi = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= mpad) return;
    float r[EF_REC];
#pragma unroll
    for (int i = 0; i < EF_REC; i++) r[i] = 0.f;
    if (l < m) {
        const float *p = xyz2 + ((size_t)bi * m + l) * 3;
        r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
        for (int v = 0; v < nlv; v++) r[4 + v] = ratios[(size_t)bi * b_stride + (size_t)v * lv_stride + roff + l];
    }
    if (pairs) {  // two columns' records interleaved field by field: (x_l, x_l+1), (y_l, y_l+1), ... -- scalar register PAIRS for packed operations
        float *q = rec + ((size_t)bi * mpad + (l & ~1)) * EF_REC + (l & 1);
#pragma unroll
        for (int i = 0; i < EF_REC; i++) q[2 * i] = r[i];
        return;
    }
    float4 *q = (float4 *)(rec + ((size_t)bi * mpad + l) * EF_REC);
#pragma unroll
    for (int i = 0; i < EF_REC / 4; i++) q[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
}

template <int NLV, bool GRAD, bool LASTZERO, bool SQ>
__global__ __launch_bounds__(TPB) void emd_fused_kernel(int n, int m, int mpad, int lspan,
                                                        const float *__restrict__ xyz1,
                                                        const float *__restrict__ rec,
                                                        const float *__restrict__ ratios,
                                                        size_t lv_stride, size_t b_stride,
                                                        LevelConsts lc, float *__restrict__ partial,
                                                        float *__restrict__ grad1,
                                                        float *__restrict__ grad2) {
    static_assert(NLV + 4 <= EF_REC, "column record too small");
    __shared__ float qs[GRAD ? MG_TL : 1][TPB + 1];
    __shared__ float4 sx1[GRAD ? TPB : 1];
    __shared__ float ps[GRAD ? TPB / MG_TL : 1][MG_TL][3];
    __shared__ float wsum[TPB / 64];
    // (a sample's workgroups on ONE XCD, rf::xcd_contiguous: the sweeps' order)
    const unsigned per_ = gridDim.x * gridDim.y;
    const unsigned lin_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned lgc_ = rf::xcd_contiguous(lin_, per_ * gridDim.z);
    const int bi = lgc_ / per_;
    const int by = (lgc_ - bi * per_) / gridDim.x, bx = (lgc_ - bi * per_) - by * gridDim.x;
    const int t = threadIdx.x;
    const int k = bx * TPB + t;
    const bool live = k < n;
    const int kk = live ? k : n - 1;
    const float *__restrict__ A = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ R = rec + (size_t)bi * mpad * EF_REC;
    const float x1 = A[kk * 3], y1 = A[kk * 3 + 1], z1 = A[kk * 3 + 2];
    float rl[NLV], cl[NLV];
#pragma unroll
    for (int v = 0; v < NLV; v++) {
        rl[v] = live ? ratios[(size_t)bi * b_stride + (size_t)v * lv_stride + kk] : 0.f;
        cl[v] = lc.c[v];
    }
    const float t0 = cl[0] < 0.f ? kSkipArg / -cl[0] : INFINITY;  // (uniform)
    if (GRAD) sx1[t] = make_float4(x1, y1, z1, 0.f);
    float ax = 0.f, ay = 0.f, az = 0.f, csum = 0.f;
    const int lbeg = by * lspan;
    const int lend = min(mpad, lbeg + lspan);  // multiples of MG_TL; records beyond m are zero
    const int bl = t & (MG_TL - 1);
    const int br = t / MG_TL;
    if constexpr (!GRAD) {
        // cost only: two columns per step as the halves of packed fp32 operations, their operands in scalar register pairs (the
        // pair layout of emd_pack_cols_kernel); per entry the same operations in the same order, the cost summed column by column
        for (int l0 = lbeg; l0 < lend; l0 += MG_TL) {
#pragma unroll 2
            for (int l = 0; l < MG_TL; l += 2) {
                const am_v2f *__restrict__ c = (const am_v2f *)(R + (size_t)(l0 + l) * EF_REC);  // uniform: 32 dwords by scalar loads
                const am_v2f dx = c[0] - x1, dy = c[1] - y1, dz = c[2] - z1;
                const am_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
                am_v2f e[NLV];
                am_v2f acc = {0.f, 0.f};
                // (the per-column level guards of am_match_kernel do not pay here: a PAIR of columns is dead at a level only when both
                // are -- 54 % of the pairs still need levels 3-4 at C4 against 32 % of the columns -- and the branches cost the
                // packed chain more than the skipped exponentials return: 0.588 against 0.562 ms per call, same device)
                if (SQ) {
                    level_weights2<NLV, LASTZERO, SQ, 1>(d2, cl, e);
                    if (__ballot(d2.x < t0 || d2.y < t0) != 0ull) {
                        asm volatile("; level 0 kept");
                        const am_v2f a0 = d2 * cl[0];
                        acc = __builtin_elementwise_fma(rl[0] * am_v2f{fast_exp2(a0.x), fast_exp2(a0.y)}, c[4], acc);
                    }
#pragma unroll
                    for (int v = 1; v < NLV; v++) acc = __builtin_elementwise_fma(rl[v] * e[v], c[4 + v], acc);
                } else {
                    level_weights2<NLV, LASTZERO, SQ>(d2, cl, e);
#pragma unroll
                    for (int v = 0; v < NLV; v++) acc = __builtin_elementwise_fma(rl[v] * e[v], c[4 + v], acc);
                }
                csum = fmaf(sqrtf(d2.x), acc.x, csum);
                csum = fmaf(sqrtf(d2.y), acc.y, csum);
            }
        }
    } else
    for (int l0 = lbeg; l0 < lend; l0 += MG_TL) {
#pragma unroll 4
        for (int l = 0; l < MG_TL; l++) {
            const float4 *__restrict__ c = (const float4 *)(R + (size_t)(l0 + l) * EF_REC);  // uniform
            const float4 cx = c[0], r0 = c[1], r1 = c[2], r2 = c[3];
            const float rr[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
            const float dx = cx.x - x1, dy = cx.y - y1, dz = cx.z - z1;  // xyz2 - xyz1, as :207
            const float d2 = rf::d2_fma(dx, dy, dz);
            float e[NLV];
            float acc = 0.f;
            // (am_match_kernel's per-row level guards measured SLOWER here, 0.670 against 0.639 ms per call with gradients: this loop is
            // unrolled four columns deep and the branches break that up)
            if (SQ) {  // the sharpest level under its own wave-uniform test (see am_match_kernel)
                level_weights<NLV, LASTZERO, SQ, 1>(d2, cl, e);
                if (__ballot(d2 < t0) != 0ull) {
                    asm volatile("; level 0 kept");
                    acc = fmaf(rl[0] * fast_exp2(d2 * cl[0]), rr[0], 0.f);
                }
#pragma unroll
                for (int v = 1; v < NLV; v++) acc = fmaf(rl[v] * e[v], rr[v], acc);
            } else {
                level_weights<NLV, LASTZERO, SQ>(d2, cl, e);
#pragma unroll
                for (int v = 0; v < NLV; v++) acc = fmaf(rl[v] * e[v], rr[v], acc);
            }
            csum = fmaf(sqrtf(d2), acc, csum);
            if (GRAD) {
                const float q = acc * __builtin_amdgcn_rsqf(fmaxf(d2, 1e-20f));
                ax = fmaf(-dx, q, ax);  // (x1 - x2) q: negation is exact
                ay = fmaf(-dy, q, ay);
                az = fmaf(-dz, q, az);
                qs[l][t] = q;
            }
        }
        if (GRAD) {
            __syncthreads();
            {
                const float4 cx = *(const float4 *)(R + (size_t)(l0 + bl) * EF_REC);
                float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll 8
                for (int j = 0; j < MG_TL; j++) {
                    const int kq = br * MG_TL + j;
                    const float q = qs[bl][kq];
                    const float4 p = sx1[kq];
                    sx = fmaf(cx.x - p.x, q, sx);
                    sy = fmaf(cx.y - p.y, q, sy);
                    sz = fmaf(cx.z - p.z, q, sz);
                }
                ps[br][bl][0] = sx;
                ps[br][bl][1] = sy;
                ps[br][bl][2] = sz;
            }
            __syncthreads();
            if (t < MG_TL * 3) {
                const int l = t / 3, c = t - l * 3;
                if (l0 + l < m) {
                    float v = 0.f;
#pragma unroll
                    for (int r = 0; r < TPB / MG_TL; r++) v += ps[r][l][c];
                    atomicAdd(&grad2[((size_t)bi * m + l0 + l) * 3 + c], v);
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) csum += __shfl_down(csum, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = csum;
    __syncthreads();
    if (t == 0)
        partial[((size_t)bi * gridDim.y + by) * gridDim.x + bx] =
            (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (GRAD && live) {
        float *g = grad1 + ((size_t)bi * n + k) * 3;
        atomicAdd(g + 0, ax);
        atomicAdd(g + 1, ay);
        atomicAdd(g + 2, az);
    }
}

// ---- the cost alone, columns in the order of their LAST LIVE LEVEL (round 6, late) -------------------------------------------
// A column of set 2 is exactly +0 in ratioR from the level after its last live one on (am_compact_kernel), and the cost does
// not care in which order the columns are summed.  emd_pack_cols_sorted_kernel therefore lays a sample's column records out by
// CLASS -- last live level 0 | 1-2 | 3-4 | 5-6 | 7-9, index order inside a class (a function of the sample alone), every class
// padded to an even count with an all-zero record -- and emd_fused_cls_kernel walks a class with exactly the level pairs it needs:
// no test per column, the packed two-column step intact.  At C4 59 % of the columns end at levels 1-2 and 9 % at level 0: 1.4
// exponentials and 2.8 terms per entry instead of 4 and 9; the terms left out are fma(p, +0, acc) = acc.  (The same guards PER
// COLUMN PAIR inside emd_fused_kernel measured slower: a pair is dead only when both columns are, and the branches break the chain.)
constexpr int EF_NCLS = 5;  // (ten levels: the reference schedule)
__device__ __forceinline__ int ef_class_of(int dl) { return dl == 0 ? 0 : (dl >= 7 ? 4 : (dl + 1) >> 1); }
// Two launches, a workgroup per EP_TPB consecutive columns (one workgroup per sample walking all of them took 150 us at 16384
// points, its 64-byte records scattered over the class segments): emd_class_count_kernel leaves every workgroup's class counts,
// emd_pack_cols_sorted_kernel turns the counts before it into its bases and places its columns -- class base + the workgroups
// before + the waves before (LDS) + the lanes before (ballot): index order inside a class.
constexpr int EP_TPB = 1024;
__device__ __forceinline__ int ef_column_class(int l, int m, int mpad, const float *__restrict__ RR, size_t lv_stride, float (&r)[EF_REC]) {
    // r[4 + v] = ratioR of level v (a column beyond m: zeros, class 0; beyond mpad: no class)
    if (l >= mpad) return -1;
    if (l >= m) return 0;
#pragma unroll
    for (int v = 0; v < 10; v++) r[4 + v] = RR[(size_t)v * lv_stride + l];
    int dl = 0;
#pragma unroll
    for (int v = 1; v < 10; v++) dl = r[4 + v] != 0.f ? v : dl;
    return ef_class_of(dl);
}
__global__ __launch_bounds__(EP_TPB) void emd_class_count_kernel(int m, int mpad, const float *__restrict__ ratios, size_t lv_stride,
                                                                size_t b_stride, int roff, int *__restrict__ wgcnt) {
    __shared__ unsigned wtot[EP_TPB / 64][EF_NCLS];
    const int bi = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float r[EF_REC];
#pragma unroll
    for (int i = 0; i < EF_REC; i++) r[i] = 0.f;
    const int c = ef_column_class(g * EP_TPB + tid, m, mpad, ratios + (size_t)bi * b_stride + roff, lv_stride, r);
#pragma unroll
    for (int k = 0; k < EF_NCLS; k++) {
        const unsigned long long bk = __ballot(c == k);
        if (lane == 0) wtot[wave][k] = (unsigned)__builtin_popcountll(bk);
    }
    __syncthreads();
    if (tid < EF_NCLS) {
        unsigned tot = 0;
        for (int w = 0; w < EP_TPB / 64; w++) tot += wtot[w][tid];
        wgcnt[((size_t)bi * gridDim.x + g) * 8 + tid] = (int)tot;
    }
}
__global__ __launch_bounds__(EP_TPB) void emd_pack_cols_sorted_kernel(int m, int mpad, const float *__restrict__ xyz2,
                                                                     const float *__restrict__ ratios, size_t lv_stride,
                                                                     size_t b_stride, int roff, const int *__restrict__ wgcnt,
                                                                     float *__restrict__ rec, size_t rstride, int *__restrict__ coff) {
    __shared__ unsigned wtot[EP_TPB / 64][EF_NCLS];
    __shared__ unsigned cbase[EF_NCLS + 1], ctot[EF_NCLS], mybase[EF_NCLS];
    const int bi = blockIdx.y, g = blockIdx.x, G = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *__restrict__ R = rec + (size_t)bi * rstride * EF_REC;
    if (tid < EF_NCLS) {  // this class over the sample's workgroups: its total, and what lies before this workgroup
        unsigned tot = 0, bef = 0;
        for (int q = 0; q < G; q++) {
            const unsigned v = (unsigned)wgcnt[((size_t)bi * G + q) * 8 + tid];
            bef += q < g ? v : 0u;
            tot += v;
        }
        ctot[tid] = tot;
        mybase[tid] = bef;
    }
    float r[EF_REC];
#pragma unroll
    for (int i = 0; i < EF_REC; i++) r[i] = 0.f;
    const int l = g * EP_TPB + tid;
    const int c = ef_column_class(l, m, mpad, ratios + (size_t)bi * b_stride + roff, lv_stride, r);
    if (l < m) {
        const float *pt = xyz2 + ((size_t)bi * m + l) * 3;
        r[0] = pt[0], r[1] = pt[1], r[2] = pt[2];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned mine = 0;  // lanes of this wave before this one in the same class
#pragma unroll
    for (int k = 0; k < EF_NCLS; k++) {
        const unsigned long long bk = __ballot(c == k);
        if (lane == 0) wtot[wave][k] = (unsigned)__builtin_popcountll(bk);
        if (c == k) mine = (unsigned)__builtin_popcountll(bk & below);
    }
    __syncthreads();
    if (tid == 0) {
        unsigned off = 0;
        for (int k = 0; k < EF_NCLS; k++) {
            cbase[k] = off;
            off += (ctot[k] + 1u) & ~1u;
        }
        cbase[EF_NCLS] = off;
        if (g == 0)
            for (int k = 0; k <= EF_NCLS; k++) coff[bi * 8 + k] = (int)cbase[k];
    }
    __syncthreads();
    auto put = [&](unsigned p, const float (&rr)[EF_REC]) {  // (the pair layout of emd_pack_cols_kernel)
        float *q = R + (size_t)(p & ~1u) * EF_REC + (p & 1u);
#pragma unroll
        for (int i = 0; i < EF_REC; i++) q[2 * i] = rr[i];
    };
    if (c >= 0) {
        unsigned p = mine;
#pragma unroll
        for (int k = 0; k < EF_NCLS; k++) {
            unsigned bef = 0;
#pragma unroll
            for (int w = 0; w < EP_TPB / 64; w++) bef += w < wave ? wtot[w][k] : 0u;
            if (c == k) p += cbase[k] + mybase[k] + bef;
        }
        put(p, r);
    }
    if (g == 0 && tid < EF_NCLS && (ctot[tid] & 1u)) {  // the all-zero record that makes a class's count even
        float z[EF_REC];
#pragma unroll
        for (int i = 0; i < EF_REC; i++) z[i] = 0.f;
        put(cbase[tid] + ctot[tid], z);
    }
}

// one two-column step of class CLS (the columns' records c: pair layout); the level-ordered fma chain of am_match_kernel over the
// class's levels -- weights of odd levels from the next even one's by two squarings, level 9's is 1.0
// (`f`: the first EF_NEED(CLS) floats of the pair's record, wave-uniform -- scalar registers)
constexpr int ef_need(int cls) { return 8 + 2 * (cls == 0 ? 1 : (cls == 4 ? 10 : 2 * cls + 1)); }  // xyz pairs, pad, the class's ratioR pairs
template <int CLS>
__device__ __forceinline__ void ef_pair_step(const float (&f)[ef_need(CLS)], float x1, float y1, float z1, const float (&rl)[10],
                                             const float (&cl)[10], float t0, float &csum) {
    auto c = [&](int i) { return am_v2f{f[2 * i], f[2 * i + 1]}; };
    const am_v2f dx = c(0) - x1, dy = c(1) - y1, dz = c(2) - z1;
    const am_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
    auto ex2 = [](am_v2f a) { return am_v2f{fast_exp2(a.x), fast_exp2(a.y)}; };
    am_v2f e1 = {0.f, 0.f}, e2 = e1, e3 = e1, e4 = e1, e5 = e1, e6 = e1, e7 = e1, e8 = e1;
    if constexpr (CLS >= 4) { e8 = ex2(d2 * cl[8]); const am_v2f q = e8 * e8; e7 = q * q; }
    if constexpr (CLS >= 3) { e6 = ex2(d2 * cl[6]); const am_v2f q = e6 * e6; e5 = q * q; }
    if constexpr (CLS >= 2) { e4 = ex2(d2 * cl[4]); const am_v2f q = e4 * e4; e3 = q * q; }
    if constexpr (CLS >= 1) { e2 = ex2(d2 * cl[2]); const am_v2f q = e2 * e2; e1 = q * q; }
    am_v2f acc = {0.f, 0.f};
    if (__ballot(d2.x < t0 || d2.y < t0) != 0ull) {  // (uniform) the sharpest level: beyond t0 its weight is exactly +0
        asm volatile("; level 0 kept");
        acc = __builtin_elementwise_fma(rl[0] * ex2(d2 * cl[0]), c(4), acc);
    }
    if constexpr (CLS >= 1) { acc = __builtin_elementwise_fma(rl[1] * e1, c(5), acc); acc = __builtin_elementwise_fma(rl[2] * e2, c(6), acc); }
    if constexpr (CLS >= 2) { acc = __builtin_elementwise_fma(rl[3] * e3, c(7), acc); acc = __builtin_elementwise_fma(rl[4] * e4, c(8), acc); }
    if constexpr (CLS >= 3) { acc = __builtin_elementwise_fma(rl[5] * e5, c(9), acc); acc = __builtin_elementwise_fma(rl[6] * e6, c(10), acc); }
    if constexpr (CLS >= 4) {
        acc = __builtin_elementwise_fma(rl[7] * e7, c(11), acc);
        acc = __builtin_elementwise_fma(rl[8] * e8, c(12), acc);
        acc = __builtin_elementwise_fma(rl[9] * am_v2f{1.0f, 1.0f}, c(13), acc);
    }
    // (v_sqrt_f32, 1 ulp: the correctly rounded sqrtf is 16 instructions, twice per step -- a third of a class-1 step; the cost's
    // bar is rel 1e-5, and match_cost, the reference's op, keeps sqrtf)
    csum = fmaf(__builtin_amdgcn_sqrtf(d2.x), acc.x, csum);
    csum = fmaf(__builtin_amdgcn_sqrtf(d2.y), acc.y, csum);
}

// cost only, the reference schedule (ten levels, the last 0, quarter chain): thread <-> row k, a workgroup's span of the
// class-sorted records [by * lspan, by * lspan + lspan) cut at the class boundaries.  (Two rows per thread -- a step's 128
// bytes of scalar loads serving 128 rows -- measured SLOWER: fused earth_mover at C4 0.525 against 0.513 ms, 4 x 16384^2 911 against
// 864 us for this kernel: 84 registers, five waves per SIMD.)
__global__ __launch_bounds__(TPB) void emd_fused_cls_kernel(int n, int lspan, const float *__restrict__ xyz1,
                                                            const float *__restrict__ rec, size_t rstride,
                                                            const int *__restrict__ coff, const float *__restrict__ ratios,
                                                            size_t lv_stride, size_t b_stride, LevelConsts lc,
                                                            float *__restrict__ partial) {
    __shared__ float wsum[TPB / 64];
    const unsigned per_ = gridDim.x * gridDim.y;
    const unsigned lin_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned lgc_ = rf::xcd_contiguous(lin_, per_ * gridDim.z);
    const int bi = lgc_ / per_;
    const int by = (lgc_ - bi * per_) / gridDim.x, bx = (lgc_ - bi * per_) - by * gridDim.x;
    const int t = threadIdx.x;
    const float *__restrict__ A = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ R = rec + (size_t)bi * rstride * EF_REC;
    const int k = bx * TPB + t;
    const bool live = k < n;
    const int kk = live ? k : n - 1;
    const float x1 = A[kk * 3], y1 = A[kk * 3 + 1], z1 = A[kk * 3 + 2];
    float rl[10], cl[10];
#pragma unroll
    for (int v = 0; v < 10; v++) {
        rl[v] = live ? ratios[(size_t)bi * b_stride + (size_t)v * lv_stride + kk] : 0.f;  // (a row beyond n: every term 0)
        cl[v] = lc.c[v];
    }
    const float t0 = cl[0] < 0.f ? kSkipArg / -cl[0] : INFINITY;  // (uniform)
    const int *__restrict__ co = coff + bi * 8;
    const int lbeg = by * lspan, lend = lbeg + lspan;  // (even; the class boundaries are even too)
    float csum = 0.f;
    // a class's pairs with their records through TWO scalar register sets in turn (the sweeps' scheme): the next pair's loads are
    // on their way while this one is worked on -- asked for at the top of its own step, a record was three waits per step
    const cfloat *__restrict__ Rc = (const cfloat *)R;
#define EF_FETCH(dst, CLS, l_)                                                                            \
    _Pragma("unroll") for (int i = 0; i < ef_need(CLS); i++) dst[i] = Rc[(size_t)(l_) * EF_REC + i]
#define EF_STEPS(src, CLS) ef_pair_step<CLS>(src, x1, y1, z1, rl, cl, t0, csum)
#define EF_CLASS(CLS)                                                                                     \
    {                                                                                                     \
        const int a_ = max(lbeg, co[CLS]), e_ = min(lend, co[CLS + 1]);                                   \
        if (a_ < e_) {                                                                                    \
            float fa[ef_need(CLS)], fb[ef_need(CLS)];                                                     \
            EF_FETCH(fa, CLS, a_);                                                                        \
            for (int l = a_; l < e_; l += 4) {                                                            \
                __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): set a has arrived */                   \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                EF_FETCH(fb, CLS, min(l + 2, e_ - 2));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                EF_STEPS(fa, CLS);                                                                        \
                if (l + 2 >= e_) break;                                                                   \
                __builtin_amdgcn_s_waitcnt(0xC07F);                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                EF_FETCH(fa, CLS, min(l + 4, e_ - 2));                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                        \
                EF_STEPS(fb, CLS);                                                                        \
            }                                                                                             \
        }                                                                                                 \
    }
    EF_CLASS(0)
    EF_CLASS(1)
    EF_CLASS(2)
    EF_CLASS(3)
    EF_CLASS(4)
#undef EF_CLASS
#undef EF_STEPS
#undef EF_FETCH
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) csum += __shfl_down(csum, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = csum;
    __syncthreads();
    if (t == 0) partial[((size_t)bi * gridDim.y + by) * gridDim.x + bx] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ---- the sharp levels of the schedule ---------------------------------------------------------------------------
// At level -4^7 (then -4^6, -4^5) the weight exp2(level*log2e * d2) of a pair is EXACTLY +0 once d2 passes a threshold: v_exp_f32
// returns +0 for every argument <= -160 (tests/test_gpu_emd.py sweeps the instruction), and a pair with weight 0 adds
// fma(0 * rl, s, acc) = acc to every sum of its level.  The skipping sweeps (am_rowk / am_rowl SKIP) use that in the DENSE column
// order: they pay while a wave keeps few of its columns -- cut-offs up to d = 0.22 (levels -4^7 and -4^6: 9 / 19 % kept at C4); at
// -4^5 (42 % kept) the packed dense sweep is faster (P2 33 us against 39).
// (Rounds 3-5 also ran these levels CULLED over the sorted clouds' 16-record blocks for the cost-only rf_earth_mover of clouds of
// >= 4096 points -- sums in sorted order.  Round 6 took that route out: with the live-column sweeps behind it, it only paid at
// 16384^2 any more (2.90 against 3.11 ms; 4096^2 1.09 against 1.02), and a soak against the oracle found its COST 1.0e-5 .. 3.2e-5
// off on 3 of 17 000 random large shapes, where the sweeps in column order stay within 8e-7:
// tools/experiments/emd_cull_route.patch.txt, emd_cull_live_cost_oracle.py.)
#ifndef RFA_SKIP_MAXT
#define RFA_SKIP_MAXT 0.05f
#endif
constexpr float kSkipMaxT = RFA_SKIP_MAXT;

__global__ void probe_exp2_kernel(const float *__restrict__ x, float *__restrict__ y, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) y[i] = fast_exp2(x[i]);
}

int default_levels(float *lv) {
    int c = 0;
    for (int j = 7; j >= -2; j--) lv[c++] = (j == -2) ? 0.0f : -ldexpf(1.0f, 2 * j);
    return c;
}

struct AmLayout {
    int npad, mpad;
    size_t V;        // floats per vector pair [L: npad | R: mpad]
    size_t bstride;  // floats per batch element in the vector region: (1 + nlevels) * V
    size_t off_x1, off_x2, total;  // in floats
    bool rowsort_ok;       // the dense sweeps of the sharp levels take their rows in the clouds' spatial order (am_rowk_kernel SKIP)
    bool compact_ok;       // from the third level on the sweeps run over the LIVE columns / rows of set 2 only (am_compact_kernel)
    size_t cstride;        // floats per sample of a packed column array
    size_t off_live[2], live_floats;  // two packed sets (LiveSet), used in turn
    size_t off_maskk, off_maskl;  // (rowsort_ok) the skipping sweeps' column lists, rows of set 1 / rows of set 2 (am_rowk_kernel MASK)
    int nsa, nsb;          // padded sizes of the two sorted sets
    size_t off_sa, off_sb;
};
#ifndef RFA_ROWSORT_MIN_PAIRS
#define RFA_ROWSORT_MIN_PAIRS 6.0e7
#endif
constexpr double ROWSORT_MIN_PAIRS = RFA_ROWSORT_MIN_PAIRS;

int round_up_i(int v, int q) { return (v + q - 1) / q * q; }

// mode (include/rfops.h): RF_EMD_AUTO -- the routes by the size of the whole batch (below); RF_EMD_SWEPT -- every level as a
// dense sweep over the clouds in the caller's order and every launch shape taken as for b = 1: a sample's bits do not depend on
// the batch it is called in (the reference is batch-independent per sample, tf_approxmatch.cu:13).
AmLayout am_layout(int b, int n, int m, int nlevels, int mode = RF_EMD_AUTO) {
    AmLayout L;
    L.npad = round_up_i(n, CPAD);
    L.mpad = round_up_i(m, CPAD);
    L.V = (size_t)L.npad + L.mpad;
    L.bstride = L.V * (size_t)(1 + nlevels);
    size_t off = (size_t)b * L.bstride + 64;  // + slack: scalar prefetch runs SUB entries ahead
    L.off_x1 = off;
    off += (size_t)b * L.npad * 3 + 64;
    L.off_x2 = off;
    off += (size_t)b * L.mpad * 3 + 64;
    const bool swept = mode == RF_EMD_SWEPT;
    // (sizes alone decide; same device: 32 x 1024^2 = 3.4e7 pairs 0.324 ms with the sort against 0.316 without, 32 x 2048^2 = 1.3e8 pairs 1.026 against 1.057)
    L.rowsort_ok = !swept && n >= 512 && m >= 512 && (double)b * n * m >= ROWSORT_MIN_PAIRS && rfp::pruned_supported(b, n, m);
    L.nsa = L.nsb = 0;
    L.off_sa = L.off_sb = 0;
    if (L.rowsort_ok) {
        L.nsa = rfp::sorted_view(b, n, nullptr).npad;
        L.nsb = rfp::sorted_view(b, m, nullptr).npad;
        off = (off + 63) / 64 * 64;  // 256-byte alignment of the sorted sets
        L.off_sa = off;
        off += (rfp::sorted_bytes(b, n) + 3) / 4 + 64;
        off = (off + 63) / 64 * 64;
        L.off_sb = off;
        off += (rfp::sorted_bytes(b, m) + 3) / 4 + 64;
    }
    // (by the clouds' sizes alone, never by b: what a sample's sweeps sum over must not depend on the batch it is called in)
    L.compact_ok = !swept && n >= 512 && m >= 512;
    L.cstride = (size_t)L.mpad + 64;
    L.off_live[0] = L.off_live[1] = 0;
    L.live_floats = (size_t)b * L.cstride * 6 + (size_t)b * 4 + 64;  // cc (3) + s3 + s1 + rows, counts
    if (L.compact_ok) {
        for (int q = 0; q < 2; q++) {
            off = (off + 63) / 64 * 64;
            L.off_live[q] = off;
            off += L.live_floats;
        }
    }
    L.off_maskk = L.off_maskl = 0;
    if (L.rowsort_ok) {
        // per (wave of 64 or 128 sorted rows, column segment) a count and up to seglen 16-bit column numbers
        const size_t wk = (size_t)b * rf::ceil_div(L.nsa, 64) * ((size_t)L.mpad + 16);
        const size_t wl = (size_t)b * rf::ceil_div(L.nsb, 64) * ((size_t)L.npad + 16);
        off = (off + 63) / 64 * 64;
        L.off_maskk = off;
        off += wk / 2 + 64;
        L.off_maskl = off;
        off += wl / 2 + 64;
    }
    L.total = off;
    return L;
}

// waves per workgroup (= column segments): the smallest power of two that gives >= 4096 waves,
// keeping >= 64 columns per segment
int pick_nseg(int b, int rows, int cols_pad, int rpt) {
    long base = (long)b * rf::ceil_div(rows, 64 * rpt);
    int nseg = 1;
    const long target = RFA_TARGET_WAVES;
    while (nseg < 16 && base * nseg < target && cols_pad / (nseg * 2) >= 64) nseg *= 2;
    return nseg;
}

// multiL / multiR: integer division as in tf_approxmatch.cu:4-10
void am_multipliers(int n, int m, float &multiL, float &multiR) {
    if (n >= m) { multiL = 1.f; multiR = (float)(n / m); }
    else        { multiL = (float)(m / n); multiR = 1.f; }
}

// The level pipeline (P1 / P2 / fused P3+P1 launches) of the large-cloud path: fills the
// workspace's per-level ratio vectors.  Shared by rf_approxmatch_levels (which then materialises
// match) and rf_earth_mover (which does not).
int am_run_levels(int b, int n, int m, const float *xyz1, const float *xyz2, int nlevels,
                  const LevelConsts &lc, float multiL, float multiR, void *workspace, hipStream_t s,
                  int mode = RF_EMD_AUTO) {
    AmLayout L = am_layout(b, n, m, nlevels, mode);
    float *w = (float *)workspace;
    float *remainL = w, *remainR = w + L.npad;          // slot 0 of the vector region
    float *ratios = w + L.V;                            // slot 1+v: [ratioL npad | ratioR mpad]
    float *x1p = w + L.off_x1, *x2p = w + L.off_x2;
    // Live columns (am_compact_kernel): from level vC on, every sweep runs over the columns / rows of set 2 whose scalars are not
    // exactly 0 -- which, at the broad end of the schedule, is a few per cent of them.  (Round 5 took the three broadest levels
    // from a truncated Taylor expansion about the clouds' centre instead of sweeping them: 121 us per call at C4 for what these
    // sweeps now do in 40, exactly -- tools/experiments/emd_fgt_route.patch.txt.)
    // vC: the first level behind the last one that takes the skipping sweeps (the packed sets are handed from level to level),
    // the third at the earliest (the reference schedule: 2; its 50-level stretching: 10)
    int vC = 2;
    for (int v = 2; v < nlevels; v++)
        if (lc.c[v] < 0.f && kSkipArg / -lc.c[v] <= kSkipMaxT) vC = v + 1;
    const bool compact = L.compact_ok && vC + 2 <= nlevels;
    // (padded entries of every vector must read 0 -- they are column scalars of padded columns: am_init writes them)
    {
        AmInit ai;
        ai.npts[0] = n, ai.npts[1] = m, ai.npad[0] = L.npad, ai.npad[1] = L.mpad;
        ai.fill[0] = multiL, ai.fill[1] = multiR;
        ai.xyz[0] = xyz1, ai.xyz[1] = xyz2, ai.xyzp[0] = x1p, ai.xyzp[1] = x2p;
        ai.xyzp_stride[0] = (size_t)L.npad * 3, ai.xyzp_stride[1] = (size_t)L.mpad * 3;
        ai.vec = w, ai.stride = L.bstride, ai.V = L.V, ai.nslots = 1 + nlevels, ai.b = b;
        RF_LAUNCH("am_init", am_init_kernel, dim3(rf::ceil_div(max(L.npad, L.mpad), AI_TPB), b, 2), dim3(AI_TPB), 0, s, ai);
    }

    // the sharp levels that stay on the dense sweeps: rows in the clouds' spatial order, columns with all-zero weights skipped
    // (am_rowk_kernel SKIP; bit-identical sums).  Level v qualifies when its weight is exactly 0 from a d2 of at most kSkipMaxT on.
    auto skip_t = [&](int v) { return (v >= 0 && v < nlevels && lc.c[v] < 0.f) ? kSkipArg / -lc.c[v] : INFINITY; };
    const int *permA = nullptr, *permB = nullptr;
    const float *boxA = nullptr, *boxB = nullptr;  // the sorted sets' superblock boxes (the listing launches' first test)
    if (L.rowsort_ok && skip_t(0) <= kSkipMaxT) {
        const rfp::Sorted so[2] = {rfp::sorted_view(b, n, w + L.off_sa), rfp::sorted_view(b, m, w + L.off_sb)};
        const int nn[2] = {n, m};
        const float *src[2] = {xyz1, xyz2};
        if (int e = rfp::sort_sets(b, 2, nn, src, so, s, nullptr)) return e;
        permA = so[0].orig;
        permB = so[1].orig;
        boxA = so[0].box64;
        boxB = so[1].box64;
    }

    // 2 rows per lane (measured best of 1 / 2 / 4: longer compute per scalar prefetch covers the L2
    // latency of the s_loads without dropping below 4 waves per SIMD)
    constexpr int RPT = 2;
    // (the column segments fix the order of a row's sum: under RF_EMD_SWEPT they are those of a batch of one)
    const int bseg = mode == RF_EMD_SWEPT ? 1 : b;
    const int segk = pick_nseg(bseg, n, L.mpad, RPT), segl = pick_nseg(bseg, m, L.npad, RPT);
    const dim3 gk(rf::ceil_div(n, 64 * RPT), b), gl(rf::ceil_div(m, 64 * RPT), b);
    const dim3 gks(rf::ceil_div(L.nsa, 64 * RPT), b), gls(rf::ceil_div(L.nsb, 64 * RPT), b);  // SKIP: over the sorted positions
    // The leading run of levels on the skipping sweeps (vE of them: 2 on the reference schedule, 10 on its 50-level stretching):
    // level 0's launches list, per wave, the columns within the BROADEST of these levels' cut-offs; the launches of levels
    // 1 .. vE - 1 visit only the listed columns (am_rowk_kernel MASK) -- the geometry does not change between the levels of a call.
    int vE = 0;
    for (int v = 0; v < nlevels && permA && permB; v++) {
        if (!(skip_t(v) <= kSkipMaxT && (v == 0 || (lc.c[v - 1] < 0.f && lc.c[v - 1] <= lc.c[v])))) break;
        vE = v + 1;
    }
    const bool masked = vE >= 2 && nlevels > 2 && n < 65536 && m < 65536 &&  // (16-bit column numbers ...
                        L.mpad / segk <= 65535 && L.npad / segl <= 65535;  // ... and 16-bit per-wave counts: a segment of 65536 columns, all listed, would wrap to 0)
    const float tmaskE = vE >= 1 ? skip_t(vE - 1) : 0.f;
    unsigned short *maskk = (unsigned short *)(w + L.off_maskk), *maskl = (unsigned short *)(w + L.off_maskl);
    auto live_set = [&](int q) {
        float *base = w + L.off_live[q];
        LiveSet ls;
        ls.cc = base;
        ls.s3 = base + (size_t)b * L.cstride * 3;
        ls.s1 = ls.s3 + (size_t)b * L.cstride;
        ls.rows = (int *)(ls.s1 + (size_t)b * L.cstride);
        ls.cnt = ls.rows + (size_t)b * L.cstride;
        return ls;
    };
    const LiveSet setA = compact ? live_set(0) : LiveSet{}, setB = compact ? live_set(1) : LiveSet{};
    // level v >= vC: P2 packs its rows into out(v) = (v - vC) even ? B : A, from in(v) = out(v - 1) (in(vC) = A: am_compact_kernel's
    // live set); the fused sweep in front of it reads am_compact_kernel's column set (B) at vC, out(v - 1) after
    auto pack_live = [&](int v, const float *ratioR_v) -> int {  // after P2 of level v = vC - 1
        float *ratioR_later = ratios + (size_t)(v + 1) * L.V + L.npad;
        RF_LAUNCH("am_compact", am_compact_kernel, dim3(b), dim3(CK_TPB), 0, s, m, (const float *)x2p, (size_t)L.mpad * 3, ratioR_v,
                  (const float *)remainR, ratioR_later, L.V, nlevels - 1 - v, L.bstride, setB, setA, L.cstride);
        return RF_OK;
    };
    for (int v = 0; v < nlevels; v++) {
        float *ratioL = ratios + (size_t)v * L.V, *ratioR = ratioL + L.npad;
        const bool zero = lc.c[v] == 0.0f;  // e = exp2(d2 * 0) = 1 exactly: no exponential needed
        const int *gptr = nullptr;  // (the sweeps' guard word: unused)
        const float tsk = skip_t(v);        // (a fused P3 of level v-1 is sharper or equal wherever this one is skippable)
        const bool skip = permA && tsk <= kSkipMaxT && (v == 0 || (lc.c[v - 1] < 0.f && lc.c[v - 1] <= lc.c[v]));
#define AM_ROWK_ARGS(pR_, pL_, cprev)                                                                 \
    n, L.mpad / segk, xyz1, (const float *)x2p, (size_t)L.mpad * 3, pR_, (const float *)remainR, pL_,  \
        remainL, ratioL, L.bstride, cprev, lc.c[v], permA, L.nsa, tsk, skip_t(v - 1), gptr, 1
        const bool live = compact && v >= vC && !skip;  // this level's sweeps run over the live columns / rows (packed after P2 of level v - 1)
        const LiveSet p2out = ((v - vC) & 1) ? setA : setB, p2in = v == vC ? setA : (((v - vC) & 1) ? setB : setA);
        const LiveSet kin = v == vC ? setB : p2in;
#define AM_ROWK_LIVE(pL_, cprev)                                                                                         \
    n, 0, xyz1, (const float *)kin.cc, L.cstride * 3, (const float *)kin.s3, (const float *)kin.s1, pL_, remainL, ratioL, L.bstride, cprev, \
        lc.c[v], (const int *)nullptr, 0, INFINITY, INFINITY, gptr, 1, (unsigned short *)nullptr, 0.f, (const int *)kin.cnt, L.cstride
        if (live) {
            const float *pL = ratios + (size_t)(v - 1) * L.V;
            if (zero) {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 2, RPT, 0, 0, 1>), gk, dim3(64 * segk), 0, s, AM_ROWK_LIVE(pL, lc.c[v - 1]));
            } else if (lc.c[v - 1] == lc.c[v]) {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 3, RPT, 0, 0, 1>), gk, dim3(64 * segk), 0, s, AM_ROWK_LIVE(pL, lc.c[v - 1]));
            } else {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 1, RPT, 0, 0, 1>), gk, dim3(64 * segk), 0, s, AM_ROWK_LIVE(pL, lc.c[v - 1]));
            }
        } else
#undef AM_ROWK_LIVE
        if (skip && v == 0 && masked) {  // ... and lists, per wave, the columns the next level's sweep will have to visit
            RF_LAUNCH("am_p1", (am_rowk_kernel<false, 1, RPT, 1, 1>), gks, dim3(64 * segk), 0, s,
                      AM_ROWK_ARGS((const float *)remainR, (const float *)remainL, 0.f), maskk, tmaskE, (const int *)nullptr, (size_t)0, boxA);
        } else if (skip && v == 0) {
            RF_LAUNCH("am_p1", (am_rowk_kernel<false, 1, RPT, 1>), gks, dim3(64 * segk), 0, s,
                      AM_ROWK_ARGS((const float *)remainR, (const float *)remainL, 0.f));
        } else if (skip && v >= 1 && v < vE && masked) {  // only the columns level 0's sweep listed (a repeated multiplier: the two
                                                           // exponentials have the same argument -- the same bits as the shared one)
            const float *pL = ratios + (size_t)(v - 1) * L.V, *pR = pL + L.npad;
            RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 1, RPT, 1, 2>), gks, dim3(64 * segk), 0, s,
                      AM_ROWK_ARGS(pR, pL, lc.c[v - 1]), maskk, 0.f);
        } else if (skip && lc.c[v - 1] != lc.c[v]) {
            const float *pL = ratios + (size_t)(v - 1) * L.V, *pR = pL + L.npad;
            RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 1, RPT, 1>), gks, dim3(64 * segk), 0, s,
                      AM_ROWK_ARGS(pR, pL, lc.c[v - 1]));
        } else if (v == 0) {
            if (zero) {
                RF_LAUNCH("am_p1", (am_rowk_kernel<false, 2, RPT>), gk, dim3(64 * segk), 0, s,
                          AM_ROWK_ARGS((const float *)remainR, (const float *)remainL, 0.f));
            } else {
                RF_LAUNCH("am_p1", (am_rowk_kernel<false, 1, RPT>), gk, dim3(64 * segk), 0, s,
                          AM_ROWK_ARGS((const float *)remainR, (const float *)remainL, 0.f));
            }
        } else {
            const float *pL = ratios + (size_t)(v - 1) * L.V, *pR = pL + L.npad;
            if (zero) {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 2, RPT>), gk, dim3(64 * segk), 0, s,
                          AM_ROWK_ARGS(pR, pL, lc.c[v - 1]));
            } else if (lc.c[v - 1] == lc.c[v]) {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 3, RPT>), gk, dim3(64 * segk), 0, s,
                          AM_ROWK_ARGS(pR, pL, lc.c[v - 1]));
            } else {
                RF_LAUNCH("am_p3p1", (am_rowk_kernel<true, 1, RPT>), gk, dim3(64 * segk), 0, s,
                          AM_ROWK_ARGS(pR, pL, lc.c[v - 1]));
            }
        }
#undef AM_ROWK_ARGS
        if (live) {
            // (the first two live levels still have rows enough for the long items: 122 / 89 us against 141 / 102 with the short ones)
            const int prows = (n > PL_BIG_N && v >= vC + 2) ? PL_ROWS_BIG : PL_ROWS;
            const dim3 gp(min(rf::ceil_div(L.mpad, prows) * b, 2048));  // (eight workgroups of four waves per CU: all resident)
#define AM_P2_LIVE(Z, ROWS_)                                                                                                      \
    RF_LAUNCH("am_p2", (am_p2_live_kernel<Z, ROWS_>), gp, dim3(PL_TPB), 0, s, L.npad, (const float *)x1p, (size_t)L.npad * 3,         \
              (const float *)ratioL, remainR, ratioR, L.V, nlevels - 1 - v, L.bstride, lc.c[v], p2in, p2out, L.cstride, L.mpad, b)
            if (zero && prows == PL_ROWS) {
                AM_P2_LIVE(true, PL_ROWS);
            } else if (zero) {
                AM_P2_LIVE(true, PL_ROWS_BIG);
            } else if (prows == PL_ROWS) {
                AM_P2_LIVE(false, PL_ROWS);
            } else {
                AM_P2_LIVE(false, PL_ROWS_BIG);
            }
#undef AM_P2_LIVE
        } else
        if (zero) {
            RF_LAUNCH("am_p2", (am_rowl_kernel<RPT, true>), gl, dim3(64 * segl), 0, s, m, L.npad / segl, xyz2,
                      (const float *)x1p, (size_t)L.npad * 3, (const float *)ratioL, remainR, ratioR,
                      L.bstride, lc.c[v], permB, L.nsb, tsk, gptr, 1);
        } else if (permB && tsk <= kSkipMaxT && masked && v < vE) {
            if (v == 0) {
                RF_LAUNCH("am_p2", (am_rowl_kernel<RPT, false, true, 1>), gls, dim3(64 * segl), 0, s, m, L.npad / segl, xyz2,
                          (const float *)x1p, (size_t)L.npad * 3, (const float *)ratioL, remainR, ratioR,
                          L.bstride, lc.c[v], permB, L.nsb, tsk, gptr, 1, maskl, tmaskE, (const int *)nullptr, boxB);
            } else {
                RF_LAUNCH("am_p2", (am_rowl_kernel<RPT, false, true, 2>), gls, dim3(64 * segl), 0, s, m, L.npad / segl, xyz2,
                          (const float *)x1p, (size_t)L.npad * 3, (const float *)ratioL, remainR, ratioR,
                          L.bstride, lc.c[v], permB, L.nsb, tsk, gptr, 1, maskl, 0.f);
            }
        } else if (permB && tsk <= kSkipMaxT) {
            RF_LAUNCH("am_p2", (am_rowl_kernel<RPT, false, true>), gls, dim3(64 * segl), 0, s, m, L.npad / segl, xyz2,
                      (const float *)x1p, (size_t)L.npad * 3, (const float *)ratioL, remainR, ratioR,
                      L.bstride, lc.c[v], permB, L.nsb, tsk, gptr, 1);
        } else {
            RF_LAUNCH("am_p2", (am_rowl_kernel<RPT, false>), gl, dim3(64 * segl), 0, s, m, L.npad / segl, xyz2,
                      (const float *)x1p, (size_t)L.npad * 3, (const float *)ratioL, remainR, ratioR,
                      L.bstride, lc.c[v], permB, L.nsb, tsk, gptr, 1);
        }
        if (compact && v + 1 == vC)  // the first two packed sets (from here on every P2 packs the next one itself)
            if (int e = pack_live(v, ratioR)) return e;
    }
    return RF_OK;
}

}  // namespace

extern "C" {

int rf_probe_exp2(const float *x, float *y, int count, rf_stream_t stream) {
    if (count < 0 || (count > 0 && (!x || !y))) return RF_EINVAL;
    if (count == 0) return RF_OK;
    RF_LAUNCH("probe_exp2", probe_exp2_kernel, dim3(rf::ceil_div(count, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
              count);
    return RF_OK;
}

static bool emd_mode_ok(int mode) { return mode == RF_EMD_AUTO || mode == RF_EMD_SWEPT; }

size_t rf_approxmatch_workspace_bytes(int b, int n, int m, int nlevels) {
    return rf_approxmatch_mode_workspace_bytes(b, n, m, nlevels, RF_EMD_AUTO);
}

size_t rf_approxmatch_mode_workspace_bytes(int b, int n, int m, int nlevels, int mode) {
    if (b <= 0 || n <= 0 || m <= 0 || !emd_mode_ok(mode)) return 0;
    if (nlevels <= 0) nlevels = 10;
    return am_layout(b, n, m, nlevels, mode).total * sizeof(float);
}

int rf_approxmatch_levels(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                          const float *levels_host, int nlevels, void *workspace,
                          size_t workspace_bytes, rf_stream_t stream) {
    return rf_approxmatch_mode(b, n, m, xyz1, xyz2, match, levels_host, nlevels, workspace, workspace_bytes, stream, RF_EMD_AUTO);
}

int rf_approxmatch_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                        const float *levels_host, int nlevels, void *workspace,
                        size_t workspace_bytes, rf_stream_t stream, int mode) {
    float lv_default[16];
    if (!levels_host && nlevels == 0) {  // the reference schedule
        nlevels = default_levels(lv_default);
        levels_host = lv_default;
    }
    if (b < 0 || b > MAX_BATCH || n < 0 || m < 0 || nlevels <= 0 || nlevels > MAX_LEVELS || !levels_host || !emd_mode_ok(mode))
        return RF_EINVAL;
    if (b == 0 || n == 0 || m == 0) return RF_OK;
    if (!xyz1 || !xyz2 || !match || !workspace || !rf::aligned16(workspace)) return RF_EINVAL;
    if (workspace_bytes < rf_approxmatch_mode_workspace_bytes(b, n, m, nlevels, mode)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float multiL, multiR;
    am_multipliers(n, m, multiL, multiR);

    if (n <= AM_SMALL && m <= AM_SMALL) {
        LevelConsts lcs;
        for (int v = 0; v < MAX_LEVELS; v++) lcs.c[v] = v < nlevels ? levels_host[v] * kLog2e : 0.f;
        RF_LAUNCH("am_small", am_small_kernel, dim3(b), dim3(AM_SMALL), 0, s, n, m, nlevels, lcs, multiL,
                  multiR, xyz1, xyz2, match);
        return RF_OK;
    }
    LevelConsts lc;
    for (int v = 0; v < MAX_LEVELS; v++) lc.c[v] = v < nlevels ? levels_host[v] * kLog2e : 0.f;
    {
        const int st = am_run_levels(b, n, m, xyz1, xyz2, nlevels, lc, multiL, multiR, workspace, s, mode);
        if (st != RF_OK) return st;
    }
    const AmLayout L = am_layout(b, n, m, nlevels, mode);
    const float *ratios = (const float *)workspace + L.V;
    // P3 of the last level only updates remainL, which nothing reads afterwards: not launched.
    const dim3 gm(rf::ceil_div(n, AMM_TPB), rf::ceil_div(m, LSEG), b);
    if (nlevels == 10 && lc.c[9] == 0.0f && quarter_chain(lc.c, 10, true)) {  // the reference schedule
        RF_LAUNCH("am_match", (am_match_kernel<10, true, true>), gm, dim3(AMM_TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)ratios, L.V, L.bstride, L.npad, 0, 10, lc, match);
    } else if (nlevels == 10 && lc.c[9] == 0.0f) {
        RF_LAUNCH("am_match", (am_match_kernel<10, true>), gm, dim3(AMM_TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)ratios, L.V, L.bstride, L.npad, 0, 10, lc, match);
    } else if (nlevels == 10) {
        RF_LAUNCH("am_match", (am_match_kernel<10, false>), gm, dim3(AMM_TPB), 0, s, n, m, xyz1, xyz2,
                  (const float *)ratios, L.V, L.bstride, L.npad, 0, 10, lc, match);
    } else {  // any other schedule: every level in one pass (am_match_any_kernel)
#define AM_MATCH_ANY(NG)                                                                                              \
    RF_LAUNCH("am_match", (am_match_any_kernel<NG>), gm, dim3(AMM_TPB), 0, s, n, m, xyz1, xyz2, (const float *)ratios, \
              L.V, L.bstride, L.npad, nlevels, lc, fresh, match)
        unsigned long long fresh = 1ull;  // bit v: level v's multiplier differs from level v - 1's (a new exponential)
        for (int v = 1; v < nlevels; v++) fresh |= (lc.c[v] != lc.c[v - 1]) ? 1ull << v : 0ull;
        static_assert(MAX_LEVELS == 4 * LVG, "the four instantiations below cover every admissible schedule");
        int rep = 0;  // every multiplier exactly `rep` times in a row?
        for (int r = 2; r <= 8 && !rep; r++) {
            bool ok = nlevels % r == 0;
            for (int v = 0; ok && v < nlevels; v++) ok = ((fresh >> v & 1ull) != 0ull) == (v % r == 0);
            if (ok) rep = r;
        }
        if (rep == 5 && nlevels <= 60) {  // (BASELINE configs[3])
            RF_LAUNCH("am_match", (am_match_any_kernel<4, 5>), gm, dim3(AMM_TPB), 0, s, n, m, xyz1, xyz2, (const float *)ratios,
                      L.V, L.bstride, L.npad, nlevels, lc, fresh, match);
        } else if (nlevels <= LVG) {
            AM_MATCH_ANY(1);
        } else if (nlevels <= 2 * LVG) {
            AM_MATCH_ANY(2);
        } else if (nlevels <= 3 * LVG) {
            AM_MATCH_ANY(3);
        } else {
            AM_MATCH_ANY(4);
        }
#undef AM_MATCH_ANY
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
    if (b < 0 || b > MAX_BATCH || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0) return RF_OK;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0 || m == 0) {
        RF_ZERO(cost, sizeof(float) * b, s);
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

// The gradient pass over `match` (outputs zero-filled by the caller): whole rows per workgroup where the layout allows it.
constexpr int MCG_ROWS_WG = 1024;  // workgroups the l-ranges are cut for: four per CU (1536 / 2048 at 6 waves per SIMD: +3 .. +6 %)
static int mcg_launch(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match, float *grad1,
                       float *grad2, hipStream_t s) {
    const int kspan = TPB * MR_KPL;
    // (a last k-block that is mostly dead lanes costs more than the tile form's padding: 32 x 1028 x 1000 46-49 vs 44.6 us)
    if (n % MR_KPL == 0 && 4 * (long)n >= 3L * rf::ceil_div(n, kspan) * kspan && m >= 2 * MR_DEPTH &&
        (((uintptr_t)xyz1 | (uintptr_t)match) & 15) == 0) {
        const int kb = rf::ceil_div(n, kspan);
        int lsplit = rf::ceil_div(MCG_ROWS_WG, b * kb);
        lsplit = max(1, min(lsplit, m / MR_DEPTH));
        lsplit = max(lsplit, rf::ceil_div(m, MR_LSPAN_MAX - MR_DEPTH));
        const int lspan = rf::ceil_div(rf::ceil_div(m, lsplit), MR_DEPTH) * MR_DEPTH;
        dim3 g(kb, rf::ceil_div(m, lspan), b);
        if (n % kspan == 0) {
            RF_LAUNCH("mc_grad", (mcg_rows_kernel<true>), g, dim3(TPB), 0, s, n, m, lspan, xyz1, xyz2, match, grad1, grad2);
        } else {
            RF_LAUNCH("mc_grad", (mcg_rows_kernel<false>), g, dim3(TPB), 0, s, n, m, lspan, xyz1, xyz2, match, grad1, grad2);
        }
        return RF_OK;
    }
    int lsplit = MG_LSPLIT;
    while (lsplit > 1 && m / lsplit < MG_TL) lsplit /= 2;
    const int lspan = rf::ceil_div(rf::ceil_div(m, lsplit), MG_TL) * MG_TL;
    dim3 g(rf::ceil_div(n, TPB), rf::ceil_div(m, lspan), b);
    RF_LAUNCH("mc_grad", mcg_kernel, g, dim3(TPB), 0, s, n, m, lspan, xyz1, xyz2, match, grad1, grad2);
    return RF_OK;
}

int rf_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *grad1, float *grad2, rf_stream_t stream) {
    if (b < 0 || b > MAX_BATCH || n < 0 || m < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) RF_ZERO(grad1, sizeof(float) * 3 * (size_t)b * n, s);
    if ((size_t)b * m) RF_ZERO(grad2, sizeof(float) * 3 * (size_t)b * m, s);
    if (b == 0 || n == 0 || m == 0) return RF_OK;
    return mcg_launch(b, n, m, xyz1, xyz2, match, grad1, grad2, s);
}

// ---- row f1: earth_mover fused ------------------------------------------------------------
namespace {
struct EmdLayout {
    bool small;
    size_t off_rec, off_partial, off_match, off_mc, off_coff, total;  // floats
    size_t rstride;  // records per sample in the column-record array (mpad + the class padding)
    int lsplit, lspan;
};
EmdLayout emd_layout(int b, int n, int m, int mode = RF_EMD_AUTO) {
    EmdLayout E;
    E.small = n <= AM_SMALL && m <= AM_SMALL;
    E.lsplit = 1; E.lspan = 0;
    if (E.small) {
        E.off_match = 0;
        E.off_mc = (size_t)b * n * m;
        E.total = E.off_mc + rf_matchcost_workspace_bytes(b, n, m) / sizeof(float);
        E.off_rec = E.off_partial = E.off_coff = 0;
        E.rstride = 0;
        return E;
    }
    const AmLayout L = am_layout(b, n, m, 10, mode);
    // enough workgroups to fill the chip: >= 4096 of 4 waves, l-spans of whole 32-column tiles
    // (the l-spans fix the order of the cost's partial sums: under RF_EMD_SWEPT they are those of a batch of one)
    const long base = (long)(mode == RF_EMD_SWEPT ? 1 : b) * rf::ceil_div(n, TPB);
    int lsplit = 1;
    while (lsplit < 64 && base * lsplit < 4096 && L.mpad / (lsplit * 2) >= 2 * MG_TL) lsplit *= 2;
    // (the spans cover the class-sorted records of the cost-only form: up to mpad + one padding record per class)
    E.rstride = (size_t)L.mpad + 64;
    E.lspan = round_up_i(rf::ceil_div(L.mpad + 2 * EF_NCLS, lsplit), MG_TL);
    E.lsplit = rf::ceil_div(L.mpad + 2 * EF_NCLS, E.lspan);
    E.off_rec = L.total;
    E.off_partial = E.off_rec + (size_t)b * E.rstride * EF_REC + 64;
    E.off_coff = E.off_partial + (size_t)b * rf::ceil_div(n, TPB) * E.lsplit;
    E.total = E.off_coff + (size_t)b * 8 + (size_t)b * rf::ceil_div(L.mpad, EP_TPB) * 8 + 64;  // class offsets, then the workgroups' class counts
    E.off_match = E.off_mc = 0;
    return E;
}
}  // namespace

size_t rf_earth_mover_workspace_bytes(int b, int n, int m) { return rf_earth_mover_mode_workspace_bytes(b, n, m, RF_EMD_AUTO); }

size_t rf_earth_mover_mode_workspace_bytes(int b, int n, int m, int mode) {
    if (b <= 0 || n <= 0 || m <= 0 || !emd_mode_ok(mode)) return 0;
    return emd_layout(b, n, m, mode).total * sizeof(float);
}

int rf_earth_mover(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                   float *grad1, float *grad2, void *workspace, size_t workspace_bytes,
                   rf_stream_t stream) {
    return rf_earth_mover_mode(b, n, m, xyz1, xyz2, cost, grad1, grad2, workspace, workspace_bytes, stream, RF_EMD_AUTO);
}

int rf_earth_mover_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                        float *grad1, float *grad2, void *workspace, size_t workspace_bytes,
                        rf_stream_t stream, int mode) {
    if (b < 0 || b > MAX_BATCH || n < 0 || m < 0 || !emd_mode_ok(mode)) return RF_EINVAL;
    if ((grad1 == nullptr) != (grad2 == nullptr)) return RF_EINVAL;
    if (b == 0) return RF_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool want_grad = grad1 != nullptr;
    if (!cost) return RF_EINVAL;
    if (want_grad) {
        if ((size_t)b * n) RF_ZERO(grad1, sizeof(float) * 3 * (size_t)b * n, s);
        if ((size_t)b * m) RF_ZERO(grad2, sizeof(float) * 3 * (size_t)b * m, s);
    }
    if (n == 0 || m == 0) {
        RF_ZERO(cost, sizeof(float) * b, s);
        return RF_OK;
    }
    if (!xyz1 || !xyz2 || !workspace || !rf::aligned16(workspace)) return RF_EINVAL;
    if (workspace_bytes < rf_earth_mover_mode_workspace_bytes(b, n, m, mode)) return RF_EWORKSPACE;
    const EmdLayout E = emd_layout(b, n, m, mode);
    float *w = (float *)workspace;
    float lv[16];
    const int nl = default_levels(lv);
    LevelConsts lc;
    for (int v = 0; v < MAX_LEVELS; v++) lc.c[v] = v < nl ? lv[v] * kLog2e : 0.f;
    float multiL, multiR;
    am_multipliers(n, m, multiL, multiR);
    if (E.small) {
        // launch-bound sizes: one workgroup per sample builds match (<= 256 KiB) in the workspace,
        // then the ordinary match_cost(+grad) kernels read it back out of L2
        float *match = w + E.off_match;
        RF_LAUNCH("am_small", am_small_kernel, dim3(b), dim3(AM_SMALL), 0, s, n, m, nl, lc, multiL, multiR,
                  xyz1, xyz2, match);
        int st = rf_matchcost(b, n, m, xyz1, xyz2, match, cost, w + E.off_mc,
                              rf_matchcost_workspace_bytes(b, n, m), stream);
        if (st != RF_OK) return st;
        if (want_grad) {
            int lsplit = MG_LSPLIT;
            while (lsplit > 1 && m / lsplit < MG_TL) lsplit /= 2;
            const int lspan = rf::ceil_div(rf::ceil_div(m, lsplit), MG_TL) * MG_TL;
            dim3 g(rf::ceil_div(n, TPB), rf::ceil_div(m, lspan), b);
            RF_LAUNCH("mc_grad", mcg_kernel, g, dim3(TPB), 0, s, n, m, lspan, xyz1, xyz2,
                      (const float *)match, grad1, grad2);
        }
        return RF_OK;
    }
    {
        const int st = am_run_levels(b, n, m, xyz1, xyz2, nl, lc, multiL, multiR, workspace, s, mode);
        if (st != RF_OK) return st;
    }
    const AmLayout L = am_layout(b, n, m, nl, mode);
    const float *ratios = w + L.V;
    float *rec = w + E.off_rec, *partial = w + E.off_partial;
    const dim3 g(rf::ceil_div(n, TPB), E.lsplit, b);
    const bool sq = quarter_chain(lc.c, 10, true);
    if (!want_grad && sq) {  // the cost alone: the columns by the class of their last live level (emd_fused_cls_kernel)
        int *coff = (int *)(w + E.off_coff);
        int *wgcnt = coff + (size_t)b * 8;
        const dim3 gp(rf::ceil_div(L.mpad, EP_TPB), b);
        RF_LAUNCH("emd_pack_cols", emd_class_count_kernel, gp, dim3(EP_TPB), 0, s, m, L.mpad, ratios, L.V, L.bstride, L.npad, wgcnt);
        RF_LAUNCH("emd_pack_cols", emd_pack_cols_sorted_kernel, gp, dim3(EP_TPB), 0, s, m, L.mpad, xyz2, ratios, L.V, L.bstride,
                  L.npad, (const int *)wgcnt, rec, E.rstride, coff);
        const dim3 gc(rf::ceil_div(n, TPB), E.lsplit, b);
        RF_LAUNCH("emd_fused", emd_fused_cls_kernel, gc, dim3(TPB), 0, s, n, E.lspan, xyz1, (const float *)rec, E.rstride,
                  (const int *)coff, ratios, L.V, L.bstride, lc, partial);
        RF_LAUNCH("mc_final", mc_final_kernel, dim3(b), dim3(256), 0, s, (const float *)partial, (int)(gc.x * gc.y), cost);
        return RF_OK;
    } else {
        RF_LAUNCH("emd_pack_cols", emd_pack_cols_kernel, dim3(rf::ceil_div(L.mpad, 256), b), dim3(256), 0, s, m,
                  L.mpad, nl, xyz2, ratios, L.V, L.bstride, L.npad, rec, want_grad ? 0 : 1);
        if (want_grad && sq) {
            RF_LAUNCH("emd_fused_grad", (emd_fused_kernel<10, true, true, true>), g, dim3(TPB), 0, s, n, m, L.mpad, E.lspan,
                      xyz1, (const float *)rec, ratios, L.V, L.bstride, lc, partial, grad1, grad2);
        } else if (want_grad) {
            RF_LAUNCH("emd_fused_grad", (emd_fused_kernel<10, true, true, false>), g, dim3(TPB), 0, s, n, m, L.mpad, E.lspan,
                      xyz1, (const float *)rec, ratios, L.V, L.bstride, lc, partial, grad1, grad2);
        } else {
            RF_LAUNCH("emd_fused", (emd_fused_kernel<10, false, true, false>), g, dim3(TPB), 0, s, n, m, L.mpad, E.lspan, xyz1,
                      (const float *)rec, ratios, L.V, L.bstride, lc, partial, grad1, grad2);
        }
    }
    RF_LAUNCH("mc_final", mc_final_kernel, dim3(b), dim3(256), 0, s, (const float *)partial,
              (int)(g.x * g.y), cost);
    return RF_OK;
}

}  // extern "C"
