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
//                     grid), a 15-bit key per point -- sort-tile-recursive for clouds held in
//                     registers (up to 16384 points: equal-count x slabs, equal-count y strips per
//                     slab, z rank inside the strip, boustrophedon), the Hilbert
//                     curve over the equalised cells above that -- and a counting sort in LDS
//                     (32768 bins = 128 KiB of the CU's 160 KiB).  Output: the cloud's coordinates (x, y, z packed: a 16-record
//                     block is three aligned 64-byte scalar loads) and original indices in key
//                     order, padded to a multiple of 64, and the axis-aligned boxes of every
//                     16-record block and 64-record superblock.
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
// it, the second such block, and a flag raised when a THIRD block's minimum equals it bit for bit.
// Without the flag all candidates attaining the minimum sit in those one or two blocks: a re-scan
// of their 16 records takes the lowest original index among the exact matches (two blocks cover
// every duplicated pair of points, the reference resample_pcd's case: DESIGN.md 5.1f).  With the
// flag (points repeated 3+ times across blocks, symmetric configurations) the wave re-scans, for
// that one query, every superblock whose bound does not exceed the minimum, lanes across
// candidates, and reduces the lowest matching index.
//
//   nnp_grad_sorted_kernel  (rf_chamfer_step only) the backward of both directions in SORTED space:
//                     the sweep's epilogue leaves each query's winner position, its own gradient
//                     term and a 64-bit mask of the winner buckets its group touches; this kernel
//                     accumulates the scattered halves per tile of whole buckets in LDS and writes
//                     every gradient row once, in original order.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "nn_pruned.hpp"

namespace {

// tuning knobs (tools/build_variant.py compiles A/B libraries with -D...; the defaults are the product; the values a knob
// accepts are asserted below: RFP_NSH must stay 4 while RFP_TILE16 = 1).  Decided experiments -- the Hilbert order of the register
// sort, the heavy-direction-first grid order, the timing-only ablations of the sorted-space backward -- live in
// tools/experiments/nn_pruned_decided_knobs.patch.txt; the instrumented builds (scan histograms, phase stamps, cloud end times) in
// tools/experiments/nn_pruned_instrumented_builds.patch.txt.
#ifndef RFP_NSH
#define RFP_NSH 4   // waves sharing one query group when a direction has few groups
#endif
#ifndef RFP_SPLIT_BELOW
#define RFP_SPLIT_BELOW 4096
#endif
#ifndef RFP_SORT_SPLIT
#define RFP_SORT_SPLIT 2     // workgroups per cloud of more than RFP_SORT_SPLIT_ABOVE points: 1 (never split), 2 or 4
                             // (4 measures the same as 2: 26.4 vs 26.3 us at C2 -- what is left of a workgroup's time
                             // no longer scales with its share of the records)
#endif
#ifndef RFP_SORT_SPLIT_ABOVE
#define RFP_SORT_SPLIT_ABOVE 8192
#endif
#ifndef RFP_WPE
#define RFP_WPE 7   // waves per SIMD the sweep's register allocation is held to (6: no SGPR spills, 8: spills inside the loop)
#endif
#ifndef RFP_QSAMPLE
#define RFP_QSAMPLE 3  // the quantile histograms take every (RFP_QSAMPLE + 1)-th point: 3 = a quarter of the cloud
#endif
#ifndef RFP_STR_SMALL_N
#define RFP_STR_SMALL_N 4096
#endif
#ifndef RFP_CROWD_DIV
#define RFP_CROWD_DIV 4  // a cloud is flagged crowded when more than 1 / RFP_CROWD_DIV of its sort's waves are
#endif
#ifndef RFP_STR_LARGE_LEAF
#define RFP_STR_LARGE_LEAF 48
#endif
#ifndef RFP_STR_SMALL_LEAF
#define RFP_STR_SMALL_LEAF 32
#endif
#ifndef RFP_TILE16
#define RFP_TILE16 1  // directions with few query groups: 1 = a wave per 16-query tile, four lanes per query, every quad walking its
                      // OWN list of candidate blocks with gathered scans (round 4); 0 = four waves share a
                      // 64-query group and stream every needed block through SGPRs to all 64 lanes (rounds 1-3)
#endif
constexpr int NSH = RFP_NSH;
constexpr int BS = 16;             // candidates per block
constexpr int SBB = 4;             // blocks per superblock
constexpr int SB = BS * SBB;       // 64 records: one superblock = one query group = one wave
constexpr int KEYBITS = 15;        // 5 bits per axis
constexpr int NBINS = 1 << KEYBITS;
constexpr int STPB = 1024;         // sort kernel threads
constexpr int RPT = 16;            // points per thread of the register-resident sort (n <= 16384)
constexpr int HB = 256;            // equalisation histogram bins per axis
static_assert(rfp::kMaxPoints / SB <= 1024, "superblock id must fit the key low 10 bits");
// sweep_tile16 gives wave `wib` of a workgroup the tile of block `wib` of its group (qpos = (g * SBB + wib) * BS, box at
// tb + wib * 6) and merges exactly four bucket masks: with another NSH waves 4.. would run into the next group (past npad on the
// last one) or half the queries would never be written
static_assert(!RFP_TILE16 || NSH == SBB, "the quad-per-query tiles need one wave per 16-record block of a group: RFP_NSH = 4");
constexpr unsigned IDMASK = 0x3FFu;
constexpr int B16F = SBB * 6;      // floats per superblock in box16: 4 x (lo.xyz, hi.xyz)
constexpr int B64F = 8;            // floats per superblock in box64: lo.xyz, -, hi.xyz, -

// Sorted cloud, per batch element and set:
//   xyz   (npad, 3)      coordinates in key order, padding = +inf
//   orig  (npad)         original index of each record, padding = -1
//   box16 (npad/64, 24)  per superblock, for each of its 4 blocks: lo.xyz, hi.xyz
//   box64 (npad/64, 8)   lo.xyz, 0, hi.xyz, 0
struct SortArgs {
    int b;
    int n[2], npad[2];
    const float *src[2];  // (b, n, 3)
    float *xyz[2];
    int *orig[2];
    float *box16[2];
    float *box64[2];
    int nsets;  // 2 for the Chamfer sweep (both clouds of every batch element), 1 for a single set
    int split[2];  // workgroups per cloud of the set (register-resident kernel): 1, or 2 / 4 = one per slice of the key space
    int *pos0[2];  // (b) sorted position of the point with original index 0 (what an index "0" of the reference's
                   // NaN / no-candidate policy refers to, for the sorted-space backward)
    int str_s[2];  // STR order: slabs per cloud = strips per slab = round(cbrt(n / 64))
    unsigned long long *dbg;  // optional (with stats): s_memtime stamps of the sort's phases, [16..31]
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

// bin of a coordinate: (v - lo) * scale clamped to [0, HB-1].  One v_med3_f32 does the clamp; a NaN
// (NaN input, or inf * 0 when the axis has no extent) leaves it as NaN or a bound and converts to
// 0 or a bound -- any bin is acceptable, the keys only steer the order.
// (`off` = -lo * scale: one fma, one float clamp, one conversion -- the clamp leaves [0, HB - 1] or, for a NaN, whatever
// v_med3_f32 makes of it, which v_cvt_i32_f32 turns into 0 or a bound; as (v - lo) * scale with an integer clamp behind the
// conversion it was six instructions, 15 times per point of the sort)
__device__ __forceinline__ int axis_bin(float v, float off, float scale) {
    const int b = (int)__builtin_amdgcn_fmed3f(fmaf(v, scale, off), 0.f, (float)(HB - 1));
    return b & (HB - 1);  // (free insurance: a table index stays inside its table whatever the conversion returned)
}

__device__ __forceinline__ unsigned spread5(unsigned v) {  // bit b -> bit 3b
    v = (v | (v << 8)) & 0x100Fu;
    v = (v | (v << 4)) & 0x10C3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

// inclusive prefix sum over the wave in 6 DPP steps (row shifts, then the two row broadcasts)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

#define RFP_ROW(OP, N) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float wave_min_f32(float v) {  // uniform result; inputs not NaN
    RFP_ROW("v_min_f32_dpp", 8);
    RFP_ROW("v_min_f32_dpp", 4);
    RFP_ROW("v_min_f32_dpp", 2);
    RFP_ROW("v_min_f32_dpp", 1);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    RFP_ROW("v_max_f32_dpp", 8);
    RFP_ROW("v_max_f32_dpp", 4);
    RFP_ROW("v_max_f32_dpp", 2);
    RFP_ROW("v_max_f32_dpp", 1);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
#undef RFP_ROW

// LDS histogram increments on DEGENERATE clouds.  ds_add_u32 serialises over lanes that hit the same address: on a
// cloud collapsed onto a few spots (the untrained RFNet's output: 16384 points on ~120 spots, 3-5 distinct bins among
// the 64 consecutive points of a wave instruction, interleaved) the histogram phases take 2-5x their normal time
// (quantiles 8.0 k -> 28.5 k ticks, keys 11.8 k -> 18.6 k, positions 2.0 k -> 11.5 k: profiles/r04_sort_stamps.txt).
// The QUANTILE histograms only steer the order, so a wave that finds >= 8 of its lanes in one bin at its first sample
// (`hist_is_dense`) feeds them from a quarter of its lanes (a sixteenth of the cloud instead of a quarter).  The key
// histogram and the positions are exact and stay as they are: pre-aggregating them in the wave (one add per distinct
// bin found by ballot / readlane, leader lanes, ranks by mbcnt) was built and measured SLOWER than the serialised
// atomics -- quantiles 28.5 k -> 39.5 k ticks, keys 18.6 k -> 27.2 k, positions 11.5 k -> 13.0 k (profiles/r04_sort_stamps.txt).
__device__ __forceinline__ bool hist_is_dense(unsigned bin, bool act) {
    const unsigned long long todo = __builtin_amdgcn_ballot_w64(act);
    if (todo == 0ull) return false;
    const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(todo));
    return __builtin_popcountll(__builtin_amdgcn_ballot_w64(act && bin == first)) >= 8;
}

constexpr int HALF = 9216;  // records staged in LDS at a time (16 B each, over the dead histogram + 16 KiB): a
                            // split cloud's half (8192 +- the quantiles' sampling error) fits one round

// Clouds of up to RPT * STPB = 16384 points: the points are loaded ONCE into registers; every
// later phase is LDS and ALU work.  The sorted records are staged in LDS (over the histogram,
// dead once every point has its position), 8192 at a time, so that the boxes come from LDS and
// the arrays leave the CU as coalesced stores (a direct scatter is 4 x 16384 single-dword stores).
__global__ __launch_bounds__(STPB) void nnp_sort_reg_kernel(SortArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned hist[HALF * 4];  // NBINS bins, later HALF staged records
    __shared__ unsigned ahist[3][HB];
    __shared__ unsigned char slabmap[HB];        // x bin -> slab (equal mass)
    __shared__ unsigned short zmap[HB];          // z bin -> rank in 0..511 (equal mass)
    __shared__ unsigned char stripmap[16 * HB];  // (slab, y bin) -> strip of that slab (equal mass inside the slab), snaked
    __shared__ float red[STPB / 64][6];
    __shared__ unsigned wsum[STPB / 64];
    __shared__ unsigned lowcnt[STPB / 64][3];
    __shared__ float frame[6];  // lo[3], scale[3]
    __shared__ unsigned ncrowded;  // waves whose points crowd into few bins
    __shared__ unsigned badw[STPB / 64];  // per wave: a NaN or infinite coordinate among its points (query_ball_boxes reads the cloud's flag)

    // cloud-major logical order, each XCD a contiguous eighth (as the sweep: the XCD that sorts a
    // batch element is the one that sweeps it, its L2 still holding the records)
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    if (threadIdx.x == 0) ncrowded = 0;
    // A large cloud is shared by H = 2 or 4 workgroups, one per SLICE OF THE KEY SPACE (top key bits).  All
    // load the whole cloud and derive the same frame, cells and keys (no communication: the frame is a
    // pure function of the cloud); each then histograms, scans, places and writes out only the points
    // of its own half -- the half of the work that does not shrink otherwise (LDS atomics, scan,
    // staging, boxes, write-out).  Half 0's segment is padded to a multiple of 64 records and half 1
    // starts behind it, so the two never share a superblock; the set is allocated 64 records longer.
    const int wpb = a.split[0] + (a.nsets > 1 ? a.split[1] : 0);  // workgroups per batch element
    const int bi = logical / wpb;
    int rem = logical - bi * wpb;
    const int set = (a.nsets > 1 && rem >= a.split[0]) ? 1 : 0;
    if (set) rem -= a.split[0];
    const int H = a.split[set], half = rem;  // H = 1: half = 0, the whole key space
    const int n = a.n[set], npad = a.npad[set];
    const float *__restrict__ src = a.src[set] + (size_t)bi * n * 3;
    float *__restrict__ oxyz = a.xyz[set] + (size_t)bi * npad * 3;
    int *__restrict__ oorig = a.orig[set] + (size_t)bi * npad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int stamp_no = 16;
    auto stamp = [&]() {  // phase timing of the LAST workgroup (set 1, last batch element), thread 0
        if (a.dbg && bi == a.b - 1 && set == a.nsets - 1 && half == H - 1 && tid == 0 && stamp_no < 32) a.dbg[stamp_no] = clock64();
        stamp_no++;
    };
    stamp();

    float px[RPT], py[RPT], pz[RPT];
    unsigned pk[RPT];  // key, then position
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const int i = tid + k * STPB;
        px[k] = py[k] = pz[k] = 0.f;
        if (i < n) {
            // one 12-byte load per point (global_load_dwordx3: a wave's 64 consecutive points are 768 contiguous bytes);
            // three strided dword loads walk the same 12 cache lines three times and the load phase is bound by that
            struct P3 {
                float x, y, z;
            };
            const P3 v = *(const P3 *)(src + (size_t)i * 3);
            px[k] = v.x;
            py[k] = v.y;
            pz[k] = v.z;
        }
    }
    {
        uint4 *h4 = (uint4 *)hist;
#pragma unroll
        for (int k = 0; k < NBINS / 4 / STPB; k++) h4[tid + k * STPB] = make_uint4(0, 0, 0, 0);
    }
    if (tid < 3 * HB) (&ahist[0][0])[tid] = 0;
    unsigned *yhist = hist + NBINS;  // [slab][HB], in the part of `hist` that only the staging uses (dead until then)
    for (int i = tid; i < 16 * HB; i += STPB) yhist[i] = 0;

    stamp();
    // 1. bounding box of the finite coordinates.  Fast path: plain min / max (fminf / fmaxf drop NaN by themselves); only a wave
    // whose result is not finite -- an infinite coordinate, or no point at all -- repeats its pass with the per-coordinate filter
    // (wave-uniform branch; the filter is 3 of the 5 instructions per coordinate: 1.3 k of the sort's 50 k cycles)
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#define RFP_BBOX_PASS(FILTER)                                                     \
    _Pragma("unroll") for (int k = 0; k < RPT; k++) {                             \
        if (tid + k * STPB < n) {                                                 \
            const float v[3] = {px[k], py[k], pz[k]};                             \
            _Pragma("unroll") for (int c = 0; c < 3; c++) {                       \
                if (!(FILTER) || isfinite(v[c])) {                                \
                    lo[c] = fminf(lo[c], v[c]);                                   \
                    hi[c] = fmaxf(hi[c], v[c]);                                   \
                }                                                                 \
            }                                                                     \
        }                                                                         \
    }                                                                             \
    _Pragma("unroll") for (int c = 0; c < 3; c++) {                               \
        lo[c] = wave_min_f32(lo[c]);                                              \
        hi[c] = wave_max_f32(hi[c]);                                              \
    }
        RFP_BBOX_PASS(false)
        if (!(isfinite(lo[0]) && isfinite(lo[1]) && isfinite(lo[2]) && isfinite(hi[0]) && isfinite(hi[1]) && isfinite(hi[2]))) {  // (uniform)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                lo[c] = INFINITY;
                hi[c] = -INFINITY;
            }
            RFP_BBOX_PASS(true)
        }
#undef RFP_BBOX_PASS
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                red[wave][c] = lo[c];
                red[wave][3 + c] = hi[c];
            }
        }
        // any NaN or infinite coordinate in the cloud?  (v * 0 is NaN exactly then; absent points hold 0.)  The boxes below
        // exclude such points, which is all the Chamfer sweep needs; the ball query must know, because a NaN distance IS inside
        // every ball (grouping.hip).  3 VALU per point here instead of a launch of its own there.
        float nf = 0.f;
#pragma unroll
        for (int k = 0; k < RPT; k++) nf = fmaf(pz[k], 0.f, fmaf(py[k], 0.f, fmaf(px[k], 0.f, nf)));
        const unsigned long long bad_lanes = __ballot(nf != nf);  // (by the whole wave: not inside the lane-0 branch)
        if (lane == 0) badw[wave] = bad_lanes != 0ull ? 1u : 0u;  // (read behind the barriers below)
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
    const float fs[3] = {frame[3], frame[4], frame[5]};
    const float fl[3] = {-frame[0] * fs[0], -frame[1] * fs[1], -frame[2] * fs[2]};  // axis_bin's offsets

    stamp();
    // 2. per-axis histograms of a quarter of the points: the cells only need approximate
    // quantiles, and same-address LDS atomics serialise.  The choice of k is WAVE-uniform (a
    // scalar branch skips the other three quarters; wave w takes k = -w mod 4, so every index
    // range of the cloud is sampled).
    // (wave-uniform) this wave's points crowd into few bins: decided on its first sample, k = -wave mod 4
    bool dense;
    {
        const int w3 = wave & RFP_QSAMPLE;
        const float x0 = w3 == 0 ? px[0] : (w3 == 3 ? px[1] : (w3 == 2 ? px[2] : px[3]));
        const int k0 = (4 - w3) & 3;
        dense = RFP_QSAMPLE == 3 && hist_is_dense((unsigned)axis_bin(x0, fl[0], fs[0]), tid + k0 * STPB < n);
    }
    const bool feeds = !dense || (lane & 3) == 0;  // (a crowded wave: a quarter of its lanes feed the quantile histograms)
    if (dense && lane == 0) atomicAdd(&ncrowded, 1u);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        if (((k + wave) & RFP_QSAMPLE) != 0) continue;
        if (feeds && tid + k * STPB < n) {
            // (x for the slabs, z for the ranks; y is histogrammed per slab below -- the marginal y histogram of the Hilbert
            // order's cells was still being filled here until late in round 4, a third of this pass's atomics for nothing)
            atomicAdd(&ahist[0][axis_bin(px[k], fl[0], fs[0])], 1u);
            atomicAdd(&ahist[2][axis_bin(pz[k], fl[2], fs[2])], 1u);
        }
        __builtin_amdgcn_sched_barrier(0);  // (one sample at a time: interleaved, their temporaries push the kernel past its 128 registers)
    }
    __syncthreads();
    // Sort-tile-recursive order from histograms: SS slabs of equal mass along x (marginal x histogram), inside
    // every slab SS strips of equal mass along y (the slab's own y histogram), inside a strip the points by z
    // (rank of the z bin, 9 bits) -- tiles of 64 consecutive records are then near-cubic cells with disjoint boxes
    // (the Hilbert runs of rounds 1-2 are ragged unions of curve cells whose boxes overlap their neighbours': -35 %
    // superblock steps and box tests, -10 % block scans in the numpy model tools/experiments/str_model.py).  Strips
    // alternate direction from slab to slab and z from strip to strip (boustrophedon), so a tile that straddles two
    // strips stays compact; in raster order the same key is 50 % WORSE than Hilbert.
    const int SS = a.str_s[set];
    if (wave == 0 || wave == 2) {
        unsigned c4[4], sm = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c4[k] = ahist[wave][lane * 4 + k];
            sm += c4[k];
        }
        const unsigned incl = wave_incl_scan(sm);
        const unsigned total = max((unsigned)__builtin_amdgcn_readlane((int)incl, 63), 1u);
        unsigned run = incl - sm;
        // (equal-mass cells by ONE float reciprocal per table instead of an integer division per bin: a cell boundary may move by
        // a histogram bin against the exact quotient -- the order only steers the culling -- and every workgroup of a cloud
        // computes the same floats)
        const float per = (wave == 0 ? (float)SS : 512.f) / (float)total;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned q = (unsigned)((float)run * per);
            if (wave == 0) {
                slabmap[lane * 4 + k] = (unsigned char)(q > (unsigned)SS - 1u ? (unsigned)SS - 1u : q);
            } else {
                zmap[lane * 4 + k] = (unsigned short)(q > 511u ? 511u : q);
            }
            run += c4[k];
        }
    }
    __syncthreads();
    // the y histogram of every slab, from the same quarter of the points
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        if (((k + wave) & RFP_QSAMPLE) != 0) continue;
        if (feeds && tid + k * STPB < n)
            atomicAdd(&yhist[(int)slabmap[axis_bin(px[k], fl[0], fs[0])] * HB + axis_bin(py[k], fl[1], fs[1])], 1u);
    }
    __syncthreads();
    if (wave < SS) {  // wave <-> slab
        unsigned c4[4], sm = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c4[k] = yhist[wave * HB + lane * 4 + k];
            sm += c4[k];
        }
        const unsigned incl = wave_incl_scan(sm);
        const unsigned total = max((unsigned)__builtin_amdgcn_readlane((int)incl, 63), 1u);
        unsigned run = incl - sm;
        const float per = (float)SS / (float)total;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned st = (unsigned)((float)run * per);
            st = st > (unsigned)SS - 1u ? (unsigned)SS - 1u : st;
            stripmap[wave * HB + lane * 4 + k] = (unsigned char)((wave & 1) ? (unsigned)SS - 1u - st : st);
            run += c4[k];
        }
    }
    __syncthreads();
    // (yhist's words are staging space again from phase 6 on; the histogram proper is the first NBINS words)
    const int ncol = SS * SS;
    auto col_start = [&](int q) { return (q * ncol + H - 1) / H; };  // first column of slice q (H slices of equal column count)
    const int cs1 = H > 1 ? col_start(1) : 0x7FFFFFFF, cs2 = H > 2 ? col_start(2) : 0x7FFFFFFF,
              cs3 = H > 3 ? col_start(3) : 0x7FFFFFFF;  // (uniform; the divisions stay out of the per-point loop)
    const unsigned kbase = (unsigned)col_start(half) << 9;
    const int nb_local = (col_start(half + 1) - col_start(half)) << 9;

    stamp();
    unsigned below[3] = {0u, 0u, 0u};  // wave-uniform counters
    // 3. keys and their histogram: three table lookups and an atomic per point -- the LDS pipe, 12 k of the sort's 45 k cycles.
    // (Looking the strip / z rank up only in the lanes whose point may be / is this workgroup's own -- the slab alone decides
    // that outside the one slab the cut runs through -- was built and is SLOWER, 16.2 k cycles: the lookups then sit in
    // per-lane branches, one wait each, instead of 48 independent reads the scheduler interleaves.)
#pragma unroll
    for (int k = 0; k < RPT; k++) {
        const bool valid = tid + k * STPB < n;
        const int slab = slabmap[axis_bin(px[k], fl[0], fs[0])];
        const int col = slab * SS + (int)stripmap[slab * HB + axis_bin(py[k], fl[1], fs[1])];
        unsigned zq = zmap[axis_bin(pz[k], fl[2], fs[2])];
        // a crowded wave (near-copies on a few spots) spreads its points over 16 adjacent z ranks by lane: the key
        // histogram and the positions are exact and their same-address LDS atomics serialise -- 46 of 64 lanes on one
        // bin otherwise; copies of one spot are the same place, so the ORDER among them is free and culling loses nothing
        if (dense) zq = (zq & ~15u) | (unsigned)(lane & 15);
        const unsigned key = ((unsigned)col << 9) | ((col & 1) ? 511u - zq : zq);
        int slice = col >= cs1;  // (two workgroups per cloud, the product's split: one comparison)
        if (H > 2) slice += (col >= cs2) + (col >= cs3);  // (uniform)
        const bool own = valid && slice == half;
        pk[k] = own ? key - kbase : 0xFFFFFFFFu;
        if (own) atomicAdd(&hist[pk[k]], 1u);
        // points of the slices below this workgroup's (its segment starts behind theirs)
        if (H > 1) {
#pragma unroll
            for (int q = 0; q < 3; q++)
                if (q < half) below[q] += (unsigned)__builtin_popcountll(__ballot(valid && slice == q));
        }
    }
    if (H > 1 && lane == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) lowcnt[wave][q] = below[q];
    }
    __syncthreads();

    stamp();
    // 4. exclusive scan of the 32768 bins: each wave owns 2048 consecutive bins, 8 steps of 256
    // (4 per lane, one ds_read_b128: consecutive lanes on consecutive banks)
    int cown = 0;  // points of this workgroup's half
    {
        constexpr int STEPS = NBINS / (STPB / 64) / 256;  // 8
        const int steps = min(STEPS, (nb_local + 4095) / 4096);  // each wave owns steps * 256 consecutive bins
        uint4 *h4 = (uint4 *)hist + (size_t)wave * steps * 64;
        uint4 v[STEPS];
        unsigned tot = 0;
#pragma unroll
        for (int it = 0; it < STEPS; it++) {
            v[it] = it < steps ? h4[it * 64 + lane] : make_uint4(0, 0, 0, 0);
            tot += v[it].x + v[it].y + v[it].z + v[it].w;
        }
        tot = wave_incl_scan(tot);
        if (lane == 63) wsum[wave] = tot;
        __syncthreads();
        // the 16 wave totals: ONE LDS read per lane and a wave scan (a loop over wsum[] is a chain of
        // dependent LDS latencies in every wave)
        const unsigned wtot = lane < STPB / 64 ? wsum[lane] : 0u;
        const unsigned wincl = wave_incl_scan(wtot);
        unsigned carry = (unsigned)__builtin_amdgcn_readlane((int)(wincl - wtot), __builtin_amdgcn_readfirstlane(wave));
        cown = __builtin_amdgcn_readlane((int)wincl, STPB / 64 - 1);
#pragma unroll
        for (int it = 0; it < STEPS; it++) {
            if (it < steps) {  // uniform
                const unsigned sm = v[it].x + v[it].y + v[it].z + v[it].w;
                const unsigned incl = wave_incl_scan(sm);
                uint4 out;
                out.x = carry + incl - sm;
                out.y = out.x + v[it].x;
                out.z = out.y + v[it].y;
                out.w = out.z + v[it].z;
                h4[it * 64 + lane] = out;
                carry += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
    }
    __syncthreads();

    stamp();
    // 5. positions (the order inside a key is whatever the atomics give: results do not depend on it)
#pragma unroll
    for (int k = 0; k < RPT; k++)
        if (pk[k] != 0xFFFFFFFFu) pk[k] = atomicAdd(&hist[pk[k]], 1u);  // position inside this half's segment
    __syncthreads();  // the histogram is dead from here on

    stamp();
    // 6. per half of 8192 records: stage in LDS as (x, y, z, original index) records, boxes from
    // LDS, coalesced write-out
    float4 *stg = (float4 *)hist;
    float *__restrict__ b16 = a.box16[set] + (size_t)bi * (npad / SB) * B16F;
    float *__restrict__ b64 = a.box64[set] + (size_t)bi * (npad / SB) * B64F;
    // this workgroup's segment of the sorted set: records [base, base + seglen), of which the first
    // `cown` are points and the rest padding.  Half 0: [0, roundup(cown, 64)); half 1 (or the only
    // workgroup): from behind half 0's segment to the end of the set.
    int base = 0;  // every slice's segment is padded to a multiple of 64 records on its own
    for (int q = 0; q < half; q++) {
        unsigned c = 0;
        for (int w = 0; w < STPB / 64; w++) c += lowcnt[w][q];
        base += ((int)c + SB - 1) / SB * SB;
    }
    const int seglen = ((cown + SB - 1) / SB) * SB;  // the points and the padding of their last superblock
    // whole superblocks of padding behind the last segment (a split set is allocated one more than
    // it may need): written straight to memory, no staging round for them
    if (half == H - 1) {
        const int t0 = base + seglen;
        for (int j = t0 * 3 + tid; j < npad * 3; j += STPB) oxyz[j] = INFINITY;
        for (int j = t0 + tid; j < npad; j += STPB) oorig[j] = -1;
        for (int j = (t0 / SB) * B16F + tid; j < (npad / SB) * B16F; j += STPB) b16[j] = ((j % 6) < 3) ? INFINITY : -INFINITY;
        for (int j = (t0 / SB) * B64F + tid; j < (npad / SB) * B64F; j += STPB) {
            const int c = j % B64F;
            b64[j] = c < 3 ? INFINITY : (c >= 4 && c < 7 ? -INFINITY : 0.f);
        }
    }
    if (tid == 0 && pk[0] != 0xFFFFFFFFu) a.pos0[set][bi] = base + (int)pk[0];  // point 0 is thread 0's first
    // CROWDED cloud (more than a quarter of the waves found >= 8 of their 64 consecutive points in one x bin -- a test that
    // waves of ordinary clouds practically never pass, so a low bar costs nothing and misses fewer half-crowded clouds: the untrained
    // network's collapsed output): as a CANDIDATE set it sends every query through hundreds of near-tied blocks, which the
    // shared-group sweep streams to 64 lanes at the VALU rate and the quad tiles would chase one latency-bound round at a
    // time (258 us instead of 120 at C2) -- the sweep reads this flag per cloud (all workgroups of a cloud agree: same data)
    if (tid == 0 && half == 0) {
        a.pos0[set][a.b + bi] = ncrowded * RFP_CROWD_DIV > STPB / 64 ? 1 : 0;
        unsigned nbad = 0;
        for (int w = 0; w < STPB / 64; w++) nbad |= badw[w];
        a.pos0[set][2 * a.b + bi] = nbad != 0u ? 1 : 0;
    }
    for (int h0 = 0; h0 < seglen; h0 += HALF) {
        const int cnt = min(HALF, seglen - h0);  // multiple of 64
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = tid + k * STPB;
            const int p = (int)pk[k] - h0;
            if (pk[k] != 0xFFFFFFFFu && p >= 0 && p < HALF) stg[p] = make_float4(px[k], py[k], pz[k], __int_as_float(i));
        }
        for (int p = cown - h0 + tid; p < cnt; p += STPB)  // padding records live at positions >= cown
            if (p >= 0) stg[p] = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
        __syncthreads();
        stamp();
        // boxes: TWO threads per 16-record block (8 records each), an octet of lanes per superblock (cnt / 16 is a multiple of 4:
        // octets are complete and inside one wave).  Record (u + block + 4 half) % 8 of the half at step u: the two halves of a
        // block and blocks 8 apart would otherwise meet in the same banks.  (One thread per block left half the workgroup idle
        // behind a chain of 16 dependent-issue LDS reads: 3.8 k of the sort's 48 k cycles.)
        // (a split half may hold up to HALF = 9216 records: 1152 half-blocks for 1024 threads -- a second trip for the first lanes)
#pragma unroll 1
        for (int t2 = tid; t2 < cnt / BS * 2; t2 += STPB) {
            const int blk = t2 >> 1, hf = t2 & 1;
            float l[3] = {INFINITY, INFINITY, INFINITY}, hh[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4  // (the points' 64 registers stay live across a staging round: 8 records in flight spill)
            for (int u = 0; u < BS / 2; u++) {
                const int r = blk * BS + hf * (BS / 2) + ((u + blk + 4 * hf) & (BS / 2 - 1));
                const float4 v = stg[r];
                if (h0 + r < cown) {  // padding excluded; NaN coordinates drop out of fminf/fmaxf
                    l[0] = fminf(l[0], v.x); hh[0] = fmaxf(hh[0], v.x);
                    l[1] = fminf(l[1], v.y); hh[1] = fmaxf(hh[1], v.y);
                    l[2] = fminf(l[2], v.z); hh[2] = fmaxf(hh[2], v.z);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                l[c] = fminf(l[c], __shfl_xor(l[c], 1, 64));
                hh[c] = fmaxf(hh[c], __shfl_xor(hh[c], 1, 64));
            }
            const int gblk = (base + h0) / BS + blk;
            if (hf == 0) {
                float *o = b16 + (size_t)(gblk >> 2) * B16F + (gblk & 3) * 6;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    o[c] = l[c];
                    o[3 + c] = hh[c];
                }
            }
#pragma unroll
            for (int x = 2; x <= 4; x <<= 1) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    l[c] = fminf(l[c], __shfl_xor(l[c], x, 64));
                    hh[c] = fmaxf(hh[c], __shfl_xor(hh[c], x, 64));
                }
            }
            if ((gblk & 3) == 0 && hf == 0) {
                float *o64 = b64 + (size_t)(gblk >> 2) * B64F;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    o64[c] = l[c];
                    o64[4 + c] = hh[c];
                }
                o64[3] = o64[7] = 0.f;
            }
        }
        stamp();
        {
            // write-out: one 16-byte LDS read per record (consecutive lanes, consecutive records: conflict-free), out as a 12-byte
            // store into the packed (x, y, z) stream -- a wave's 64 records are 768 contiguous bytes, as on the load side -- and a
            // 4-byte store of the original index: 8 reads + 16 stores per thread.  (Per dword it was 32 + 32: 4.4 k cycles; as
            // 16-byte stores of 4 consecutive floats the LDS reads are strided by 16 / 3 words and conflict: slower still.)
            struct P3 {
                float x, y, z;
            };
            for (int j = tid; j < cnt; j += STPB) {
                const float4 v = stg[j];
                *(P3 *)(oxyz + (size_t)(base + h0 + j) * 3) = P3{v.x, v.y, v.z};
                oorig[base + h0 + j] = __float_as_int(v.w);
            }
        }
        __syncthreads();
        stamp();
    }
}

// REG: the cloud (n <= RPT * STPB points) is loaded once into registers -- every later phase is
// LDS and ALU work only; otherwise each phase re-reads the points (L2-resident) from `src`.
template <bool REG>
__global__ __launch_bounds__(STPB) void nnp_sort_kernel(SortArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned hist[NBINS];
    __shared__ unsigned ahist[3][HB];
    __shared__ unsigned char cellmap[3][HB];
    __shared__ float red[STPB / 64][6];
    __shared__ unsigned wsum[STPB / 64];
    __shared__ float frame[6];  // lo[3], scale[3]
    __shared__ unsigned nbad;   // != 0: a NaN or infinite coordinate in the cloud (see nnp_sort_reg_kernel)

    // cloud-major logical order, each XCD a contiguous eighth (as the sweep: the XCD that sorts a
    // batch element is the one that sweeps it, its L2 still holding the records)
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    const int bi = logical / a.nsets, set = logical - bi * a.nsets;
    if (threadIdx.x == 0) nbad = 0;
    const int n = a.n[set], npad = a.npad[set];
    const float *__restrict__ src = a.src[set] + (size_t)bi * n * 3;
    float *__restrict__ oxyz = a.xyz[set] + (size_t)bi * npad * 3;
    int *__restrict__ oorig = a.orig[set] + (size_t)bi * npad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float px[REG ? RPT : 1], py[REG ? RPT : 1], pz[REG ? RPT : 1];
    unsigned pkey[REG ? RPT : 1];
    if constexpr (REG) {
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int i = tid + k * STPB;
            px[k] = py[k] = pz[k] = 0.f;
            if (i < n) {
                px[k] = src[(size_t)i * 3 + 0];
                py[k] = src[(size_t)i * 3 + 1];
                pz[k] = src[(size_t)i * 3 + 2];
            }
        }
    }
    // f(k, i, x, y, z) for every point of this thread; k indexes the register copy (REG only)
    auto for_points = [&](auto f) {
        if constexpr (REG) {
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int i = tid + k * STPB;
                if (i < n) f(k, i, px[k], py[k], pz[k]);
            }
        } else {
            for (int i = tid; i < n; i += STPB) f(0, i, src[(size_t)i * 3 + 0], src[(size_t)i * 3 + 1], src[(size_t)i * 3 + 2]);
        }
    };

    for (int i = tid; i < NBINS; i += STPB) hist[i] = 0;
    if (tid < 3 * HB) (&ahist[0][0])[tid] = 0;

    // 1. bounding box of the finite coordinates
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    __syncthreads();  // nbad = 0 above
    for_points([&](int, int, float x, float y, float z) {
        const float v[3] = {x, y, z};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (isfinite(v[c])) {
                lo[c] = fminf(lo[c], v[c]);
                hi[c] = fmaxf(hi[c], v[c]);
            } else {
                bad = true;
            }
        }
    });
    if (__ballot(bad) != 0ull && lane == 0) nbad = 1u;
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
    const float fs[3] = {frame[3], frame[4], frame[5]};
    const float fl[3] = {-frame[0] * fs[0], -frame[1] * fs[1], -frame[2] * fs[2]};  // axis_bin's offsets

    // 2. per-axis histograms of a quarter of the points (every 4th, staggered over the threads:
    // the cells only need approximate quantiles, and same-address LDS atomics serialise)
    for_points([&](int, int i, float x, float y, float z) {
        if ((((unsigned)i >> 10) + (unsigned)i) & 3u) return;
        atomicAdd(&ahist[0][axis_bin(x, fl[0], fs[0])], 1u);
        atomicAdd(&ahist[1][axis_bin(y, fl[1], fs[1])], 1u);
        atomicAdd(&ahist[2][axis_bin(z, fl[2], fs[2])], 1u);
    });
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
        const unsigned total = max(__shfl(incl, 63, 64), 1u);
        unsigned run = incl - s;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // cell of a bin = the 1/32-quantile its first sample falls into
            const unsigned cell = (run * 32u) / total;  // run <= 65536: no overflow
            cellmap[wave][lane * 4 + k] = (unsigned char)(cell > 31u ? 31u : cell);
            run += c4[k];
        }
    }
    __syncthreads();

    auto key_of = [&](float x, float y, float z) {
        const unsigned cx = cellmap[0][axis_bin(x, fl[0], fs[0])];
        const unsigned cy = cellmap[1][axis_bin(y, fl[1], fs[1])];
        const unsigned cz = cellmap[2][axis_bin(z, fl[2], fs[2])];
        return hilbert15(cx, cy, cz);
    };

    // 3. key histogram
    for_points([&](int k, int, float x, float y, float z) {
        const unsigned key = key_of(x, y, z);
        if constexpr (REG) pkey[k] = key;
        atomicAdd(&hist[key], 1u);
    });
    __syncthreads();

    // 4. exclusive scan of the 32768 bins.  Each wave owns 2048 consecutive bins and walks them in
    // 8 steps of 256 (4 per lane, one ds_read_b128: consecutive lanes, consecutive banks -- a
    // thread-per-32-bins layout would put all 64 lanes on two banks).
    {
        constexpr int STEPS = NBINS / STPB / 4 * 0 + (NBINS / (STPB / 64) / 256);  // 8
        uint4 *h4 = (uint4 *)hist + (size_t)wave * (NBINS / (STPB / 64) / 4);
        unsigned tot = 0;
#pragma unroll
        for (int it = 0; it < STEPS; it++) {
            const uint4 v = h4[it * 64 + lane];
            tot += v.x + v.y + v.z + v.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if (lane == 0) wsum[wave] = tot;
        __syncthreads();
        unsigned carry = 0;
        for (int w = 0; w < wave; w++) carry += wsum[w];
#pragma unroll
        for (int it = 0; it < STEPS; it++) {
            const uint4 v = h4[it * 64 + lane];
            const unsigned sm = v.x + v.y + v.z + v.w;
            unsigned incl = sm;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            uint4 out;
            out.x = carry + incl - sm;
            out.y = out.x + v.x;
            out.z = out.y + v.y;
            out.w = out.z + v.z;
            h4[it * 64 + lane] = out;
            carry += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();

    // 5. scatter (the order inside a key is whatever the atomics give: results do not depend on it)
    for_points([&](int k, int i, float x, float y, float z) {
        unsigned key;
        if constexpr (REG) key = pkey[k];
        else key = key_of(x, y, z);
        const unsigned pos = atomicAdd(&hist[key], 1u);
        if (i == 0) {
            a.pos0[set][bi] = (int)pos;
            a.pos0[set][a.b + bi] = 0;  // (crowded flag: the register-resident sort's test only)
            a.pos0[set][2 * a.b + bi] = nbad != 0u ? 1 : 0;
        }
        oxyz[(size_t)pos * 3 + 0] = x;
        oxyz[(size_t)pos * 3 + 1] = y;
        oxyz[(size_t)pos * 3 + 2] = z;
        oorig[pos] = i;
    });
    for (int i = n + tid; i < npad; i += STPB) {
        oxyz[(size_t)i * 3 + 0] = INFINITY;
        oxyz[(size_t)i * 3 + 1] = INFINITY;
        oxyz[(size_t)i * 3 + 2] = INFINITY;
        oorig[i] = -1;
    }
    // (the barrier's workgroup-scope release/acquire is enough: the records are re-read by this
    // workgroup only; an agent-scope __threadfence() here writes back the whole L2 -- 40 us)
    __syncthreads();

    // 6. boxes of the 16-record blocks and the 64-record superblocks (padding and NaN excluded):
    // a quad of lanes per superblock, one block each (npad/16 and STPB are multiples of 4, so a
    // quad is always complete and inside one wave)
    float *__restrict__ b16 = a.box16[set] + (size_t)bi * (npad / SB) * B16F;
    float *__restrict__ b64 = a.box64[set] + (size_t)bi * (npad / SB) * B64F;
    for (int blk = tid; blk < npad / BS; blk += STPB) {
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        const float *p = oxyz + (size_t)blk * BS * 3;
        const int *po = oorig + (size_t)blk * BS;
        for (int u = 0; u < BS; u++) {
            if (po[u] >= 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    l[c] = fminf(l[c], p[u * 3 + c]);
                    h[c] = fmaxf(h[c], p[u * 3 + c]);
                }
            }
        }
        float *o = b16 + (size_t)(blk >> 2) * B16F + (blk & 3) * 6;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            o[c] = l[c];
            o[3 + c] = h[c];
        }
#pragma unroll
        for (int x = 1; x <= 2; x <<= 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                l[c] = fminf(l[c], __shfl_xor(l[c], x, 64));
                h[c] = fmaxf(h[c], __shfl_xor(h[c], x, 64));
            }
        }
        if ((blk & 3) == 0) {
            float *o64 = b64 + (size_t)(blk >> 2) * B64F;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                o64[c] = l[c];
                o64[4 + c] = h[c];
            }
            o64[3] = o64[7] = 0.f;
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
    int nw[2];       // waves per query group: 1 or NSH
    int wg0, wg1;    // workgroups per batch element of direction 0 / 1
    int kstride;     // key-list entries per wave (dynamic LDS: waves * kstride * 4 bytes)
};

// What the sweep of rf_chamfer_step leaves behind for the sorted-space backward (nnp_grad_sorted_kernel), per
// query and in SORTED query order -- coalesced stores from the lane that owns the query:
//   one 16-byte record {wp, own.x, own.y, own.z} (round 4: one store, one load per query; rounds 3: two arrays):
//   wp   sorted position of its nearest neighbour in the other set (-1: padding)
//   own  its own gradient term (q - winner) * 2 gd, three floats (the winner's coordinates are one gather at
//        the end of the wave's life, hidden under the other waves' scans).  The term the query scatters into
//        its winner's gradient is exactly -own (tf_nndistance_g.cu:146-151: the same product, subtracted), so
//        the backward needs no coordinates at all.
// and per 64-query group a 64-bit mask of the BUCKETS of the other set its winners fall into (bucket = sorted
// position / qbucket, 64 buckets per set): a destination tile of the backward -- a whole number of buckets --
// then visits only the groups whose mask meets its own.  gd: (b, n[dir]) upstream gradients in original order.
// Few kernel arguments on purpose (the sweep is SGPR-bound): the three arrays of both sets live in ONE buffer
// whose layout is a function of (b, npad) -- emit_layout() below, host and device alike -- and pos0 (the sorted
// position of original index 0) sits behind box64 inside the sorted set.
struct GradEmit {
    const float *gd[2];
    char *base;
    int qbucket[2];  // sorted positions per bucket of set d: ceil(npad[d] / 64)
};
struct EmitView {
    int4 *rec;                 // (b, npad) one 16-byte record per query: {wp, own.x, own.y, own.z (float bits)}
    unsigned long long *mask;  // (b, npad / 64)
};
__host__ __device__ inline size_t emit_align(size_t v) { return (v + 255) / 256 * 256; }
__host__ __device__ inline size_t emit_set_bytes(int b, int npad) {
    return emit_align((size_t)b * npad * sizeof(int4)) + emit_align((size_t)b * (npad / 64) * sizeof(unsigned long long));
}
__host__ __device__ inline EmitView emit_layout(char *base, int b, int npad0, int npad1, int set) {
    char *p = base + (set ? emit_set_bytes(b, npad0) : 0);
    const int npad = set ? npad1 : npad0;
    EmitView v;
    v.rec = (int4 *)p;
    p += emit_align((size_t)b * npad * sizeof(int4));
    v.mask = (unsigned long long *)p;
    return v;
}

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

// lower bound of d2 between point q and the box [lo,hi], same instruction sequence as d2
__device__ __forceinline__ float box_bound(float qx, float qy, float qz, float lx, float ly, float lz, float hx,
                                           float hy, float hz) {
    const float gx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
    const float gy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
    const float gz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
    return rf::d2_fma(gx, gy, gz);
}
// box to box: gap per axis between [alo,ahi] and [blo,bhi]
__device__ __forceinline__ float boxbox_bound(const float *alo, const float *ahi, const float4 blo, const float4 bhi) {
    const float gx = fmaxf(fmaxf(blo.x - ahi[0], alo[0] - bhi.x), 0.f);
    const float gy = fmaxf(fmaxf(blo.y - ahi[1], alo[1] - bhi.y), 0.f);
    const float gz = fmaxf(fmaxf(blo.z - ahi[2], alo[2] - bhi.z), 0.f);
    return rf::d2_fma(gx, gy, gz);
}

// minimum of d2 over the 8 records r[0..23] (x,y,z packed), folded into cm
__device__ __forceinline__ float scan8(const float (&r)[24], float qx, float qy, float qz, float cm) {
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
        const float d0 = rf::d2_fma(r[u * 3 + 0] - qx, r[u * 3 + 1] - qy, r[u * 3 + 2] - qz);
        const float d1 = rf::d2_fma(r[u * 3 + 3] - qx, r[u * 3 + 4] - qy, r[u * 3 + 5] - qz);
        cm = min3_acc(cm, d0, d1);
    }
    return cm;
}

// One query group (64 sorted points of set `dir`, one per lane) against set 1-dir: nearest
// neighbour of every point -> dist_dir, idx_dir (b, n[dir]).  SHARED4: the 4 waves of the
// workgroup work on the same group, wave `wib` taking every 4th candidate superblock.
// stats (optional): [dir][4] = waves, superblock steps, max steps of a wave, block scans; [8+dir] = max scans.
template <bool SHARED4, bool GRAD>
__device__ __forceinline__ void sweep_group(
    const SweepArgs &a, const GradEmit &ge, const int dir, const int gid, const int wib, const int lane,
    unsigned *__restrict__ keys_dyn, int *shbest, float (*md)[64], unsigned (*mi)[64], int (*mp)[64],
    const float *__restrict__ xyz0, const float *__restrict__ xyz1,
    const int *__restrict__ orig0, const int *__restrict__ orig1, const float *__restrict__ b16_0,
    const float *__restrict__ b16_1, const float *__restrict__ b64_0, const float *__restrict__ b64_1,
    float *__restrict__ dist0, float *__restrict__ dist1, int *__restrict__ idx0, int *__restrict__ idx1,
    unsigned long long *__restrict__ stats) {
    const int cd = 1 - dir;
    constexpr bool shared4 = SHARED4;
    const int G = a.groups[dir];
    const int sub = shared4 ? wib : 0, nsub = shared4 ? NSH : 1;
    const int bi = gid / G, g = gid - bi * G;

    const float *__restrict__ Q = (dir ? xyz1 : xyz0) + (size_t)bi * a.npad[dir] * 3;
    const int *__restrict__ Qo = (dir ? orig1 : orig0) + (size_t)bi * a.npad[dir];
    const float *__restrict__ C = (dir ? xyz0 : xyz1) + (size_t)bi * a.npad[cd] * 3;
    const int *__restrict__ Co = (dir ? orig0 : orig1) + (size_t)bi * a.npad[cd];
    const int nsb = a.npad[cd] / SB;
    const float *__restrict__ CB16 = (dir ? b16_0 : b16_1) + (size_t)bi * nsb * B16F;
    const float *__restrict__ CB64 = (dir ? b64_0 : b64_1) + (size_t)bi * nsb * B64F;

    const float qx = Q[(size_t)(g * SB + lane) * 3 + 0];
    const float qy = Q[(size_t)(g * SB + lane) * 3 + 1];
    const float qz = Q[(size_t)(g * SB + lane) * 3 + 2];
    const int qorig = Qo[g * SB + lane];
    const bool valid = qorig >= 0;
    // A query with a NaN coordinate can never tighten its bound (every d2 is NaN): it takes no part
    // in the traversal -- it would drag its whole wave through every superblock -- and is written as
    // (NaN, 0), what the reference returns for it (tf_nndistance_g.cu:27-31).
    const bool qnan = qx != qx || qy != qy || qz != qz;
    const bool part = valid && !qnan;
    const float *gb = (dir ? b64_1 : b64_0) + ((size_t)bi * G + g) * B64F;  // uniform
    const float glo[3] = {gb[0], gb[1], gb[2]}, ghi[3] = {gb[4], gb[5], gb[6]};

    const int nmine = (nsb - sub + nsub - 1) / nsub;
    unsigned *__restrict__ keys = keys_dyn + (size_t)wib * a.kstride;
    if (shared4) {
        if (wib == 0) shbest[lane] = 0x7F800000;  // +inf
        __syncthreads();
    }

    float best = INFINITY;  // this wave's own minimum, the block that first attained it, tie flag
    int bblk = 0, bblk2 = -1;  // the block that first attained the minimum, the second one that equalled it
    bool tie = false;
    // <= best: also what the other waves of the group have found.  -inf for lanes that take no part
    // (padding): `bound <= cull` / `cull >= bound` are then false without a separate mask.
    float cull = part ? INFINITY : -INFINITY;
    unsigned n_step = 0, n_scan = 0;
    auto stamp = [&](int) {};

    // One traversal of the candidate superblocks in ascending order of a lower bound, for the lanes
    // with `part` set, whose box is [blo, bhi].
    //   TRACK = false: the search proper.  Updates best / bblk / tie / cull.
    //   TRACK = true:  second pass for the lanes whose minimum was attained in more than one
    //                  visited block (duplicated points, symmetric configurations): the minima are
    //                  final (cull == best), every block that can hold a candidate with d2 == best
    //                  is visited again, and the lowest original index among the exact matches is
    //                  reduced into besti2.  (A per-lane wave-cooperative re-scan was 50x slower
    //                  than the dense sweep on clouds of identical points.)
    // Lower bound box <-> candidate superblock, truncated (downwards) into the high 22 bits of a
    // key whose low 10 bits are the superblock id: the wave minimum of the keys is the next
    // superblock.  Entry e of this wave's list is superblock sub + nsub*e; lane e % 64 owns it
    // (writes it, consumes it, keeps the minimum of its entries in `lmin`).
    auto traverse = [&](auto track_c, const float *blo, const float *bhi, unsigned &besti2, int &wpos2) {
        constexpr bool TRACK = decltype(track_c)::value;
        // up to 320 entries (every cloud that fits the register-resident sort: 16384 points = 256
        // superblocks, 257 when two workgroups sorted it): the lane's entries e = lane + 64 i live in
        // KK registers and the LDS list is not used at all
        constexpr int KK = 5;
        const bool inreg = nmine <= 64 * KK;  // uniform
        unsigned kk[KK] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        auto kk_min = [&]() { return min(min(min(kk[0], kk[1]), min(kk[2], kk[3])), kk[4]); };
        auto entry_key = [&](int e, const float *lo3, const float *hi3) {
            const int s = sub + nsub * e;
            const float4 *cb = (const float4 *)(CB64 + (size_t)s * B64F);
            const float lb = boxbox_bound(lo3, hi3, cb[0], cb[1]);
            return (__float_as_uint(lb) & ~IDMASK) | (unsigned)s;
        };
        unsigned lmin = 0xFFFFFFFFu;
        if (nmine <= 64) {  // uniform: one entry per lane
            kk[0] = lane < nmine ? entry_key(lane, blo, bhi) : 0xFFFFFFFFu;
            lmin = kk[0];
        } else if (inreg) {
            // (clamped entries, select afterwards: under `e < nmine` every round is a branch with its own wait -- five dependent
            // round trips for a 16384-point candidate cloud)
#pragma unroll
            for (int i = 0; i < KK; i++) {
                const int e = lane + 64 * i;
                if constexpr (TRACK) {  // (the rare second traversal keeps the round-by-round form: it holds more state, the batch spills)
                    if (i * 64 < nmine) kk[i] = e < nmine ? entry_key(e, blo, bhi) : 0xFFFFFFFFu;
                } else {
                    const unsigned key = entry_key(min(e, nmine - 1), blo, bhi);
                    kk[i] = e < nmine ? key : 0xFFFFFFFFu;
                }
            }
            lmin = kk_min();
        } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (int e = lane; e < nmine; e += 64) {
                const unsigned key = entry_key(e, blo, bhi);
                keys[e] = key;
                lmin = min(lmin, key);
            }
        }
        int nact_ref = 64;  // active lanes when the keys were last (re)computed
        stamp(TRACK ? 4 : 0);
        for (;;) {
            // the step's serial chain (reduction, box load, tests) goes ahead of other waves' scans:
            // -3 % at 16384^2, nothing at C2; the opposite priority is 3 % slower
            __builtin_amdgcn_s_setprio(2);
            unsigned kmin = wave_min_u32(lmin);
            if (kmin == 0xFFFFFFFFu) break;
            if (shared4 && !TRACK) cull = fminf(cull, __int_as_float(shbest[lane]));
            float bound = __uint_as_float(kmin & ~IDMASK);
            // A lane whose minimum is below the bound of every remaining superblock is finished for
            // good (the keys ascend).  The traversal ends when no lane is left ...
            const unsigned long long act = __builtin_amdgcn_ballot_w64(cull >= bound);
            if (act == 0ull) break;  // every remaining superblock is strictly farther than every lane's minimum
            // ... and when half of the lanes have finished since the keys were computed, the box of
            // the remaining lanes replaces the previous box: a far outlier no longer keeps the
            // bounds of all 64 queries loose (the heaviest waves were 6x the average, and they set
            // the kernel's duration).  The new keys are bounds for the active lanes only, which is
            // all that is left.
            const int nact = __builtin_popcountll(act);
            if (nact * 2 <= nact_ref) {  // (thresholds between 1/4 and 7/8 measure the same)
                nact_ref = nact;
                const bool on = (act >> lane) & 1ull;
                const float alo[3] = {wave_min_f32(on ? qx : INFINITY), wave_min_f32(on ? qy : INFINITY),
                                      wave_min_f32(on ? qz : INFINITY)};
                const float ahi[3] = {wave_max_f32(on ? qx : -INFINITY), wave_max_f32(on ? qy : -INFINITY),
                                      wave_max_f32(on ? qz : -INFINITY)};
                lmin = 0xFFFFFFFFu;
                if (inreg) {
#pragma unroll
                    for (int i = 0; i < KK; i++)
                        if (i * 64 < nmine && kk[i] != 0xFFFFFFFFu) kk[i] = entry_key(lane + 64 * i, alo, ahi);  // (consumed ones stay)
                    lmin = kk_min();
                } else {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                    for (int e = lane; e < nmine; e += 64) {
                        if (keys[e] == 0xFFFFFFFFu) continue;  // consumed
                        const unsigned key = entry_key(e, alo, ahi);
                        keys[e] = key;
                        lmin = min(lmin, key);
                    }
                }
                kmin = wave_min_u32(lmin);
                if (kmin == 0xFFFFFFFFu) break;
                bound = __uint_as_float(kmin & ~IDMASK);
                if (__builtin_amdgcn_ballot_w64(cull >= bound) == 0ull) break;
            }
            const int s = (int)(kmin & IDMASK);
            const int e = (s - sub) / nsub;
            // the 4 block boxes of this superblock: 24 SGPRs, in flight during the key-list upkeep
            float bx[B16F];
            {
                const float *bp = CB16 + (size_t)s * B16F;  // uniform
#pragma unroll
                for (int i = 0; i < B16F; i++) bx[i] = bp[i];
            }
            if (nmine <= 64) {  // one entry per lane (C2: 32 superblocks, or a quarter of 256)
                kk[0] = lane == e ? 0xFFFFFFFFu : kk[0];
                lmin = kk[0];
            } else if (inreg) {  // (a two-register path for 65..128 entries measured slower for every shape: one more branch level)
                const bool mine = lane == (e & 63);
#pragma unroll
                for (int i = 0; i < KK; i++) kk[i] = (mine && (e >> 6) == i) ? 0xFFFFFFFFu : kk[i];
                lmin = kk_min();  // (guarding the unused registers by nmine measured slower: 158 vs 153 us at 16384^2)
            } else if (lane == (e & 63)) {
                keys[e] = 0xFFFFFFFFu;
                lmin = 0xFFFFFFFFu;
                // 4 entries per trip, all four LDS reads in flight together (clamped indices
                // re-read the consumed entry: 0xFFFFFFFF, neutral)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                for (int ee = lane; ee < nmine; ee += 256) {
                    const unsigned k0 = keys[ee];
                    const unsigned k1 = keys[ee + 64 < nmine ? ee + 64 : e];
                    const unsigned k2 = keys[ee + 128 < nmine ? ee + 128 : e];
                    const unsigned k3 = keys[ee + 192 < nmine ? ee + 192 : e];
                    lmin = min(min(lmin, k0), min(min(k1, k2), k3));
                }
            }
            n_step++;
            unsigned need = 0;
#pragma unroll
            for (int j = 0; j < SBB; j++) {
                const float lb = box_bound(qx, qy, qz, bx[j * 6 + 0], bx[j * 6 + 1], bx[j * 6 + 2], bx[j * 6 + 3],
                                           bx[j * 6 + 4], bx[j * 6 + 5]);
                if (__builtin_amdgcn_ballot_w64(lb <= cull) != 0ull) need |= 1u << j;
            }
            stamp(TRACK ? 4 : 1);
            if (need == 0) continue;
            if constexpr (TRACK) {
                while (need) {
                    const int j = __builtin_ctz(need);
                    need &= need - 1;
                    n_scan++;
                    const float *cp = C + (size_t)(s * SBB + j) * BS * 3;  // uniform: scalar loads
                    const int *ob = Co + (size_t)(s * SBB + j) * BS;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        float r[24];
                        unsigned io[8];
#pragma unroll
                        for (int i = 0; i < 24; i++) r[i] = cp[h * 24 + i];
#pragma unroll
                        for (int i = 0; i < 8; i++) io[i] = (unsigned)ob[h * 8 + i];  // padding carries 0xFFFFFFFF
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const float d = rf::d2_fma(r[u * 3 + 0] - qx, r[u * 3 + 1] - qy, r[u * 3 + 2] - qz);
                            if constexpr (GRAD) {
                                const bool better = d == cull && io[u] < besti2;
                                wpos2 = better ? (s * SBB + j) * BS + h * 8 + u : wpos2;
                            }
                            besti2 = min(besti2, d == cull ? io[u] : 0xFFFFFFFFu);
                        }
                    }
                }
                continue;
            }
            // Surviving blocks: 16 records = two halves of 8 (24 SGPRs each), ping-pong: while one
            // half is being evaluated the next one -- of this block or of the next surviving block --
            // is in flight.  Scalar loads return out of order, so every wait is lgkmcnt(0), placed
            // BEFORE the next issue (as in nn_sweep_kernel).
            float ra[24], rb[24];
            int j = __builtin_ctz(need);
            need &= need - 1;
            __builtin_amdgcn_s_setprio(0);
            {
                // the first surviving block: both halves in one batch (one exposed latency, not two)
                const float *cp = C + (size_t)(s * SBB + j) * BS * 3;
#pragma unroll
                for (int i = 0; i < 24; i++) ra[i] = cp[i];
#pragma unroll
                for (int i = 0; i < 24; i++) rb[i] = cp[24 + i];
            }
            bool have_rb = true;
            for (;;) {
                const int blk = s * SBB + j;
                n_scan++;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                if (!have_rb) {
                    const float *cp = C + (size_t)blk * BS * 3 + 24;
#pragma unroll
                    for (int i = 0; i < 24; i++) rb[i] = cp[i];
                }
                have_rb = false;
                __builtin_amdgcn_sched_barrier(0);
                float cm = scan8(ra, qx, qy, qz, INFINITY);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
                const bool more = need != 0;
                if (more) {
                    j = __builtin_ctz(need);
                    need &= need - 1;
                    const float *cp = C + (size_t)(s * SBB + j) * BS * 3;
#pragma unroll
                    for (int i = 0; i < 24; i++) ra[i] = cp[i];
                }
                __builtin_amdgcn_sched_barrier(0);
                cm = scan8(rb, qx, qy, qz, cm);
                if (cm < best) {
                    best = cm;
                    bblk = blk;
                    bblk2 = -1;
                    tie = false;
                } else if (cm == best) {
                    // a SECOND block with the same minimum is remembered and simply re-scanned too (copies of one point
                    // -- data_util.resample_pcd fills short scans with duplicates -- are neighbours in the sorted
                    // order and straddle at most a block boundary); only a third one sends the lane to the second traversal
                    if (bblk2 < 0) bblk2 = blk;
                    else tie = true;
                }
                if (!more) break;
            }
            cull = fminf(cull, best);
            if (shared4) atomicMin(&shbest[lane], __float_as_int(cull));
            stamp(2);
        }
    };

    unsigned besti = 0xFFFFFFFFu;
    int wpos = -1;  // (GRAD) sorted position of the winner in the candidate set
    traverse(std::false_type{}, glo, ghi, besti, wpos);
    __builtin_amdgcn_s_setprio(0);
    stamp(1);

    // lowest original index among the exact matches of the winning block (and of the second block that equalled it).
    // Per lane gathers, 8 records at a time: 6 + 2 sixteen-byte loads issued TOGETHER, then the arithmetic.  (Rounds 1-3 wrote
    // this as a loop over the records with the index load inside `if (d == best)`: the compiler kept that load
    // conditional -- it may not speculate it -- and the 16 records became 32 dependent memory round trips, 74 % of a
    // one-wave group's lifetime at C2 by s_memtime stamps, profiles/r04_rescan.txt.)
    auto rescan = [&](int wb) {
        constexpr int RB = shared4 ? 4 : 8;  // records per batch (the shared-group form holds more state: 8 spill there)
        const float4 *cp4 = (const float4 *)(C + (size_t)wb * BS * 3);  // per lane; a block is 192 aligned bytes
        const int4 *co4 = (const int4 *)(Co + (size_t)wb * BS);
#pragma unroll 1  // (all 16 records in flight at once need 64 registers: spills at the kernel's 72)
        for (int h = 0; h < BS / RB; h++) {
            float4 r4[RB * 3 / 4];
            int4 o4[RB / 4];
#pragma unroll
            for (int i = 0; i < RB * 3 / 4; i++) r4[i] = cp4[h * (RB * 3 / 4) + i];
#pragma unroll
            for (int i = 0; i < RB / 4; i++) o4[i] = co4[h * (RB / 4) + i];
            float r[RB * 3];
            int o[RB];
#pragma unroll
            for (int i = 0; i < RB * 3 / 4; i++) {
                r[i * 4 + 0] = r4[i].x;
                r[i * 4 + 1] = r4[i].y;
                r[i * 4 + 2] = r4[i].z;
                r[i * 4 + 3] = r4[i].w;
            }
#pragma unroll
            for (int i = 0; i < RB / 4; i++) {
                o[i * 4 + 0] = o4[i].x;
                o[i * 4 + 1] = o4[i].y;
                o[i * 4 + 2] = o4[i].z;
                o[i * 4 + 3] = o4[i].w;
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const float d = rf::d2_fma(r[u * 3 + 0] - qx, r[u * 3 + 1] - qy, r[u * 3 + 2] - qz);
                const bool m = d == best;
                if constexpr (GRAD) {
                    const bool better = m && (unsigned)o[u] < besti;
                    wpos = better ? wb * BS + h * RB + u : wpos;
                }
                besti = m ? min(besti, (unsigned)o[u]) : besti;  // padding carries 0xFFFFFFFF
            }
        }
    };
    // upstream gradient of this query's distance (GRAD): requested with the re-scan's gathers, used by the emit.  (At the top of
    // the function its address waits for the query's original index, and every load behind it in program order -- the
    // candidate boxes of the keys -- waited with it: one more round trip at the head of every wave.)
    float gq = 0.f;
    if constexpr (GRAD) {
        if (valid) gq = ge.gd[dir][(size_t)bi * a.n[dir] + qorig];
    }
    rescan(bblk);
    if (__ballot(bblk2 >= 0) != 0ull) {
        if (bblk2 >= 0) rescan(bblk2);
    }
    stamp(3);
    // queries whose minimum was attained in more than one visited block: second traversal
    const bool flagged = tie && part;
    if (__ballot(flagged) != 0ull) {
        const float flo[3] = {wave_min_f32(flagged ? qx : INFINITY), wave_min_f32(flagged ? qy : INFINITY),
                              wave_min_f32(flagged ? qz : INFINITY)};
        const float fhi[3] = {wave_max_f32(flagged ? qx : -INFINITY), wave_max_f32(flagged ? qy : -INFINITY),
                              wave_max_f32(flagged ? qz : -INFINITY)};
        cull = flagged ? best : -INFINITY;  // the wave's own minima are final: match against them
        unsigned besti2 = 0xFFFFFFFFu;
        int wpos2 = -1;
        traverse(std::true_type{}, flo, fhi, besti2, wpos2);
        if (flagged) {
            besti = besti2;
            wpos = wpos2;
        }
    }

    stamp(4);
    if (stats && lane == 0) {
        atomicAdd(&stats[dir * 4 + 0], 1ull);
        atomicAdd(&stats[dir * 4 + 1], (unsigned long long)n_step);
        atomicMax(&stats[dir * 4 + 2], (unsigned long long)n_step);
        atomicAdd(&stats[dir * 4 + 3], (unsigned long long)n_scan);
        atomicMax(&stats[8 + dir], (unsigned long long)n_scan);
        atomicAdd(&stats[12 + dir], (unsigned long long)n_scan * (SB * BS));  // directed pairs evaluated (64 queries x 16 candidates per scan)
        stats[14 + dir] = SB * BS;  // pairs per counted scan of the LAST wave to report (a launch may mix both forms: use [12 + dir])
    }

    if (shared4) {
        md[wib][lane] = best;
        mi[wib][lane] = besti;
        if constexpr (GRAD) mp[wib][lane] = wpos;
        __syncthreads();
        if (wib != 0) return;
#pragma unroll
        for (int w = 1; w < NSH; w++) {
            const float d = md[w][lane];
            const unsigned i = mi[w][lane];
            if (d < best || (d == best && i < besti)) {
                best = d;
                besti = i;
                if constexpr (GRAD) wpos = mp[w][lane];
            }
        }
    }
    if (valid) {
        (dir ? dist1 : dist0)[(size_t)bi * a.n[dir] + qorig] = qnan ? NAN : best;
        (dir ? idx1 : idx0)[(size_t)bi * a.n[dir] + qorig] = (qnan || besti == 0xFFFFFFFFu) ? 0 : (int)besti;
    }
    if constexpr (GRAD) {
        // index 0 of the NaN / no-match policy above = the candidate set's point with ORIGINAL index 0
        const int *pos0 = (const int *)((const char *)(cd ? b64_1 : b64_0) +
                                        emit_align((size_t)a.b * (a.npad[cd] / SB) * B64F * sizeof(float)));
        const int p0 = pos0[bi];  // uniform
        const EmitView ev = emit_layout(ge.base, a.b, a.npad[0], a.npad[1], dir);
        const int w = (qnan || besti == 0xFFFFFFFFu) ? p0 : wpos;
        const int wc = valid ? w : 0;
        // (keeping the winner's coordinates from the re-scan instead of this gather: same time, 87.7 vs 88.0 us per step, and a scratch slot)
        const float cx = C[(size_t)wc * 3 + 0], cy = C[(size_t)wc * 3 + 1], cz = C[(size_t)wc * 3 + 2];
        const float g2 = gq + gq;  // the reference's arithmetic: g = gd + gd; (a - b) * g rounded on its own
        const size_t r = (size_t)bi * a.npad[dir] + g * SB + lane;
        ev.rec[r] = make_int4(valid ? w : -1, __float_as_int((qx - cx) * g2), __float_as_int((qy - cy) * g2),
                              __float_as_int((qz - cz) * g2));
        // the buckets this group's winners fall into: one trip per DISTINCT bucket (a handful: the winners of 64
        // consecutive sorted queries are neighbours)
        const int bkt = w / ge.qbucket[cd];
        unsigned long long todo = __builtin_amdgcn_ballot_w64(valid), gm = 0ull;
        while (todo) {
            const int bb = __builtin_amdgcn_readlane(bkt, __builtin_ctzll(todo));
            gm |= 1ull << bb;
            todo &= ~__builtin_amdgcn_ballot_w64(bkt == bb);
        }
        if (lane == 0) ev.mask[(size_t)bi * G + g] = gm;
    }
}

// ------------------------------------------------------------------------------------------
// Round 4: the few-groups direction (a small query set against a large candidate set: C2's 2048 queries in
// 16384) as QUAD-PER-QUERY tiles.  In sweep_group a candidate block is streamed through SGPRs to all 64 lanes
// whenever ONE of them needs it: 22 % of the lane x block evaluations of such a group are needed by the lane that
// runs them (profiles/r03_str_model.txt), and no seed or order changes that (tools/experiments/seed_model.py: an
// oracle seed still scans 55 of 68 blocks per group) -- the unit of 64 queries x 16 candidates is the limit.  Here a
// wave owns ONE 16-record block of the sorted query set; the four lanes of a quad share a query and split a
// candidate block's 16 records (3 x dwordx4 = 48 B gathered per lane: 2.3e12 pairs/s chip-wide from L2,
// tools/ubench/gather_scan.hip -- the suggested DPP row-broadcast operand form costs 1.74x per pair instead,
// tools/ubench/valu_rate.hip mode 19), and every quad keeps its OWN lists, so the 16 quads of a wave scan DIFFERENT
// blocks at the same time: ~50 pairs per query instead of ~1100 (tools/experiments/quad_model.py).
//   1  keys of all candidate superblocks against the tile's box (lanes = superblocks); those that overlap it (or the
//      nearest) are the seed candidates
//   2  seed: per query the nearest seed superblock, its nearest block, one scan -> every query starts with a minimum
//      ~1.3x its final distance, and U = the tile's largest minimum
//   3  the tile's superblock list: key <= U
//   4  per query (lanes = (query, list entry)): superblocks whose box bound <= the query's minimum -> the quad's list
//   5  per quad (lane = one of the superblock's 4 blocks): block bounds -> the quad's block list
//   6  drain: every quad pops its next block whose bound is still <= its minimum and scans it
// Steps 4-6 run in bounded chunks (a list that could overflow is drained first), so any input is handled by the same
// code.  Same outputs as sweep_group: the same d2 sequence on the same pairs' survivors, bounds culled strictly, the
// lowest original index among exact ties (two equal blocks re-scanned, a third one -> exhaustive pass).
#ifndef RFP_T16_CAPB
#define RFP_T16_CAPB 16
#endif
#ifndef RFP_T16_UN6
#define RFP_T16_UN6 2
#endif
constexpr int T16_CAPS = 32;  // entries of a quad's superblock list
constexpr int T16_CAPB = RFP_T16_CAPB;  // entries of a quad's block list (the lists of a 4-tile workgroup + its key lists must stay under
                              // 160 KB / 7: with 32 entries the launch's workgroups -- the other direction's too -- drop to 6 per CU)
constexpr int T16_UN5 = 4;    // list rounds per trip of step 4 (their gathers in flight together)
constexpr int T16_UN6 = RFP_T16_UN6;    // block-test rounds per trip of step 5
constexpr int T16_KK = 5;     // tile keys kept in registers: candidate sets of up to 320 superblocks
struct SharedLds {  // sweep_group<true>: the group's shared minima and the four waves' results
    int shbest[64];
    float md[RFP_NSH][64];
    unsigned mi[RFP_NSH][64];
    int mp[RFP_NSH][64];
};
struct T16Lds {
    unsigned short qsb[16][T16_CAPS];
    alignas(16) float2 qe[16][T16_CAPB];  // (bound, block id): a pair of entries is one 16-byte LDS read
};

#define RFP_QUAD(OP, PERM) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float quad_min_f32(float v) {  // inputs not NaN
    RFP_QUAD("v_min_f32_dpp", "[1,0,3,2]");
    RFP_QUAD("v_min_f32_dpp", "[2,3,0,1]");
    return v;
}
__device__ __forceinline__ unsigned quad_min_u32(unsigned v) {
    RFP_QUAD("v_min_u32_dpp", "[1,0,3,2]");
    RFP_QUAD("v_min_u32_dpp", "[2,3,0,1]");
    return v;
}
__device__ __forceinline__ int quad_max_i32(int v) {
    RFP_QUAD("v_max_i32_dpp", "[1,0,3,2]");
    RFP_QUAD("v_max_i32_dpp", "[2,3,0,1]");
    return v;
}
#undef RFP_QUAD
// the quad's lexicographic minimum of (value, index): every lane ends with the same pair
template <int CTL>
__device__ __forceinline__ void quad_lexmin_step(float &v, int &i) {
    const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTL, 0xf, 0xf, false));
    const int oi = __builtin_amdgcn_update_dpp(0, i, CTL, 0xf, 0xf, false);
    const bool b = (ov < v) | ((ov == v) & (oi < i));
    v = b ? ov : v;
    i = b ? oi : i;
}
__device__ __forceinline__ void quad_lexmin(float &v, int &i) {
    quad_lexmin_step<0xB1>(v, i);  // quad_perm [1,0,3,2]
    quad_lexmin_step<0x4E>(v, i);  // quad_perm [2,3,0,1]
}
// LDS written by some lanes of a wave and read by others: the wave's LDS operations execute in order, the fence
// only keeps the compiler from moving them across
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <bool GRAD>
__device__ __forceinline__ void sweep_tile16(
    const SweepArgs &a, const GradEmit &ge, const int dir, const int gid, const int wib, const int lane,
    unsigned *__restrict__ keys_dyn, T16Lds *__restrict__ tl, unsigned long long *gmsh, unsigned *gmcnt,
    const float *__restrict__ xyz0, const float *__restrict__ xyz1,
    const int *__restrict__ orig0, const int *__restrict__ orig1, const float *__restrict__ b16_0,
    const float *__restrict__ b16_1, const float *__restrict__ b64_0, const float *__restrict__ b64_1,
    float *__restrict__ dist0, float *__restrict__ dist1, int *__restrict__ idx0, int *__restrict__ idx1,
    unsigned long long *__restrict__ stats) {
    const int cd = 1 - dir;
    const int G = a.groups[dir];
    const int bi = gid / G, g = gid - bi * G;
    const int qi = lane >> 2, k = lane & 3;  // query of the tile, quarter of a block

    const float *__restrict__ Q = (dir ? xyz1 : xyz0) + (size_t)bi * a.npad[dir] * 3;
    const int *__restrict__ Qo = (dir ? orig1 : orig0) + (size_t)bi * a.npad[dir];
    const float *__restrict__ C = (dir ? xyz0 : xyz1) + (size_t)bi * a.npad[cd] * 3;
    const int *__restrict__ Co = (dir ? orig0 : orig1) + (size_t)bi * a.npad[cd];
    const int nsb = a.npad[cd] / SB;
    const float *__restrict__ CB16 = (dir ? b16_0 : b16_1) + (size_t)bi * nsb * B16F;
    const float *__restrict__ CB64 = (dir ? b64_0 : b64_1) + (size_t)bi * nsb * B64F;

    const int qpos = (g * SBB + wib) * BS + qi;  // sorted position of this quad's query
    const float qx = Q[(size_t)qpos * 3 + 0], qy = Q[(size_t)qpos * 3 + 1], qz = Q[(size_t)qpos * 3 + 2];
    const int qorig = Qo[qpos];
    const bool valid = qorig >= 0;
    const bool qnan = qx != qx || qy != qy || qz != qz;  // (NaN, 0), as sweep_group
    const bool part = valid && !qnan;
    // the tile's box = the query set's own block box (padding and NaN excluded by the sort)
    const float *tb = (dir ? b16_1 : b16_0) + ((size_t)bi * G + g) * B16F + wib * 6;  // uniform
    const float tlo[3] = {tb[0], tb[1], tb[2]}, thi[3] = {tb[3], tb[4], tb[5]};

    unsigned *__restrict__ sbl = keys_dyn + (size_t)wib * a.kstride;  // this wave's superblock list (>= nsb entries)
    // the quad's running minimum, the block that first attained it, the second block that equalled it, and a flag for
    // a third (sweep_group's tie scheme: the index is resolved at the end by re-scanning one or two blocks.  Carrying
    // (index, position) through every scan instead -- a fourth gather and a lexicographic quad reduction per block, no
    // re-scan -- was built, is bit-exact, and measured SLOWER: 59.8 vs 54.3 us for the C2 sweep, profiles/r04_tile16.txt)
    float best = INFINITY, cull = part ? INFINITY : -INFINITY;
    int bblk = 0, bblk2 = -1, seedblk = -1;
    bool tie = false;
    unsigned n_round = 0, n_scan = 0;
    auto stamp = [&](int) {};

    // (minimum, lowest original index, position) of the quad's query over the 16 records of block blk: each lane its
    // 4 records, then the quad's lexicographic minimum -- every lane of the quad ends with the same triple
    struct BlockRegs {  // a lane's quarter of a candidate block: 4 records
        float4 r0, r1, r2;
    };
    auto block_load = [&](int blk) {
        const float4 *p = (const float4 *)(C + (size_t)blk * (BS * 3) + k * 12);
        BlockRegs g;
        g.r0 = p[0];
        g.r1 = p[1];
        g.r2 = p[2];
        return g;
    };
    // min d2 of the quad's query over the 16 records of a block: each lane its 4, then the quad's minimum
    auto block_min = [&](const BlockRegs &g) {
        const float4 r0 = g.r0, r1 = g.r1, r2 = g.r2;
        const float d0 = rf::d2_fma(r0.x - qx, r0.y - qy, r0.z - qz), d1 = rf::d2_fma(r0.w - qx, r1.x - qy, r1.y - qz);
        const float d2 = rf::d2_fma(r1.z - qx, r1.w - qy, r2.x - qz), d3 = rf::d2_fma(r2.y - qx, r2.z - qy, r2.w - qz);
        float cm = min3_acc(INFINITY, d0, d1);
        cm = min3_acc(cm, d2, d3);
        return quad_min_f32(cm);
    };
    auto take = [&](bool on, float cm, int blk) {
        const bool lt = on & (cm < best), eq = on & (cm == best);
        tie = lt ? false : (tie | (eq & (bblk2 >= 0)));
        bblk2 = lt ? -1 : ((eq & (bblk2 < 0)) ? blk : bblk2);
        bblk = lt ? blk : bblk;
        best = lt ? cm : best;
        cull = fminf(cull, best);
    };
    auto sb_bound = [&](int s) {  // per lane: query against the box of superblock s
        const float4 *cb = (const float4 *)(CB64 + (size_t)s * B64F);
        const float4 lo = cb[0], hi = cb[1];
        return box_bound(qx, qy, qz, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
    };
    auto blk_bound = [&](int blk) {
        const float2 *bp = (const float2 *)(CB16 + (size_t)blk * 6);
        const float2 v0 = bp[0], v1 = bp[1], v2 = bp[2];
        return box_bound(qx, qy, qz, v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
    };
    auto tile_key = [&](int s) {  // lanes = superblocks: box-to-box bound against the tile
        const float4 *cb = (const float4 *)(CB64 + (size_t)s * B64F);
        return boxbox_bound(tlo, thi, cb[0], cb[1]);
    };
    auto nibble = [&](bool pred) { return (unsigned)(__builtin_amdgcn_ballot_w64(pred) >> (lane & ~3)) & 0xFu; };

    // (up to 320 superblocks -- every cloud of the register-resident sort -- the keys stay in registers for step 3 and their
    // loads are all in flight together, AHEAD of the branch on `part`: behind it they were issued only once the query's own
    // loads had come back)
    const bool inreg = nsb <= 64 * T16_KK;  // uniform
    float kk[T16_KK];
    if (inreg) {
#pragma unroll
        for (int c = 0; c < T16_KK; c++) {
            // (clamped index, select afterwards: a load under `c * 64 + lane < nsb` is a branch with its own wait per round -- five
            // dependent round trips for a 16384-point candidate cloud, a quarter of a tile wave's life by the stamps)
            const float key = tile_key(min(c * 64 + lane, nsb - 1));
            kk[c] = c * 64 + lane < nsb ? key : INFINITY;
        }
    }
    if (__builtin_amdgcn_ballot_w64(part) != 0ull) {
        // 1. superblocks that overlap the tile box -> sbl[0 .. n0), at most 64 (they only seed); else the nearest
        int n0 = 0;
        unsigned kmin = 0xFFFFFFFFu;
        auto seed_cand = [&](int s, float lb) {
            const bool z = lb == 0.f;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(z);
            const int pos = n0 + lanes_below(m);
            if (z && pos < 64) sbl[pos] = (unsigned)s;
            n0 = min(64, n0 + __builtin_popcountll(m));
            if (s < nsb) kmin = min(kmin, (__float_as_uint(lb) & ~IDMASK) | (unsigned)s);
        };
        if (inreg) {
#pragma unroll
            for (int c = 0; c < T16_KK; c++)
                if (c * 64 < nsb) seed_cand(c * 64 + lane, kk[c]);
        } else {
            for (int s0 = 0; s0 < nsb; s0 += 64) seed_cand(s0 + lane, s0 + lane < nsb ? tile_key(s0 + lane) : INFINITY);
        }
        if (n0 == 0) {  // uniform
            kmin = wave_min_u32(kmin);
            if (lane == 0) sbl[0] = kmin & IDMASK;
            n0 = 1;
        }
        wave_lds_order();
        stamp(0);
        // 2. seed: per query the NEAREST BLOCK among the blocks of the seed superblocks (lane k of the quad takes every
        // 4th of them; a superblock's four block boxes are 96 contiguous bytes), then one scan of it
        {
            float blb = INFINITY;
            int bblk = (int)sbl[0] * SBB;
            for (int e = k; e < n0; e += 4) {
                const int s = (int)sbl[e];
                const float4 *bp = (const float4 *)(CB16 + (size_t)s * B16F);
                const float4 f0 = bp[0], f1 = bp[1], f2 = bp[2], f3 = bp[3], f4 = bp[4], f5 = bp[5];
                const float f[B16F] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w, f2.x, f2.y, f2.z, f2.w,
                                       f3.x, f3.y, f3.z, f3.w, f4.x, f4.y, f4.z, f4.w, f5.x, f5.y, f5.z, f5.w};
#pragma unroll
                for (int j = 0; j < SBB; j++) {
                    const float lb = box_bound(qx, qy, qz, f[j * 6 + 0], f[j * 6 + 1], f[j * 6 + 2], f[j * 6 + 3], f[j * 6 + 4], f[j * 6 + 5]);
                    const bool nb = lb < blb;  // (NaN query: never; it takes no part anyway)
                    blb = nb ? lb : blb;
                    bblk = nb ? s * SBB + j : bblk;
                }
            }
            quad_lexmin(blb, bblk);
            seedblk = bblk;
            take(part, block_min(block_load(seedblk)), seedblk);
            n_scan += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(part)) >> 2;
        }
        stamp(1);
        // 3. the tile's superblock list: box-to-box bound <= the largest minimum of the tile's queries
        const float U = wave_max_nonneg(part ? best : -INFINITY);
        int nl = 0;
        auto list_cand = [&](int s, float lb) {
            const bool keep = s < nsb && lb <= U;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) sbl[nl + lanes_below(m)] = (unsigned)s;
            nl += __builtin_popcountll(m);
        };
        if (inreg) {
#pragma unroll
            for (int c = 0; c < T16_KK; c++)
                if (c * 64 < nsb) list_cand(c * 64 + lane, kk[c]);
        } else {
            for (int s0 = 0; s0 < nsb; s0 += 64) list_cand(s0 + lane, s0 + lane < nsb ? tile_key(s0 + lane) : INFINITY);
        }
        wave_lds_order();
        stamp(2);
        // 4-6. bounded chunks
        int cntS = 0, cntB = 0;  // quad-uniform fill of the quad's two lists
        int r5 = 0;
        const int n5 = (nl + 3) >> 2;
        for (;;) {
            while (r5 < n5 && __builtin_amdgcn_ballot_w64(cntS > T16_CAPS - 4 * T16_UN5) == 0ull) {
                int sv[T16_UN5];
                float lbv[T16_UN5];
#pragma unroll
                for (int u = 0; u < T16_UN5; u++) {
                    const int e = (r5 + u) * 4 + k;
                    sv[u] = e < nl ? (int)sbl[e] : -1;
                }
#pragma unroll
                for (int u = 0; u < T16_UN5; u++) lbv[u] = sb_bound(sv[u] < 0 ? 0 : sv[u]);
#pragma unroll
                for (int u = 0; u < T16_UN5; u++) {
                    const bool pass = sv[u] >= 0 && lbv[u] <= cull;
                    const unsigned nib = nibble(pass);
                    if (pass) tl->qsb[qi][cntS + __builtin_popcount(nib & ((1u << k) - 1u))] = (unsigned short)sv[u];
                    cntS += __builtin_popcount(nib);
                }
                r5 += T16_UN5;
                n_round++;
            }
            wave_lds_order();
            stamp(3);
            int r6 = 0;
            while (__builtin_amdgcn_ballot_w64(r6 < cntS) != 0ull) {
                while (__builtin_amdgcn_ballot_w64(r6 < cntS) != 0ull &&
                       __builtin_amdgcn_ballot_w64(cntB > T16_CAPB - 4 * T16_UN6) == 0ull) {
                    int bv[T16_UN6];
                    float lbv[T16_UN6];
#pragma unroll
                    for (int u = 0; u < T16_UN6; u++) bv[u] = r6 + u < cntS ? (int)tl->qsb[qi][r6 + u] * SBB + k : -1;
#pragma unroll
                    for (int u = 0; u < T16_UN6; u++) lbv[u] = blk_bound(bv[u] < 0 ? 0 : bv[u]);
#pragma unroll
                    for (int u = 0; u < T16_UN6; u++) {
                        const bool pass = bv[u] >= 0 && lbv[u] <= cull && bv[u] != seedblk;
                        const unsigned nib = nibble(pass);
                        if (pass) {
                            const int pos = cntB + __builtin_popcount(nib & ((1u << k) - 1u));
                            tl->qe[qi][pos] = make_float2(lbv[u], __int_as_float(bv[u]));
                        }
                        cntB += __builtin_popcount(nib);
                    }
                    r6 += T16_UN6;
                    n_round++;
                }
                wave_lds_order();
                stamp(4);
                // two list entries per round, their gathers in flight together (the second one's bound may have been
                // overtaken by the first one's minimum: its block minimum then exceeds the new minimum and changes nothing)
                int ptr = 0;
                for (;;) {
                    bool v0, v1;
                    float4 e01;
                    for (;;) {  // every quad moves on to its next pair with a block whose bound has not been overtaken
                        const bool more = ptr < cntB;
                        e01 = *(const float4 *)&tl->qe[qi][ptr < T16_CAPB ? ptr : 0];  // (ptr is even: entries ptr, ptr + 1)
                        v0 = more && e01.x <= cull;
                        v1 = ptr + 1 < cntB && e01.z <= cull;
                        const bool skip = more && !v0 && !v1;
                        if (skip) ptr += 2;
                        if (__builtin_amdgcn_ballot_w64(skip) == 0ull) break;
                    }
                    if (__builtin_amdgcn_ballot_w64(v0 || v1) == 0ull) break;
                    const int b0 = v0 ? __float_as_int(e01.y) : seedblk, b1 = v1 ? __float_as_int(e01.w) : seedblk;
                    const BlockRegs g0 = block_load(b0), g1 = block_load(b1);  // (both gathers in flight before either is used)
                    const float c0 = block_min(g0), c1 = block_min(g1);
                    take(v0, c0, b0);
                    take(v1, c1, b1);
                    if (v0 || v1) ptr += 2;
                    n_scan += (unsigned)(__builtin_popcountll(__builtin_amdgcn_ballot_w64(v0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(v1))) >> 2;
                }
                cntB = 0;
                wave_lds_order();
                stamp(5);
            }
            cntS = 0;
            if (r5 >= n5) break;
        }
    }

    // lowest original index among the exact matches of the winning block (and of the second block that equalled it);
    // each lane its quarter, then the quad's minimum
    unsigned besti = 0xFFFFFFFFu;
    int wpos = -1;
    auto rescan = [&](int wb, unsigned &bi_, int &wp_) {
        const float4 *p = (const float4 *)(C + (size_t)wb * (BS * 3) + k * 12);
        const int4 co = *(const int4 *)(Co + (size_t)wb * BS + k * 4);
        const float4 r0 = p[0], r1 = p[1], r2 = p[2];
        const float d[4] = {rf::d2_fma(r0.x - qx, r0.y - qy, r0.z - qz), rf::d2_fma(r0.w - qx, r1.x - qy, r1.y - qz),
                            rf::d2_fma(r1.z - qx, r1.w - qy, r2.x - qz), rf::d2_fma(r2.y - qx, r2.z - qy, r2.w - qz)};
        const unsigned io[4] = {(unsigned)co.x, (unsigned)co.y, (unsigned)co.z, (unsigned)co.w};  // padding carries 0xFFFFFFFF
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool b = (d[u] == best) & (io[u] < bi_);
            bi_ = b ? io[u] : bi_;
            wp_ = b ? wb * BS + k * 4 + u : wp_;
        }
    };
    float gq = 0.f;  // (GRAD) upstream gradient of the query's distance: requested with the re-scan's gathers (see sweep_group)
    if constexpr (GRAD) {
        if (valid) gq = ge.gd[dir][(size_t)bi * a.n[dir] + qorig];
    }
    if (part) rescan(bblk, besti, wpos);
    if (__builtin_amdgcn_ballot_w64(part && bblk2 >= 0) != 0ull) {
        if (part && bblk2 >= 0) rescan(bblk2, besti, wpos);
    }
    // a third block equalled the minimum: every block that can hold a match, exhaustively (rare: points repeated
    // three times across blocks, symmetric configurations)
    const bool flagged = tie && part;
    if (__builtin_amdgcn_ballot_w64(flagged) != 0ull) {
        unsigned besti2 = 0xFFFFFFFFu;
        int wpos2 = -1;
        for (int s = 0; s < nsb; s++) {
            const float *cb = CB64 + (size_t)s * B64F;  // uniform: scalar loads
            const float lb = box_bound(qx, qy, qz, cb[0], cb[1], cb[2], cb[4], cb[5], cb[6]);
            const bool in = flagged && lb <= best;
            if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;
            const bool needk = in && blk_bound(s * SBB + k) <= best;
            const unsigned nib = nibble(needk);
#pragma unroll
            for (int j = 0; j < SBB; j++) {
                const bool nj = (nib >> j) & 1u;
                if (__builtin_amdgcn_ballot_w64(nj) == 0ull) continue;
                if (nj) rescan(s * SBB + j, besti2, wpos2);
            }
        }
        if (flagged) {
            besti = besti2;
            wpos = wpos2;
        }
    }
    {
        const unsigned mi = quad_min_u32(besti);
        wpos = quad_max_i32(besti == mi ? wpos : -1);
        besti = mi;
    }

    stamp(6);
    if (stats && lane == 0) {
        atomicAdd(&stats[dir * 4 + 0], 1ull);
        atomicAdd(&stats[dir * 4 + 1], (unsigned long long)n_round);
        atomicMax(&stats[dir * 4 + 2], (unsigned long long)n_round);
        atomicAdd(&stats[dir * 4 + 3], (unsigned long long)n_scan);
        atomicMax(&stats[8 + dir], (unsigned long long)n_scan);
        atomicAdd(&stats[12 + dir], (unsigned long long)n_scan * BS);  // directed pairs evaluated (one query x 16 candidates per scan)
        stats[14 + dir] = BS;  // pairs per counted scan of the LAST wave to report (sweep_group: 64 x 16)
    }

    if (valid && k == 0) {
        (dir ? dist1 : dist0)[(size_t)bi * a.n[dir] + qorig] = qnan ? NAN : best;
        (dir ? idx1 : idx0)[(size_t)bi * a.n[dir] + qorig] = (qnan || besti == 0xFFFFFFFFu) ? 0 : (int)besti;
    }
    if constexpr (GRAD) {
        const int *pos0 = (const int *)((const char *)(cd ? b64_1 : b64_0) +
                                        emit_align((size_t)a.b * (a.npad[cd] / SB) * B64F * sizeof(float)));
        const int p0 = pos0[bi];  // uniform
        const EmitView ev = emit_layout(ge.base, a.b, a.npad[0], a.npad[1], dir);
        const int w = (qnan || besti == 0xFFFFFFFFu) ? p0 : wpos;
        const int wc = valid ? w : 0;
        const float cx = C[(size_t)wc * 3 + 0], cy = C[(size_t)wc * 3 + 1], cz = C[(size_t)wc * 3 + 2];
        const float g2 = gq + gq;
        const size_t r = (size_t)bi * a.npad[dir] + qpos;
        if (k == 0) {
            ev.rec[r] = make_int4(valid ? w : -1, __float_as_int((qx - cx) * g2), __float_as_int((qy - cy) * g2),
                                  __float_as_int((qz - cz) * g2));
        }
        const int bkt = w / ge.qbucket[cd];
        unsigned long long todo = __builtin_amdgcn_ballot_w64(valid && k == 0), gm = 0ull;
        while (todo) {
            const int bb = __builtin_amdgcn_readlane(bkt, __builtin_ctzll(todo));
            gm |= 1ull << bb;
            todo &= ~__builtin_amdgcn_ballot_w64(bkt == bb);
        }
        // the mask is per 64-query GROUP: the workgroup's four tiles.  No barrier: a tile wave that is done leaves, its wave slot
        // with it (the slowest of four tiles takes 20 us against 15.5 on average; measured against the barrier form: sweep 48.1
        // vs 48.3 us, inside the noise).  Every wave stores its mask and counts itself in with a workgroup-scope acq_rel
        // increment; the LAST one writes the group's mask.
        if (lane == 0) {
            gmsh[wib] = gm;
            if (__hip_atomic_fetch_add(gmcnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)NSH - 1u)
                ev.mask[(size_t)bi * G + g] = gmsh[0] | gmsh[1] | gmsh[2] | gmsh[3];
        }
    }
}

// Direction 0's workgroups come first in the grid (the `a.wg0` first ones), then direction 1's.
// A direction whose groups are shared by 4 waves (few, heavy groups) uses one 256-thread workgroup
// per group.  The others use one WAVE per group: as one-wave workgroups when the launch holds
// nothing else (the dispatcher then refills every wave slot the moment it frees up), packed 4 to
// a 256-thread workgroup when the launch also holds shared groups (two launches, one per shape,
// measured slower than the packing).  A persistent grid pulling group ids from an atomic counter
// was tried as well: 3x slower -- same-address device-scope atomics serialise at ~25 ns each.
// (7 waves per SIMD: the loop's 48 record + 24 box SGPRs put the kernel at 106 SGPRs = 6 waves; capping
// it at 7 spills 16 cold ones to VGPR lanes and measures 2 % faster, capping at 8 spills into the loop)
template <bool GRAD>
__global__ __launch_bounds__(64 * NSH) __attribute__((amdgpu_waves_per_eu(RFP_WPE, RFP_WPE))) void nnp_sweep_kernel(
    SweepArgs a, GradEmit ge, const float *__restrict__ xyz0, const float *__restrict__ xyz1, const int *__restrict__ orig0,
    const int *__restrict__ orig1, const float *__restrict__ b16_0, const float *__restrict__ b16_1,
    const float *__restrict__ b64_0, const float *__restrict__ b64_1, float *__restrict__ dist0,
    float *__restrict__ dist1, int *__restrict__ idx0, int *__restrict__ idx1,
    unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys_dyn[];  // [waves][kstride]: each wave's list of superblock keys, then SharedLds | T16Lds[NSH]
    __shared__ unsigned long long gmsh[NSH];
    __shared__ unsigned gmcnt;  // (step) tile waves of the workgroup that have stored their bucket mask
    // The shared-group form's exchange arrays live in DYNAMIC LDS too, over the tiles' lists (a workgroup is one or the other):
    // as static arrays their 3.3 KB were charged to every workgroup, 20.3 KB in all = 7 workgroups per CU = exactly the 28 wave
    // slots -- and a workgroup's LDS is held until its LAST wave ends (18 us for the slowest of four against 15 on average).
    // With 17 KB, 9 fit: the wave slots bind, and a new workgroup starts as soon as any four are free.
    SharedLds *shl = (SharedLds *)(keys_dyn + (size_t)(blockDim.x >> 6) * a.kstride);
    int *shbest = shl->shbest;
    float(*md)[64] = shl->md;
    unsigned(*mi)[64] = shl->mi;
    int(*mp)[64] = shl->mp;

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt to the 8 XCDs round-robin, and each XCD has its own 4 MB L2.  The logical
    // order is cloud-major (per batch element: direction 0's workgroups, then direction 1's), and
    // the swizzle hands every XCD a CONTIGUOUS eighth of it: an XCD then works on ~b/8 clouds
    // (a few hundred KB each) instead of streaming all of them through its L2 (measured before:
    // 145 MB of L2 misses per launch for 12 MB of clouds).
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    const int wpc = a.wg0 + a.wg1;  // workgroups per batch element
    const int bi = logical / wpc;
    int wg = logical - bi * wpc;
    // (other grid orders -- direction 1 first, the two interleaved per cloud, per XCD all of direction 0's workgroups first -- all measured
    // slower: tools/experiments/sweep_grid_orders.patch.txt)
    const int dir = wg >= a.wg0;
    if (dir) wg -= a.wg0;
    if (a.nw[dir] == NSH) {
#if RFP_TILE16
        // the candidate cloud's crowded flag (written by the sort behind pos0): uniform per workgroup
        const int cdk = 1 - dir;
        const int *flags = (const int *)((const char *)(cdk ? b64_1 : b64_0) +
                                         emit_align((size_t)a.b * (a.npad[cdk] / SB) * B64F * sizeof(float))) + a.b;
        if (flags[bi]) {
            sweep_group<true, GRAD>(a, ge, dir, bi * a.groups[dir] + wg, wib, lane, keys_dyn, shbest, md, mi, mp, xyz0, xyz1,
                                    orig0, orig1, b16_0, b16_1, b64_0, b64_1, dist0, dist1, idx0, idx1, stats);
            return;
        }
        // (the tiles' lists live in DYNAMIC LDS behind the key lists: static arrays would be charged to every workgroup of
        // the kernel, and the one-wave workgroups of a launch without shared groups lose a third of their residency)
        T16Lds *t16 = (T16Lds *)(keys_dyn + (size_t)NSH * a.kstride);
        if constexpr (GRAD) {  // (all four waves are here: the barrier costs nothing at the head of the workgroup)
            if (threadIdx.x == 0) gmcnt = 0;
            __syncthreads();
        }
        sweep_tile16<GRAD>(a, ge, dir, bi * a.groups[dir] + wg, wib, lane, keys_dyn, &t16[wib], gmsh, &gmcnt, xyz0, xyz1, orig0, orig1,
                           b16_0, b16_1, b64_0, b64_1, dist0, dist1, idx0, idx1, stats);
#else
        sweep_group<true, GRAD>(a, ge, dir, bi * a.groups[dir] + wg, wib, lane, keys_dyn, shbest, md, mi, mp, xyz0, xyz1,
                                orig0, orig1, b16_0, b16_1, b64_0, b64_1, dist0, dist1, idx0, idx1, stats);
#endif
    } else {
        const int g = wg * (int)(blockDim.x >> 6) + wib;
        if (g >= a.groups[dir]) return;  // (no barriers on this path)
        sweep_group<false, GRAD>(a, ge, dir, bi * a.groups[dir] + g, wib, lane, keys_dyn, shbest, md, mi, mp, xyz0, xyz1,
                                 orig0, orig1, b16_0, b16_1, b64_0, b64_1, dist0, dist1, idx0, idx1, stats);
    }
}

// ------------------------------------------------------------------------------------------
// Backward of rf_chamfer_step in SORTED index space.  Same arithmetic as nn_grad_kernel (nn_distance.hip):
//   grad_D[j] = (x_j - y_{w(j)}) * 2 gd_D[j]  -  sum_{k in S: w(k) = j} (y_k - x_j) * 2 gd_S[k]
//             = own_D[j] - sum_{k in S: w(k) = j} own_S[k]
// with every own term already formed by the sweep.  A workgroup owns a tile of consecutive SORTED destination
// positions.  Nearest neighbours of consecutive sorted sources are spatially clustered, so the 64-source groups
// whose winners can fall into a tile are few (the sweep's bucket mask per group says which): a tile reads the
// masks of all groups (8 bytes per 64 sources), lists the matching ones and visits only those, one wave per
// group, several groups in flight: position and own term of a source are 16 bytes of coalesced loads.  The
// original-order kernel reads every source index in every tile's workgroup and issues seven loads per source for
// it (22 us at C2: 0.14 of its HBM roof).  Scatter terms meet in LDS (ds_add_f32: ~30 cycles per wave
// instruction on the CU's one LDS pipe -- what bounds this kernel, hence tiles as fat as the grid allows), the
// tile leaves through its original indices.
#ifndef RFP_GS_TPB
#define RFP_GS_TPB 256
#endif
#ifndef RFP_GS_F64
#define RFP_GS_F64 1  // the tile's sums in DOUBLE: ds_add_f64 runs at 18 lane-operations per ns and CU, ds_add_f32 at 0.8 (tools/ubench/lds_atomic_rate.hip)
#endif
#ifndef RFP_GS_SEG
#define RFP_GS_SEG 1  // runs of equal destination inside a 16-lane row summed in registers before they touch LDS (grad_tile)
#endif
#ifndef RFP_GS_KB
#define RFP_GS_KB 2  // (4 while the fp32 atomics were the bound: 12.6 vs 12.3 us now)
#endif
#ifndef RFP_GS_WG
#define RFP_GS_WG 1024  // workgroups per set aimed at
#endif
constexpr int GS_TPB = RFP_GS_TPB;
constexpr int GS_GT = 1280;             // destination records per tile at most (>= one bucket of the largest cloud: 1025)
constexpr int GS_MAXG = rfp::kMaxPoints / SB + 4;
constexpr int GS_KB = RFP_GS_KB;        // groups in flight per wave
struct GradSArgs {
    int b;
    int n[2], npad[2], gt[2], tiles[2];
    const int *orig[2];    // (b, npad)
    const int4 *rec[2];    // (b, npad) {winner's sorted position in the other set, own term x, y, z}
    const unsigned long long *mask[2];  // (b, npad / 64) buckets of the OTHER set the group's winners fall into
    int qbucket[2];        // positions per bucket of set d; gt[d] is a multiple of it
    float *grad[2];        // (b, n, 3) original order
};

// One destination tile (256 threads): `tile` in [0, tiles[0] + tiles[1]) of cloud bi.  acc: gt * 3 floats of LDS; list:
// GS_MAXG entries; nlist_p: one LDS word.  (A device function since round 4's fused-step experiment ran the same tile
// inside the sweep's launch: tools/experiments/fused_step.patch.txt)
#if RFP_GS_F64
typedef double gs_acc_t;
#else
typedef float gs_acc_t;
#endif
__device__ __forceinline__ void grad_tile(const GradSArgs &a, const int bi, int tile, gs_acc_t *__restrict__ acc,
                                          unsigned short *__restrict__ list, int *nlist_p) {
    int &nlist = *nlist_p;
    const int D = tile >= a.tiles[0];
    if (D) tile -= a.tiles[0];
    const int S = 1 - D;
    const int j0 = tile * a.gt[D], jn = min(a.gt[D], a.npad[D] - j0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int4 *__restrict__ sr = a.rec[S] + (size_t)bi * a.npad[S];
    const int ng = a.npad[S] / SB;
    const unsigned long long *__restrict__ rg = a.mask[S] + (size_t)bi * ng;
    // this tile's buckets (gt is a whole number of buckets, at most 64 of them per set)
    const int b0 = j0 / a.qbucket[D], nb = (jn + a.qbucket[D] - 1) / a.qbucket[D];
    const unsigned long long tmask = (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull)) << b0;

    if (tid == 0) nlist = 0;
    // every independent load first: the groups' masks, this thread's destinations (original index, own term)
    constexpr int RG = (GS_MAXG + GS_TPB - 1) / GS_TPB;
    unsigned long long rr[RG];
#pragma unroll
    for (int u = 0; u < RG; u++) {
        const int gi = tid + u * GS_TPB;
        rr[u] = 0ull;
        if (gi < ng) rr[u] = rg[gi];
    }
    constexpr int OWN = GS_GT / GS_TPB;
    float own[OWN][3];
    int oo[OWN];
#pragma unroll
    for (int u = 0; u < OWN; u++) {
        const int j = tid + u * GS_TPB;
        oo[u] = -1;
        own[u][0] = own[u][1] = own[u][2] = 0.f;
        if (j < jn) {
            oo[u] = a.orig[D][(size_t)bi * a.npad[D] + j0 + j];
            const int4 rc = a.rec[D][(size_t)bi * a.npad[D] + j0 + j];
            own[u][0] = __int_as_float(rc.y);
            own[u][1] = __int_as_float(rc.z);
            own[u][2] = __int_as_float(rc.w);
        }
    }
    for (int i = tid; i < jn * 3; i += GS_TPB) acc[i] = (gs_acc_t)0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RG; u++)
        if (rr[u] & tmask) list[atomicAdd(&nlist, 1)] = (unsigned short)(tid + u * GS_TPB);
    __syncthreads();
    const int nl = nlist;
    for (int base = wave; base < nl; base += (GS_TPB / 64) * GS_KB) {
        int w[GS_KB];
        float v[GS_KB][3];
#pragma unroll
        for (int i = 0; i < GS_KB; i++) {
            const int li = base + i * (GS_TPB / 64);  // uniform
            w[i] = -1;
            v[i][0] = v[i][1] = v[i][2] = 0.f;
            if (li < nl) {
                const int4 rc = sr[(int)list[li] * SB + lane];
                w[i] = rc.x;
                v[i][0] = __int_as_float(rc.y);
                v[i][1] = __int_as_float(rc.z);
                v[i][2] = __int_as_float(rc.w);
            }
        }
#pragma unroll
        for (int i = 0; i < GS_KB; i++) {
            // The sums are DOUBLES: a CU's LDS adds 18 lanes per ns with ds_add_f64 and 0.8 with ds_add_f32 (80 ns for a wave's
            // instruction whatever the addresses: tools/ubench/lds_atomic_rate.hip), which is what this kernel was bound by
            // (16.3 -> 12.8 us, step 84 -> 80.5 us).  RFP_GS_SEG: the pre-reduction that made the fp32 atomics bearable -- runs of
            // equal destination inside a 16-lane row summed in registers (segmented scan by DPP row shifts), only the last lane
            // of a run touching LDS; it changes nothing any more and is off.
            const int j = w[i] - j0;
            const int key = (unsigned)j < (unsigned)jn ? j : -1;
            float sx = -v[i][0], sy = -v[i][1], sz = -v[i][2];
#if RFP_GS_SEG
            const int kprev = __builtin_amdgcn_update_dpp(-2, key, 0x111, 0xf, 0xf, false);  // row_shr:1
            const int knext = __builtin_amdgcn_update_dpp(-2, key, 0x101, 0xf, 0xf, false);  // row_shl:1
            int f = key != kprev;  // head of a run (or of the row)
#define RFP_SEG(CTRL)                                                                                   \
    {                                                                                                   \
        const int fp = __builtin_amdgcn_update_dpp(1, f, CTRL, 0xf, 0xf, false);                        \
        const float px = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sx), CTRL, 0xf, 0xf, false)); \
        const float py = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sy), CTRL, 0xf, 0xf, false)); \
        const float pz = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sz), CTRL, 0xf, 0xf, false)); \
        sx = f ? sx : sx + px;                                                                          \
        sy = f ? sy : sy + py;                                                                          \
        sz = f ? sz : sz + pz;                                                                          \
        f |= fp;                                                                                        \
    }
            RFP_SEG(0x111) RFP_SEG(0x112) RFP_SEG(0x114) RFP_SEG(0x118)
#undef RFP_SEG
            const bool last = key != knext;
#else
            const bool last = true;
#endif
            if (key >= 0 && last) {
                atomicAdd(&acc[key * 3 + 0], (gs_acc_t)sx);
                atomicAdd(&acc[key * 3 + 1], (gs_acc_t)sy);
                atomicAdd(&acc[key * 3 + 2], (gs_acc_t)sz);
            }
        }
    }
    __syncthreads();
    float *__restrict__ out = a.grad[D] + (size_t)bi * a.n[D] * 3;
#pragma unroll
    for (int u = 0; u < OWN; u++) {
        const int j = tid + u * GS_TPB;
        if (j < jn && oo[u] >= 0) {
            // (one 12-byte store per row: the rows leave in ORIGINAL order, every lane another cache line -- three dword stores were
            // three transactions per row)
            struct P3 {
                float x, y, z;
            };
            *(P3 *)(out + (size_t)oo[u] * 3) = P3{own[u][0] + (float)acc[j * 3 + 0], own[u][1] + (float)acc[j * 3 + 1], own[u][2] + (float)acc[j * 3 + 2]};
        }
    }
}

__global__ __launch_bounds__(GS_TPB) void nnp_grad_sorted_kernel(GradSArgs a) {
    extern __shared__ __attribute__((aligned(8))) unsigned char acc_dyn[];  // gs_acc_t [gt * 3], gt = the larger of the two sets' tile sizes
    __shared__ unsigned short list[GS_MAXG];
    __shared__ int nlist;
    // cloud-major logical order, each XCD a contiguous eighth (as sort and sweep): the re-reads of a cloud's
    // records by its tiles then hit that XCD's L2 instead of crossing the fabric once per tile
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    const int wpc = a.tiles[0] + a.tiles[1];
    const int bi = logical / wpc;
    grad_tile(a, bi, logical - bi * wpc, (gs_acc_t *)acc_dyn, list, &nlist);
}

int round_up(long v, int q) { return (int)((v + q - 1) / q * q); }

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// workgroups sorting one cloud of n points, and the padded record count of a sorted set: a split
// cloud carries one more superblock (half 0's segment is padded to a multiple of 64 on its own)
int sort_split_of(int n) { return (RFP_SORT_SPLIT > 1 && n > RFP_SORT_SPLIT_ABOVE && n <= RPT * STPB) ? RFP_SORT_SPLIT : 1; }
size_t npad_of(int n) { return (size_t)round_up(n, SB) + (size_t)(sort_split_of(n) - 1) * SB; }

}  // namespace

namespace rfp {

bool pruned_supported(int b, int n, int m) {
    return b > 0 && n > 0 && m > 0 && n <= kMaxPoints && m <= kMaxPoints;
}

// One sorted set (b clouds of n points) in a caller-owned buffer: xyz | orig | box16 | box64, each
// part 256-byte aligned.  The layout is a pure function of (b, n): a "handle" is just that buffer.
size_t sorted_bytes(int b, int n) {
    if (b <= 0 || n <= 0 || n > kMaxPoints) return 0;
    const size_t npad = npad_of(n);
    return align256((size_t)b * npad * 3 * sizeof(float) + 256)  // + prefetch overrun
           + align256((size_t)b * npad * sizeof(int)) + align256((size_t)b * (npad / SB) * B16F * sizeof(float)) +
           align256((size_t)b * (npad / SB) * B64F * sizeof(float)) + align256((size_t)3 * b * sizeof(int));  // pos0 (b) | crowded (b) | non-finite (b)
}

Sorted sorted_view(int b, int n, const void *buf) {
    const size_t npad = npad_of(n);
    const char *w = (const char *)buf;
    Sorted v;
    v.npad = (int)npad;
    v.xyz = (const float *)w;
    w += align256((size_t)b * npad * 3 * sizeof(float) + 256);
    v.orig = (const int *)w;
    w += align256((size_t)b * npad * sizeof(int));
    v.box16 = (const float *)w;
    w += align256((size_t)b * (npad / SB) * B16F * sizeof(float));
    v.box64 = (const float *)w;
    w += align256((size_t)b * (npad / SB) * B64F * sizeof(float));
    v.pos0 = (const int *)w;
    return v;
}

size_t pruned_workspace_bytes(int b, int n, int m) {
    if (!pruned_supported(b, n, m)) return 0;
    return sorted_bytes(b, n) + sorted_bytes(b, m) + align256(32 * sizeof(unsigned long long));
}

// Sort `nsets` (1 or 2) sets of b clouds in ONE launch (one workgroup per cloud).
int sort_sets(int b, int nsets, const int *n, const float *const *src, const Sorted *out, hipStream_t s,
              unsigned long long *dbg) {
    if (b <= 0 || nsets < 1 || nsets > 2) return RF_EINVAL;
    SortArgs sa;
    sa.b = b;
    sa.nsets = nsets;
    sa.dbg = dbg;
    bool reg = true;
    for (int k = 0; k < 2; k++) {
        const int kk = k < nsets ? k : 0;
        if (n[kk] <= 0 || n[kk] > kMaxPoints || !src[kk]) return RF_EINVAL;
        sa.n[k] = n[kk];
        sa.npad[k] = out[kk].npad;
        sa.src[k] = src[kk];
        sa.xyz[k] = const_cast<float *>(out[kk].xyz);
        sa.orig[k] = const_cast<int *>(out[kk].orig);
        sa.box16[k] = const_cast<float *>(out[kk].box16);
        sa.box64[k] = const_cast<float *>(out[kk].box64);
        sa.pos0[k] = const_cast<int *>(out[kk].pos0);
        reg = reg && n[kk] <= RPT * STPB;
        sa.split[k] = sort_split_of(n[kk]);
        {
            // slabs per axis = cbrt(n / leaf): leaves of RFP_STR_LARGE_LEAF = 48 records (round 4: 7 slabs instead of 6 at 16384
            // points measured step 91.6 -> 90.2 us; 32 and 96 measured like 64); clouds of up to RFP_STR_SMALL_N points (the side that is swept as
            // 16-query tiles) get leaves of RFP_STR_SMALL_LEAF: their 16-record blocks are then less flat -- the model
            // (tools/experiments, round 4) gives a 2048-point set 25 instead of 36 superblocks per tile list as queries
            // and 9.5 instead of 11.3 block scans per 64-query group as candidates
            const int leaf = n[kk] <= RFP_STR_SMALL_N ? RFP_STR_SMALL_LEAF : RFP_STR_LARGE_LEAF;
            int ss = (int)lround(cbrt((double)n[kk] / leaf));
            sa.str_s[k] = ss < 1 ? 1 : (ss > 16 ? 16 : ss);
        }
    }
    if (reg) {
        const int wpb = sa.split[0] + (nsets > 1 ? sa.split[1] : 0);
        RF_LAUNCH("nnp_sort", nnp_sort_reg_kernel, dim3(wpb * b), dim3(STPB), 0, s, sa);
    } else {
        RF_LAUNCH("nnp_sort", nnp_sort_kernel<false>, dim3(nsets * b), dim3(STPB), 0, s, sa);
    }
    return RF_OK;
}

// The sweep over two sorted sets.  dirs bit 0: nearest neighbour of every point of set 0 in set 1
// (-> dist1/idx1), bit 1: the opposite (-> dist2/idx2).  A direction that is not asked for costs
// nothing: its workgroups are not launched.
static int sweep_sorted_impl(int b, int n, int m, const Sorted &s0, const Sorted &s1, float *dist1, int *idx1,
                             float *dist2, int *idx2, int dirs, hipStream_t s, unsigned long long *stats_dev,
                             const GradEmit *ge) {
    if (!pruned_supported(b, n, m) || (dirs & 3) == 0) return RF_EINVAL;
    if (((dirs & 1) && (!dist1 || !idx1)) || ((dirs & 2) && (!dist2 || !idx2))) return RF_EINVAL;
    SweepArgs wa;
    wa.b = b;
    const int nn[2] = {n, m};
    const Sorted *ss[2] = {&s0, &s1};
    for (int k = 0; k < 2; k++) {
        wa.n[k] = nn[k];
        wa.npad[k] = ss[k]->npad;
        wa.groups[k] = ss[k]->npad / SB;
        // a set with few groups cannot fill the chip with one wave per group: 4 waves share a group
        wa.nw[k] = ((long)b * wa.groups[k] < RFP_SPLIT_BELOW) ? NSH : 1;
    }
    const bool want[2] = {(dirs & 1) != 0, (dirs & 2) != 0};
    {
        int longest = 0;  // entries in a wave's list: all superblocks of the other set, or a quarter
        for (int k = 0; k < 2; k++) {
            if (!want[k]) continue;
            const int nsb = wa.groups[1 - k];
            const int len = (wa.nw[k] == NSH && !RFP_TILE16) ? (nsb + NSH - 1) / NSH : nsb;  // (a tile's list may hold every superblock)
            longest = len > longest ? len : longest;
        }
        wa.kstride = (longest + 63) / 64 * 64;
    }
    const bool shared_groups = (want[0] && wa.nw[0] == NSH) || (want[1] && wa.nw[1] == NSH);
    const int tpb = shared_groups ? 64 * NSH : 64;
    const int pack = tpb / 64;  // one-wave groups per workgroup
    const size_t aux = !shared_groups ? 0 : (RFP_TILE16 && NSH * sizeof(T16Lds) > sizeof(SharedLds) ? NSH * sizeof(T16Lds) : sizeof(SharedLds));
    const size_t shmem = pack * wa.kstride * sizeof(unsigned) + aux;
    wa.wg0 = !want[0] ? 0 : (wa.nw[0] == NSH ? wa.groups[0] : rf::ceil_div(wa.groups[0], pack));
    wa.wg1 = !want[1] ? 0 : (wa.nw[1] == NSH ? wa.groups[1] : rf::ceil_div(wa.groups[1], pack));
    if (ge) {
        RF_LAUNCH("nnp_sweep", nnp_sweep_kernel<true>, dim3((unsigned)b * (wa.wg0 + wa.wg1)), dim3(tpb),
                  shmem, s, wa, *ge, s0.xyz, s1.xyz, s0.orig, s1.orig, s0.box16,
                  s1.box16, s0.box64, s1.box64, dist1, dist2, idx1, idx2, stats_dev);
    } else {
        RF_LAUNCH("nnp_sweep", nnp_sweep_kernel<false>, dim3((unsigned)b * (wa.wg0 + wa.wg1)), dim3(tpb),
                  shmem, s, wa, GradEmit{}, s0.xyz, s1.xyz, s0.orig, s1.orig, s0.box16,
                  s1.box16, s0.box64, s1.box64, dist1, dist2, idx1, idx2, stats_dev);
    }
    return RF_OK;
}

int sweep_sorted(int b, int n, int m, const Sorted &s0, const Sorted &s1, float *dist1, int *idx1, float *dist2,
                 int *idx2, int dirs, hipStream_t s, unsigned long long *stats_dev) {
    return sweep_sorted_impl(b, n, m, s0, s1, dist1, idx1, dist2, idx2, dirs, s, stats_dev, nullptr);
}

// ---- forward + backward of one Chamfer (rf_chamfer_step) on the culled path -----------------------------
// workspace: sorted(n) | sorted(m) | per set (emit_layout): rec (b, npad) int4 {wp, own} | mask (b, npad / 64) u64
static size_t step_set_bytes(int b, int n) { return emit_set_bytes(b, (int)npad_of(n)); }

size_t pruned_step_workspace_bytes(int b, int n, int m) {
    if (!pruned_supported(b, n, m)) return 0;
    return sorted_bytes(b, n) + sorted_bytes(b, m) + step_set_bytes(b, n) + step_set_bytes(b, m);
}

int pruned_step(int b, int n, int m, const float *xyz1, const float *xyz2, const float *gd1, const float *gd2,
                float *dist1, int *idx1, float *dist2, int *idx2, float *grad_xyz1, float *grad_xyz2, void *workspace,
                size_t workspace_bytes, hipStream_t s) {
    if (!pruned_supported(b, n, m) || !gd1 || !gd2 || !grad_xyz1 || !grad_xyz2 || !rf::aligned16(workspace)) return RF_EINVAL;
    if (workspace_bytes < pruned_step_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    char *w = (char *)workspace;
    const Sorted so[2] = {sorted_view(b, n, w), sorted_view(b, m, w + sorted_bytes(b, n))};
    const int nn[2] = {n, m};
    const float *src[2] = {xyz1, xyz2};
    const float *gds[2] = {gd1, gd2};
    float *grads[2] = {grad_xyz1, grad_xyz2};
    GradEmit ge;
    GradSArgs ga;
    ga.b = b;
    ge.base = w + sorted_bytes(b, n) + sorted_bytes(b, m);
    for (int k = 0; k < 2; k++) {
        const EmitView ev = emit_layout(ge.base, b, so[0].npad, so[1].npad, k);
        ge.gd[k] = gds[k];
        ga.n[k] = nn[k];
        ga.npad[k] = so[k].npad;
        // tile size: a whole number of buckets (64 per set), about 1024 workgroups per set
        const int q = rf::ceil_div(so[k].npad, 64);
        ge.qbucket[k] = ga.qbucket[k] = q;
        long want = ((long)so[k].npad * b + RFP_GS_WG - 1) / RFP_GS_WG;
        int nbk = (int)((want + q - 1) / q);
        nbk = nbk < 1 ? 1 : nbk;
        while (nbk > 1 && nbk * q > GS_GT) nbk--;
        const int gt = nbk * q;
        ga.gt[k] = gt;
        ga.tiles[k] = rf::ceil_div(so[k].npad, gt);
        ga.orig[k] = so[k].orig;
        ga.rec[k] = ev.rec;
        ga.mask[k] = ev.mask;
        ga.grad[k] = grads[k];
    }
    if (int e = sort_sets(b, 2, nn, src, so, s, nullptr)) return e;
    if (int e = sweep_sorted_impl(b, n, m, so[0], so[1], dist1, idx1, dist2, idx2, 3, s, nullptr, &ge)) return e;
    const int gtmax = ga.gt[0] > ga.gt[1] ? ga.gt[0] : ga.gt[1];
    RF_LAUNCH("nnp_grad_sorted", nnp_grad_sorted_kernel, dim3(b * (ga.tiles[0] + ga.tiles[1])), dim3(GS_TPB),
              (size_t)gtmax * 3 * sizeof(gs_acc_t), s, ga);
    return RF_OK;
}

int pruned_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                       float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s,
                       unsigned long long *stats_out, int dirs) {
    if (!pruned_supported(b, n, m) || !rf::aligned16(workspace)) return RF_EINVAL;
    if (workspace_bytes < pruned_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    char *w = (char *)workspace;
    const Sorted so[2] = {sorted_view(b, n, w), sorted_view(b, m, w + sorted_bytes(b, n))};
    unsigned long long *stats = nullptr;
    if (stats_out) {
        stats = (unsigned long long *)(w + sorted_bytes(b, n) + sorted_bytes(b, m));
        RF_ZERO(stats, 32 * sizeof(unsigned long long), s);
    }
    const int nn[2] = {n, m};
    const float *src[2] = {xyz1, xyz2};
    if (int e = sort_sets(b, 2, nn, src, so, s, stats)) return e;
    if (int e = sweep_sorted(b, n, m, so[0], so[1], dist1, idx1, dist2, idx2, dirs, s, stats)) return e;
    if (stats_out) {
        RF_HIP(hipMemcpyAsync(stats_out, stats, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        RF_HIP(hipStreamSynchronize(s));
    }
    return RF_OK;
}

size_t sort_workspace_bytes(int b, int n) { return sorted_bytes(b, n); }

int sort_clouds(int b, int n, const float *src, void *workspace, size_t workspace_bytes, hipStream_t s, Sorted *out) {
    if (b <= 0 || n <= 0 || n > kMaxPoints || !src || !workspace || !out || !rf::aligned16(workspace)) return RF_EINVAL;
    if (workspace_bytes < sorted_bytes(b, n)) return RF_EWORKSPACE;
    *out = sorted_view(b, n, workspace);
    return sort_sets(b, 1, &n, &src, out, s, nullptr);
}

}  // namespace rfp
