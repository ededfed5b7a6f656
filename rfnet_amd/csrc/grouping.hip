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
// serially (divergent early exit, uncoalesced AoS loads, one IEEE sqrt per pair).  Here:
//   * one wave64 owns QPW = 8 queries of one cloud: each step loads 64 consecutive dataset
//     points ONCE (coalesced) and tests them against the 8 queries, whose coordinates and
//     hit counters live in SGPRs -- 8x less L1/L2 traffic than one query per wave;
//   * the per-pair test is ONE compare: max(sqrt_rn(d2),1e-20) < r  <=>  d2 < T, where T is
//     the smallest float whose correctly-rounded square root reaches r (found on the host by
//     bisection over the float bit patterns; sqrt_rn is monotone), so no sqrt on the device
//     and still exactly the reference's distance-domain predicate;
//   * v_cmp writes the 64-bit hit mask (the ballot) straight to SGPRs; hits are rare
//     (~0.3 per step), so a scalar branch skips the slot assignment (v_mbcnt prefix) unless
//     the mask is non-zero; the wave stops when all its queries have nsample hits.
//   Hits go straight to their final slots; the padding [cnt, nsample) is written once at the
//   end, so no slot is written twice.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "nn_pruned.hpp"
#include "group_internal.hpp"
#include "scatter_rows.hpp"

namespace {

// Device twin of ball_threshold() below, for a radius that lives in device memory (the reference's
// op input tensor, tf_grouping.cpp:18,93-95): the smallest float T with sqrt_rn(T) >= radius, so
// that "max(sqrt_rn(d2), 1e-20f) < radius" <=> "d2 < T".  r*r is within a few ulps of T; walk the
// float bit patterns from there with the correctly rounded device sqrt (HIP's default for
// sqrtf, -fhip-fp32-correctly-rounded-divide-sqrt) -- a handful of uniform iterations per wave.
__device__ __forceinline__ float ball_threshold_dev(float radius) {
    if (!(radius > 1e-20f)) return 0.0f;  // also NaN: no hit ever
    float t = radius * radius;
    if (!(t < INFINITY)) t = INFINITY;
    unsigned u = __float_as_uint(t);
    // down while the predecessor still reaches the radius, then up until this one does
    int steps = 0;
    while (u > 0u && steps < 8 && __fsqrt_rn(__uint_as_float(u - 1u)) >= radius) { u--; steps++; }
    while (u < 0x7F800000u && steps < 16 && !(__fsqrt_rn(__uint_as_float(u)) >= radius)) { u++; steps++; }
    if (steps >= 8) {  // radii whose square is subnormal: plain bisection over the bit patterns
        unsigned lo = 0u, hi = 0x7F800000u;
        while (hi - lo > 1u) {
            const unsigned mid = lo + (hi - lo) / 2u;
            if (__fsqrt_rn(__uint_as_float(mid)) >= radius) hi = mid; else lo = mid;
        }
        u = hi;
    }
    return __uint_as_float(u);
}

constexpr int QB_TPB = 256;  // 4 waves per workgroup
constexpr int QPW = 8;       // queries per wave
constexpr int QB_NS = 64;    // nsample up to which the hit lists are staged in LDS

// STAGE: the hit lists (QPW x nsample indices per wave) are kept in LDS during the scan and
// written out once, coalesced, at the end instead of as scattered 4-byte stores inside the loop
// (C3: 0.187 -> 0.183 ms).  nsample > QB_NS falls back to direct stores.
// Where the time goes (rocprofv3 PMC, C3): the uniform-cube balls are truncated by the faces for
// half of the queries, so nearly every wave scans all n points; the loop issues 74 VALU + 49 SALU
// instructions per 64-point step and is bound by the per-SIMD issue rate (~2.3 cycles per
// instruction of either kind), not by memory (SQ_WAIT_INST_ANY = 20 % of wave cycles).
template <bool STAGE>
__global__ __launch_bounds__(QB_TPB) void query_ball_kernel(int n, int m, int nwaves_per_batch, int b,
                                                            float thresh, const float *__restrict__ radius_dev,
                                                            int nsample,
                                                            const float *__restrict__ xyz1,
                                                            const float *__restrict__ xyz2,
                                                            int *__restrict__ idx,
                                                            int *__restrict__ pts_cnt) {
    __shared__ int stage[STAGE ? QB_TPB / 64 : 1][QPW][STAGE ? QB_NS : 1];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x * (QB_TPB / 64) + wib;
    const int bi = w / nwaves_per_batch;
    if (bi >= b) return;
    if (radius_dev) thresh = ball_threshold_dev(radius_dev[0]);  // uniform
    const int n_scan = thresh > 0.f ? n : 0;  // radius within the 1e-20 clamp: nothing is ever inside
    const int q0 = (w - bi * nwaves_per_batch) * QPW;  // first query of this wave (within the cloud)
    const int nq = min(QPW, m - q0);
    const float *__restrict__ D = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + ((size_t)bi * m + q0) * 3;
    int *__restrict__ I = idx + ((size_t)bi * m + q0) * nsample;

    float qx[QPW], qy[QPW], qz[QPW];
    int cnt[QPW], first[QPW];
#pragma unroll
    for (int i = 0; i < QPW; i++) {
        const int ii = i < nq ? i : 0;  // uniform
        qx[i] = Q[ii * 3 + 0];
        qy[i] = Q[ii * 3 + 1];
        qz[i] = Q[ii * 3 + 2];
        cnt[i] = i < nq ? 0 : nsample;  // absent queries are "done"
        first[i] = -1;
    }
    // dataset points are fetched one step ahead of the tests (the early-exit loop would otherwise
    // expose the full L2 latency on every step)
    float nx, ny, nz;
    {
        const int kk = min(lane, n - 1);
        nx = D[kk * 3 + 0]; ny = D[kk * 3 + 1]; nz = D[kk * 3 + 2];
    }
    for (int k0 = 0; k0 < n_scan; k0 += 64) {
        int open = 0;
#pragma unroll
        for (int i = 0; i < QPW; i++) open |= (cnt[i] < nsample) ? 1 : 0;
        if (!open) break;
        const int k = k0 + lane;
        const float x1 = nx, y1 = ny, z1 = nz;
        {
            const int kk = min(k + 64, n - 1);
            nx = D[kk * 3 + 0]; ny = D[kk * 3 + 1]; nz = D[kk * 3 + 2];
        }
        const unsigned long long valid = __ballot(k < n);
        // all QPW tests first (independent VALU work), the scalar bookkeeping afterwards
        unsigned long long mask[QPW], any = 0ull;
#pragma unroll
        for (int i = 0; i < QPW; i++) {
            const float d2 = rf::d2_fma(qx[i] - x1, qy[i] - y1, qz[i] - z1);
            mask[i] = __ballot(!(d2 >= thresh)) & valid;  // a NaN d2 is a hit, as in the reference (fmaxf drops the NaN)
            any |= mask[i];
        }
        if (any == 0ull) continue;  // wave-uniform
#pragma unroll
        for (int i = 0; i < QPW; i++) {
            if (mask[i] != 0ull && cnt[i] < nsample) {  // wave-uniform
                if (first[i] < 0) first[i] = k0 + __builtin_ctzll(mask[i]);
                const int pos = cnt[i] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask[i] >> 32),
                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask[i], 0));
                if (((mask[i] >> lane) & 1ull) && pos < nsample) {
                    if (STAGE) stage[wib][i][pos] = k;
                    else I[(size_t)i * nsample + pos] = k;
                }
                cnt[i] = min(nsample, cnt[i] + __builtin_popcountll(mask[i]));
            }
        }
    }
    // the wave's own LDS writes are visible to it without a barrier (in-order LDS queue)
#pragma unroll
    for (int i = 0; i < QPW; i++) {
        if (i < nq) {
            if (cnt[i] > 0) {
                if (STAGE) {
                    for (int l = lane; l < nsample; l += 64)
                        I[(size_t)i * nsample + l] = l < cnt[i] ? stage[wib][i][l] : first[i];
                } else {
                    for (int l = cnt[i] + lane; l < nsample; l += 64) I[(size_t)i * nsample + l] = first[i];
                }
            }
            if (lane == 0) pts_cnt[(size_t)bi * m + q0 + i] = cnt[i];
        }
    }
}

// ---- query_ball_point, lanes <-> queries -------------------------------------------------------
// The kernel above keeps QPW queries in SGPRs and tests 64 dataset points per step: every step pays
// ~6 scalar instructions per query for mask bookkeeping (PMC: 49 SALU next to 74 VALU per step).
// Here the roles are swapped, as in the Chamfer sweep: a wave owns 64 QUERIES (one per lane) and the
// dataset is streamed through SGPRs by scalar loads; a test is 6 VALU + 1 compare, and the hit
// path (append k to the lane's list) runs under the exec mask only when some lane hit (~20 % of the
// points at C3).  To fill the chip the dataset is cut into QS segments scanned by the QS waves of a
// workgroup for the same 64 queries; each keeps the first `nsample` hits of its segment in LDS and
// the workgroup concatenates the segments in order at the end (no global scratch, no second launch).
constexpr int QL_SUB = 8;  // dataset points per scalar-load sub-chunk
template <int QS, int NSMAX>
__global__ __launch_bounds__(64 * QS) void query_ball_lanes_kernel(int n, int m, int n_pad, int seg,
                                                                   float thresh,
                                                                   const float *__restrict__ radius_dev,
                                                                   int nsample,
                                                                   const float *__restrict__ xyz1,
                                                                   const float *__restrict__ xyz2,
                                                                   int *__restrict__ idx,
                                                                   int *__restrict__ pts_cnt) {
    __shared__ int list[QS][NSMAX][64];  // [segment][slot][query lane]
    __shared__ int segcnt[QS][64];
    const int lane = threadIdx.x & 63;
    const int sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int q = blockIdx.x * 64 + lane;
    const int qq = min(q, m - 1);
    const float *__restrict__ D = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ Q = xyz2 + ((size_t)bi * m + qq) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    int cnt = 0;
    if (radius_dev) thresh = ball_threshold_dev(radius_dev[0]);  // uniform
    // threshold 0: the radius does not exceed the 1e-20 clamp, nothing is ever inside (not even a NaN)
    const int k_begin = sg * seg, k_end = thresh > 0.f ? min(n, k_begin + seg) : k_begin;
    // Whole sub-chunks of 8 points: scalar prefetch one sub-chunk ahead, no per-point range test.
    // The hit path sits behind a wave-uniform branch on the compare mask (without it the compiler
    // predicates the 5 append instructions and issues them for every point).
    const int k_full = k_begin + (max(k_end - k_begin, 0) / QL_SUB) * QL_SUB;
    // Two register sets, ping-pong: "wait for everything outstanding, issue the load of the OTHER
    // set, compute on this set".  (Scalar loads return out of order, so the only usable wait is
    // lgkmcnt(0); placed right before the next s_load it costs no overlap -- and unlike a
    // load-then-copy double buffer it needs no s_mov per operand: 3 SALU per 7 VALU here.)
    float pa[3 * QL_SUB], pb[3 * QL_SUB];
    auto fetch = [&](float (&dst)[3 * QL_SUB], int k) {
        const float *cp = D + (size_t)min(k, n_pad) * 3;  // uniform -> s_load; clamped in bounds
#pragma unroll
        for (int i = 0; i < 3 * QL_SUB; i++) dst[i] = cp[i];
    };
    auto scan8 = [&](const float (&c)[3 * QL_SUB], int k0) {
#pragma unroll
        for (int u = 0; u < QL_SUB; u++) {
            const float d2 = rf::d2_fma(qx - c[u * 3], qy - c[u * 3 + 1], qz - c[u * 3 + 2]);
            const bool hit = !(d2 >= thresh);  // NaN d2: a hit (the reference's fmaxf(sqrtf(NaN), 1e-20f) < r)
            if (__ballot(hit) != 0ull) {  // wave-uniform
                // (the empty volatile asm keeps this a real s_cbranch: without it the two
                // conditions are merged and the append is predicated instead of skipped)
                asm volatile("; some lane hit");
                if (hit && cnt < nsample) {
                    list[sg][cnt][lane] = k0 + u;
                    cnt++;
                }
            }
        }
    };
    fetch(pa, k_begin);
    for (int k0 = k_begin; k0 < k_full; k0 += 2 * QL_SUB) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): pa has arrived
        __builtin_amdgcn_sched_barrier(0);
        fetch(pb, k0 + QL_SUB);
        __builtin_amdgcn_sched_barrier(0);
        scan8(pa, k0);
        if (k0 + QL_SUB >= k_full) break;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // pb has arrived
        __builtin_amdgcn_sched_barrier(0);
        fetch(pa, k0 + 2 * QL_SUB);
        __builtin_amdgcn_sched_barrier(0);
        scan8(pb, k0 + QL_SUB);
        if (__ballot(cnt < nsample) == 0ull) break;  // all 64 queries full
    }
    // ragged tail of the cloud (n not a multiple of 8): the last 8 points, those below k_full skipped
    if (k_full < k_end) {
        const int ks = n - QL_SUB;
#pragma unroll 1
        for (int u = 0; u < QL_SUB; u++) {
            const int k = ks + u;
            const float d2 = rf::d2_fma(qx - D[k * 3], qy - D[k * 3 + 1], qz - D[k * 3 + 2]);
            if (!(d2 >= thresh) && k >= k_full && k < k_end && cnt < nsample) {
                list[sg][cnt][lane] = k;
                cnt++;
            }
        }
    }
    segcnt[sg][lane] = cnt;
    __syncthreads();
    // concatenate: query `ql`, output slot j <- the j-th hit over the segments in order
    int *__restrict__ I = idx + ((size_t)bi * m + (size_t)blockIdx.x * 64) * nsample;
    const int nq = min(64, m - (int)blockIdx.x * 64);
    for (int e = threadIdx.x; e < nq * nsample; e += 64 * QS) {
        const int ql = e / nsample, j = e - ql * nsample;
        int total = 0, val = -1, first = -1;
#pragma unroll
        for (int s2 = 0; s2 < QS; s2++) {
            const int c = segcnt[s2][ql];
            if (first < 0 && c > 0) first = list[s2][0][ql];
            if (val < 0 && j < total + c) val = list[s2][j - total][ql];
            total += c;
        }
        if (total > 0) I[e] = val >= 0 ? val : first;  // empty balls are left untouched
    }
    if (threadIdx.x < nq) {
        int total = 0;
#pragma unroll
        for (int s2 = 0; s2 < QS; s2++) total += segcnt[s2][threadIdx.x];
        pts_cnt[(size_t)bi * m + (size_t)blockIdx.x * 64 + threadIdx.x] = min(total, nsample);
    }
}


// ---- query_ball_point over the sorted cloud's boxes (round 5) -------------------------------------
// The scans above test every dataset point against every query: B*n*m pair tests (5.4e8 at C3) for balls that hold
// 0.4 % of the cloud.  Here the dataset is first put in sort-tile-recursive order (rfp::sort_clouds, the Chamfer
// sweep's sort: 64-record superblocks with their boxes) and ONE WAVE owns one query:
//   1. lanes <-> superblocks: the box's lower bound on d2 (the per-axis gaps through the same fma chain as a pair's
//      d2, so it never exceeds the d2 of a point inside the box) against the exact threshold T -- a superblock with
//      bound >= T holds no hit;
//   2. lanes <-> the 64 records of each surviving superblock: the reference's predicate on every record
//      (!(d2 >= T): a NaN d2 is a hit, tf_grouping_g.cu:24-27 through fmaxf), hits set their ORIGINAL index's bit in
//      a per-wave LDS bitmap -- the reference keeps the nsample LOWEST original indices in ascending order, and a
//      bitmap orders any number of hits for the price of one LDS atomic each;
//   3. the bitmap is walked from index 0 (64 words per step, popcounts prefix-summed by DPP row shifts) until
//      nsample hits are placed; the row is staged in LDS and written as one coalesced store, padded with the first
//      hit (tf_grouping_g.cu:26-29); pts_cnt = min(hits, nsample); a row without hits is not written.
// A ball that reaches more than 1/8 of the superblocks, a query with a non-finite coordinate and a cloud with a
// non-finite point (its box cannot bound a NaN) take the other exact route inside the same kernel: the wave walks the
// ORIGINAL cloud in index order, 64 points per step, and stops at nsample hits -- which is soon, a ball that wide holds
// a good share of the points.  So every query runs the reference's predicate on every point that can satisfy it,
// whatever the radius: results are bit-identical to the scan kernels' (tests/test_gpu_sampling_grouping.py).
constexpr int QX_LIST = 128;  // surviving superblocks a boxed query may hold: max(16, G / 8) <= 128 for G <= 1024
constexpr int QX_STAGE = 64;  // nsample <= 64
#ifndef RFG_QX_BATCH
#define RFG_QX_BATCH 4
#endif
constexpr int QX_BATCH = RFG_QX_BATCH;  // surviving superblocks whose records are fetched together

struct QxP3 {  // one packed record of the sorted cloud (12 bytes, 4-byte aligned): ONE global_load_dwordx3
    float x, y, z;
};
// max(a, b, 0) as the one instruction it is (fmaxf chains are wrapped in canonicalising v_max x,x under -fno-fast-math;
// the operands here are never NaN: finite query, box bounds finite or +-inf)
__device__ __forceinline__ float qx_gap(float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int N>
__device__ __forceinline__ int dpp_row_shr(int v) {  // lane - N inside its row of 16, 0 where there is none
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, false);
}
// inclusive prefix sum over the wave's 64 lanes
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    v += dpp_row_shr<1>(v);
    v += dpp_row_shr<2>(v);
    v += dpp_row_shr<4>(v);
    v += dpp_row_shr<8>(v);
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31),
              r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
}

// LDS (dynamic): the cloud's superblock boxes as six arrays of gpad floats (lo.x lo.y lo.z hi.x hi.y hi.z: lanes read
// consecutive dwords, conflict-free), staged ONCE per workgroup -- with every wave reading its 8 KB of boxes from global
// memory the box phase alone was 20 of the launch's 36 us at C3 (the vector L1's 64 B/clk) -- then per wave its bitmap,
// survivor list and output row.
__global__ __launch_bounds__(1024) void query_ball_boxes_kernel(
    int n, int m, int npad, int words /* bitmap words per wave, a multiple of 128 */, float thresh,
    const float *__restrict__ radius_dev, int nsample, const float *__restrict__ xyz1,
    const float *__restrict__ xyz2, const float *__restrict__ sxyz, const int *__restrict__ sorig,
    const float *__restrict__ box64, const int *__restrict__ nonfinite, int *__restrict__ idx,
    int *__restrict__ pts_cnt, float *__restrict__ grouped /* (b, m, nsample, 3) or NULL */, int zero_empty) {
    extern __shared__ __attribute__((aligned(16))) unsigned qx_lds[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    // a sample's workgroups on ONE XCD (rf::xcd_contiguous): its sorted records and boxes (270 KB at 16384 points) then stay in
    // that XCD's L2 -- spread over all eight, 32 samples are 8.6 MB per 4 MB L2
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);
    const int bpb = (m + wpb - 1) / wpb;  // workgroups per sample
    const int bi = logical / bpb;
    const int q = (logical - bi * bpb) * wpb + wib;
    const int G = npad >> 6;
    const int gpad = (G + 63) & ~63;
    float *__restrict__ bx = (float *)qx_lds;
    unsigned *__restrict__ bm = qx_lds + 6 * gpad + (size_t)wib * (words + QX_LIST + QX_STAGE);
    int *__restrict__ list = (int *)(bm + words);
    int *__restrict__ stage = list + QX_LIST;
    const float *__restrict__ BX = box64 + (size_t)bi * G * 8;
    for (int g = threadIdx.x; g < gpad; g += blockDim.x) {
        float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
        if (g < G) {
            lo = *(const float4 *)(BX + (size_t)g * 8);
            hi = *(const float4 *)(BX + (size_t)g * 8 + 4);
        }
        bx[g] = lo.x; bx[gpad + g] = lo.y; bx[2 * gpad + g] = lo.z;
        bx[3 * gpad + g] = hi.x; bx[4 * gpad + g] = hi.y; bx[5 * gpad + g] = hi.z;
    }
    __syncthreads();  // the only workgroup barrier: from here on every wave is on its own
    if (q >= m) return;
    if (radius_dev) thresh = ball_threshold_dev(radius_dev[0]);  // uniform
    const float *__restrict__ Q = xyz2 + ((size_t)bi * m + q) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    int *__restrict__ I = idx + ((size_t)bi * m + q) * nsample;
    const float *__restrict__ SX = sxyz + (size_t)bi * npad * 3;
    const int *__restrict__ SO = sorig + (size_t)bi * npad;
    // The row leaves through here.  An EMPTY ball: pts_cnt = 0 and the row is left untouched, as the reference leaves it
    // (tf_grouping_g.cu:18-33 writes nothing without a hit) -- unless the caller asked for defined contents (zero_empty: the
    // one-call sample-and-group gathers through the row next): index 0 then.  With `grouped` the coordinates of the row's
    // points go out with it (group_point of the same indices, tf_grouping_g.cu:40-57, fused: the row is in LDS).
    auto finish = [&](int cnt) {
        if (lane == 0) pts_cnt[(size_t)bi * m + q] = cnt;
        if (cnt == 0 && !zero_empty) return;
        // (the wave's own LDS writes are visible to it without a barrier: in-order LDS queue)
        const int first = cnt > 0 ? stage[0] : 0;
        const QxP3 *__restrict__ D3 = (const QxP3 *)(xyz1 + (size_t)bi * n * 3);
        QxP3 *__restrict__ GR = grouped ? (QxP3 *)grouped + ((size_t)bi * m + q) * nsample : nullptr;
        for (int l = lane; l < nsample; l += 64) {
            const int v = l < cnt ? stage[l] : first;
            I[l] = v;
            if (GR) GR[l] = D3[v];
        }
    };
    if (!(thresh > 0.f)) {  // the radius does not exceed the 1e-20 clamp: nothing is ever inside
        finish(0);
        return;
    }
    const bool qfinite = isfinite(qx) && isfinite(qy) && isfinite(qz);
    bool by_index = !qfinite || nonfinite[bi] != 0;  // (uniform)
    int total = 0;  // hits (uniform)
    int S = 0;      // surviving superblocks (uniform)
    if (!by_index) {
        // 1. lanes <-> superblocks, 64 at a time from the staged boxes (an empty padding box has lo = +inf: never live);
        //    survivors are appended to the wave's list
        for (int r0 = 0; r0 < gpad; r0 += 64) {
            const int g = r0 + lane;
            const float gx = qx_gap(bx[g] - qx, qx - bx[3 * gpad + g]);
            const float gy = qx_gap(bx[gpad + g] - qy, qy - bx[4 * gpad + g]);
            const float gz = qx_gap(bx[2 * gpad + g] - qz, qz - bx[5 * gpad + g]);
            const bool live = rf::d2_fma(gx, gy, gz) < thresh;
            const unsigned long long mask = __ballot(live);
            if (mask != 0ull) {  // (uniform)
                const int pos = S + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (live && pos < QX_LIST) list[pos] = g;
                S += __builtin_popcountll(mask);
            }
        }
        by_index = S > max(16, G >> 3);  // a wide ball: by index instead (S <= QX_LIST otherwise)
    }
    if (!by_index) {
        for (int w = lane * 4; w < words; w += 256) *(uint4 *)(bm + w) = make_uint4(0u, 0u, 0u, 0u);
        // 2. lanes <-> records, up to QX_BATCH surviving superblocks at a time (their loads in flight together); a batch
        //    slot beyond the list is skipped by a scalar branch, not padded
        int hit_any = 0;
        const QxP3 *__restrict__ RP = (const QxP3 *)SX + lane;
        const int *__restrict__ OP = SO + lane;
        for (int i = 0; i < S; i += QX_BATCH) {
            QxP3 p[QX_BATCH];
            int o[QX_BATCH];
#pragma unroll
            for (int u = 0; u < QX_BATCH; u++) {
                if (i + u < S) {  // (uniform)
                    const int sb = __builtin_amdgcn_readfirstlane(list[i + u]);
                    p[u] = RP[sb * 64];
                    o[u] = OP[sb * 64];
                }
            }
#pragma unroll
            for (int u = 0; u < QX_BATCH; u++) {
                if (i + u < S) {  // (uniform)
                    const float d2 = rf::d2_fma(qx - p[u].x, qy - p[u].y, qz - p[u].z);
                    if (!(d2 >= thresh) && o[u] >= 0) {  // padding records carry orig = -1
                        atomicOr(&bm[o[u] >> 5], 1u << (o[u] & 31));
                        hit_any = 1;
                    }
                }
            }
        }
        if (__ballot(hit_any != 0) == 0ull) {  // empty ball
            finish(0);
            return;
        }
        // 3. the bitmap in index order, 128 words per step (two per lane)
        int placed = 0;  // (uniform)
        for (int w0 = 0; w0 < words && placed < nsample; w0 += 128) {
            const uint2 v2 = *(const uint2 *)(bm + w0 + 2 * lane);
            unsigned long long val = ((unsigned long long)v2.y << 32) | v2.x;
            const int c = __builtin_popcountll(val);
            const int incl = wave_incl_scan(c, lane);
            int p = placed + incl - c;
            while (val != 0ull && p < nsample) {
                stage[p] = (w0 + 2 * lane) * 32 + __builtin_ctzll(val);
                val &= val - 1ull;
                p++;
            }
            placed += __builtin_amdgcn_readlane(incl, 63);
        }
        total = placed;  // >= 1; may exceed nsample by the last step's surplus
    } else {
        // the cloud in index order, 64 points per step, until nsample hits
        const float *__restrict__ D = xyz1 + (size_t)bi * n * 3;
        for (int k0 = 0; k0 < n && total < nsample; k0 += 64) {
            const int k = k0 + lane;
            const int kk = min(k, n - 1);
            const float d2 = rf::d2_fma(qx - D[(size_t)kk * 3], qy - D[(size_t)kk * 3 + 1], qz - D[(size_t)kk * 3 + 2]);
            const bool hit = !(d2 >= thresh) && k < n;
            const unsigned long long mask = __ballot(hit);
            if (mask == 0ull) continue;
            const int pos = total + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (hit && pos < nsample) stage[pos] = k;
            total += __builtin_popcountll(mask);
        }
    }
    finish(min(total, nsample));
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

// c == 3 (the model's use: coordinates): a row per lane, one 12-byte load and one 12-byte store, the batch element from the grid
// (the general kernel spends two 64-bit divisions per ELEMENT: 13.6 us at C3 against 5 here)
__global__ void group_point3_kernel(int n, int per_batch, const float *__restrict__ points, const int *__restrict__ idx,
                                    float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= per_batch) return;
    const size_t bi = blockIdx.y, js = bi * per_batch + r;
    struct P3 {
        float x, y, z;
    };
    const int ii = idx[js];
    *(P3 *)(out + js * 3) = *(const P3 *)(points + (bi * n + ii) * 3);
}

// Any c: a thread row per (query, sample) slot -- its index loaded once -- and the channels as VEC-wide vectors over the row's
// lanes, 32-bit arithmetic, the batch element from the grid (the element-per-thread kernels below pay two 64-bit divisions per
// ELEMENT; they remain for what does not fit 32 bits).  GRAD: the same walk, adding grad_out into grad_points.
constexpr int GPR_TPB = 256;
constexpr int GPR_PP = 4;  // slots per thread row and block
typedef float gpr_v4f __attribute__((ext_vector_type(4)));
template <int VEC, bool GRAD>
__global__ __launch_bounds__(GPR_TPB) void group_point_rows_kernel(int n, int c, int per_batch, int tx_log2, int bpb /* blocks per sample */,
                                                                   const float *__restrict__ src /* points | grad_out */,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ dst /* out | grad_points */) {
    typedef typename std::conditional<VEC == 4, gpr_v4f, float>::type V;
    // (a sample's blocks on ONE XCD: the rows it gathers cross the fabric once instead of eight times)
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);  // a sample's workgroups on one XCD
    const size_t bi = logical / bpb;
    const int bx = logical - (unsigned)bi * bpb;
    const int TX = 1 << tx_log2, TY = GPR_TPB >> tx_log2;
    const int lx = threadIdx.x & (TX - 1), ly = threadIdx.x >> tx_log2;
    const int cv = c / VEC;
    const int *__restrict__ I = idx + bi * per_batch;
    const float *__restrict__ P = GRAD ? src + bi * per_batch * c : src + bi * n * c;  // rows of slots | rows of points
    float *__restrict__ O = GRAD ? dst + bi * n * c : dst + bi * per_batch * c;
    int js[GPR_PP], row[GPR_PP];
#pragma unroll
    for (int u = 0; u < GPR_PP; u++) {
        js[u] = (bx * GPR_PP + u) * TY + ly;
        row[u] = I[min(js[u], per_batch - 1)];
    }
    for (int l = lx; l < cv; l += TX) {
        V v[GPR_PP];
#pragma unroll
        for (int u = 0; u < GPR_PP; u++) v[u] = ((const V *)P)[(size_t)(GRAD ? min(js[u], per_batch - 1) : row[u]) * cv + l];
#pragma unroll
        for (int u = 0; u < GPR_PP; u++) {
            if (js[u] >= per_batch) continue;
            if (!GRAD) {
                __builtin_nontemporal_store(v[u], &((V *)O)[(size_t)js[u] * cv + l]);
            } else {
                float *g = O + ((size_t)row[u] * cv + l) * VEC;
                if constexpr (VEC == 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) atomicAdd(g + k, v[u][k]);
                } else {
                    atomicAdd(g, v[u]);
                }
            }
        }
    }
}

template <bool GRAD>
static int group_rows_launch(int b, int n, int c, long per_batch, const float *src, const int *idx, float *dst, const char *name,
                             hipStream_t s) {
    // (the gradient one channel per lane: a wave's atomic instruction then covers whole cache lines -- four adds per lane at a stride
    // of 16 bytes across the lanes measured 0.91 ms against 0.24 for 32 x 1024 x 32 slots of 64 channels)
    const bool vec = !GRAD && c % 4 == 0 && rf::aligned16(src) && rf::aligned16(dst);
    const int cv = vec ? c / 4 : c;
    int tx_log2 = 0;
    while ((1 << tx_log2) < cv && tx_log2 < 6) tx_log2++;
    const int spb = (GPR_TPB >> tx_log2) * GPR_PP;  // slots per block
    const long bpb = (per_batch + spb - 1) / spb;
    const dim3 grid((unsigned)(bpb * b));
    if (vec) {
        RF_LAUNCH(name, (group_point_rows_kernel<4, GRAD>), grid, dim3(GPR_TPB), 0, s, n, c, (int)per_batch, tx_log2, (int)bpb, src, idx, dst);
    } else {
        RF_LAUNCH(name, (group_point_rows_kernel<1, GRAD>), grid, dim3(GPR_TPB), 0, s, n, c, (int)per_batch, tx_log2, (int)bpb, src, idx, dst);
    }
    return RF_OK;
}
static bool group_rows_ok(int b, int n, int c, long per_batch) {
    return b <= 65535 && per_batch * b < (1L << 31) && per_batch * c < (1L << 31) && (long)n * c < (1L << 31) && per_batch < (1L << 30);
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

// smallest float T such that max(sqrt_rn(x), 1e-20f) >= radius for every x >= T, i.e.
// "max(sqrtf(d2),1e-20f) < radius"  <=>  "d2 < T" for d2 >= 0.  Bisection over the (monotone)
// non-negative float bit patterns with the host's correctly rounded sqrtf.
static float ball_threshold(float radius) {
    if (!(radius > 1e-20f)) return 0.0f;  // the clamp alone already reaches the radius: no hit ever
    unsigned lo = 0u, hi = 0x7F800000u;   // sqrt(+0) = 0 < radius ; sqrt(+inf) = inf >= radius
    if (!(sqrtf(INFINITY) >= radius)) return INFINITY;  // radius is NaN-like: unreachable (guarded above)
    while (hi - lo > 1u) {
        unsigned mid = lo + (hi - lo) / 2u;
        float x;
        memcpy(&x, &mid, 4);
        if (sqrtf(x) >= radius) hi = mid; else lo = mid;
    }
    float t;
    memcpy(&t, &hi, 4);
    return t;
}


namespace rfi {
// the boxed ball query on a sorted dataset (group_internal.hpp); the domain is the caller's to check
int ball_boxes(int b, int n, int m, float radius, const float *radius_dev, int nsample, const float *xyz1, const float *xyz2,
               const rfp::Sorted &so, int *idx, int *pts_cnt, float *grouped_xyz, int zero_empty, hipStream_t s) {
    const float thresh = radius_dev ? 1.f : ball_threshold(radius);
    const int words = ((n + 31) / 32 + 127) / 128 * 128;
    const int gpad = (so.npad / 64 + 63) & ~63;
    // queries (waves) per workgroup: as many as share one staging of the boxes without the per-wave bitmaps crowding the LDS
    // (C3, same device: 4 / 8 / 16 queries per workgroup 33.1 / 31.9 / 34.0 us)
    const int wpb = n <= 32768 ? 8 : 4;
    const size_t shmem = sizeof(unsigned) * ((size_t)6 * gpad + (size_t)wpb * (words + QX_LIST + QX_STAGE));
    RF_LAUNCH("query_ball_boxes", query_ball_boxes_kernel, dim3(rf::ceil_div(m, wpb) * b), dim3(64 * wpb), shmem, s, n, m,
              so.npad, words, thresh, radius_dev, nsample, xyz1, xyz2, so.xyz, so.orig, so.box64, so.pos0 + 2 * b, idx, pts_cnt,
              grouped_xyz, zero_empty);
    return RF_OK;
}
}  // namespace rfi

extern "C" {

static int queryball_impl(int b, int n, int m, float radius, const float *radius_dev, int nsample,
                          const float *xyz1, const float *xyz2, int *idx, int *pts_cnt, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || nsample <= 0) return RF_EINVAL;
    long nquery = (long)b * m;
    if (nquery == 0) return RF_OK;
    if (!xyz2 || !idx || !pts_cnt || (n > 0 && !xyz1)) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {  // empty dataset: every ball is empty
        RF_ZERO(pts_cnt, sizeof(int) * (size_t)nquery, s);
        return RF_OK;
    }
    const int wpb = rf::ceil_div(m, QPW);
    const long waves = (long)b * wpb;
    const float thresh = radius_dev ? 0.f : ball_threshold(radius);
    // lanes <-> queries form: needs n >= 8 (scalar prefetch of whole sub-chunks) and b <= 65535
    if (n >= QL_SUB && b <= 65535 && nsample <= 64) {
        const int n_pad = n - QL_SUB;  // last sub-chunk start that stays in bounds
        const dim3 g(rf::ceil_div(m, 64), b);
        if (nsample <= 32) {
            constexpr int QS = 8;
            const int seg = rf::ceil_div(rf::ceil_div(n, QS), QL_SUB) * QL_SUB;
            RF_LAUNCH("query_ball_point", (query_ball_lanes_kernel<QS, 32>), g, dim3(64 * QS), 0, s, n, m, n_pad,
                      seg, thresh, radius_dev, nsample, xyz1, xyz2, idx, pts_cnt);
        } else {
            constexpr int QS = 4;
            const int seg = rf::ceil_div(rf::ceil_div(n, QS), QL_SUB) * QL_SUB;
            RF_LAUNCH("query_ball_point", (query_ball_lanes_kernel<QS, 64>), g, dim3(64 * QS), 0, s, n, m, n_pad,
                      seg, thresh, radius_dev, nsample, xyz1, xyz2, idx, pts_cnt);
        }
        return RF_OK;
    }
    if (nsample <= QB_NS) {
        RF_LAUNCH("query_ball_point", query_ball_kernel<true>, dim3(rf::ceil_div(waves, QB_TPB / 64)),
                  dim3(QB_TPB), 0, s, n, m, wpb, b, thresh, radius_dev, nsample, xyz1, xyz2, idx, pts_cnt);
    } else {
        RF_LAUNCH("query_ball_point", query_ball_kernel<false>, dim3(rf::ceil_div(waves, QB_TPB / 64)),
                  dim3(QB_TPB), 0, s, n, m, wpb, b, thresh, radius_dev, nsample, xyz1, xyz2, idx, pts_cnt);
    }
    return RF_OK;
}

int rf_queryballpoint(int b, int n, int m, float radius, int nsample, const float *xyz1,
                      const float *xyz2, int *idx, int *pts_cnt, rf_stream_t stream) {
    return queryball_impl(b, n, m, radius, nullptr, nsample, xyz1, xyz2, idx, pts_cnt, stream);
}

int rf_queryballpoint_dev(int b, int n, int m, const float *radius_dev, int nsample, const float *xyz1,
                          const float *xyz2, int *idx, int *pts_cnt, rf_stream_t stream) {
    if (!radius_dev) return RF_EINVAL;
    return queryball_impl(b, n, m, 0.f, radius_dev, nsample, xyz1, xyz2, idx, pts_cnt, stream);
}


// ---- the boxed form: needs scratch (the sorted copy of the dataset unless the caller hands a rf_nn_sort handle over)
size_t rf_queryballpoint_boxes_workspace_bytes(int b, int n) {
    if (b <= 0 || n < 64 || !rfp::pruned_supported(b, n, n)) return 0;
    return rfp::sorted_bytes(b, n);
}

int rf_queryballpoint_boxes(int b, int n, int m, float radius, const float *radius_dev, int nsample, const float *xyz1,
                            const float *xyz2, const void *sorted1, int *idx, int *pts_cnt, void *workspace,
                            size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || nsample <= 0) return RF_EINVAL;
    if ((long)b * m == 0) return RF_OK;
    if (n < 64 || nsample > 64 || b > 65535 || !rfp::pruned_supported(b, n, n)) return RF_EINVAL;  // the scan kernels' domain
    if (!xyz1 || !xyz2 || !idx || !pts_cnt || !workspace || !rf::aligned16(workspace)) return RF_EINVAL;
    if (sorted1 && !rf::aligned16(sorted1)) return RF_EINVAL;
    if (workspace_bytes < rf_queryballpoint_boxes_workspace_bytes(b, n)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    rfp::Sorted so;
    if (sorted1) {
        so = rfp::sorted_view(b, n, sorted1);
    } else {
        so = rfp::sorted_view(b, n, workspace);
        const int nn[1] = {n};
        const float *src[1] = {xyz1};
        if (int e = rfp::sort_sets(b, 1, nn, src, &so, s, nullptr)) return e;
    }
    return rfi::ball_boxes(b, n, m, radius, radius_dev, nsample, xyz1, xyz2, so, idx, pts_cnt, nullptr, 0, s);
}

int rf_grouppoint(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                  float *out, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || m < 0 || nsample < 0) return RF_EINVAL;
    long per_batch = (long)m * nsample;
    long total = (long)b * per_batch * c;
    if (total == 0) return RF_OK;
    if (!points || !idx || !out) return RF_EINVAL;
    if (c == 3 && per_batch < (1L << 30) && b <= 65535) {
        RF_LAUNCH("group_point", group_point3_kernel, dim3(rf::ceil_div((int)per_batch, 256), b), dim3(256), 0,
                  (hipStream_t)stream, n, (int)per_batch, points, idx, out);
        return RF_OK;
    }
    if (group_rows_ok(b, n, c, per_batch)) return group_rows_launch<false>(b, n, c, per_batch, points, idx, out, "group_point", (hipStream_t)stream);
    RF_LAUNCH("group_point", group_point_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, n, c, per_batch, total, points, idx, out);
    return RF_OK;
}

// From this many gradient elements on the sorted-slots form (scatter_rows.hip) beats the atomics (same device, tools/ab_group_grad.py);
// below, its two launches cost more than the atomics' contention
// (4.2 M elements: 28 against 27 us; 16.8 M: 34 against 63).  Narrow rows pay earlier -- the atomics' cost is their count and
// their crowding on a line, not the bytes: 32 x 32768 slots of 3 channels, 3.1 M elements, 26 against 63 us -- hence the slot rule.
constexpr long GPG_CSR_MIN_ELEMS = 1L << 22, GPG_CSR_MIN_SLOTS = 1L << 19;
static bool gpg_csr(int b, int n, int c, long per_batch) {
    return ((long)b * per_batch * c >= GPG_CSR_MIN_ELEMS || (long)b * per_batch >= GPG_CSR_MIN_SLOTS) &&
           rfs::rows_csr_supported(b, n, c, per_batch, 1);
}

size_t rf_grouppoint_grad_workspace_bytes(int b, int n, int c, int m, int nsample) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0 || nsample <= 0) return 0;
    const long per_batch = (long)m * nsample;
    return gpg_csr(b, n, c, per_batch) ? rfs::rows_csr_workspace_bytes(b, n, per_batch) : 0;
}

int rf_grouppoint_grad_ws(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx, float *grad_points,
                          void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || c < 0 || m < 0 || nsample < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long per_batch = (long)m * nsample;
    const long total = (long)b * per_batch * c;
    if ((size_t)b * n * c && !grad_points) return RF_EINVAL;
    if (total != 0 && n != 0 && (!grad_out || !idx)) return RF_EINVAL;
    const size_t need = total != 0 && n != 0 ? rf_grouppoint_grad_workspace_bytes(b, n, c, m, nsample) : 0;
    if (need && workspace && workspace_bytes >= need && rf::aligned16(workspace))
        return rfs::rows_csr_scatter(b, n, c, per_batch, 1, grad_out, idx, nullptr, grad_points, workspace, "group_point_grad_sort",
                                     "group_point_grad", s);
    if ((size_t)b * n * c) RF_ZERO(grad_points, sizeof(float) * (size_t)b * n * c, s);
    if (total == 0 || n == 0) return RF_OK;
    if (group_rows_ok(b, n, c, per_batch)) return group_rows_launch<true>(b, n, c, per_batch, grad_out, idx, grad_points, "group_point_grad", s);
    RF_LAUNCH("group_point_grad", group_point_grad_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              s, n, c, per_batch, total, grad_out, idx, grad_points);
    return RF_OK;
}

int rf_grouppoint_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                       const int *idx, float *grad_points, rf_stream_t stream) {
    return rf_grouppoint_grad_ws(b, n, c, m, nsample, grad_out, idx, grad_points, nullptr, 0, stream);
}

}  // extern "C"
