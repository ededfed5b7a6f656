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
//
// three_nn over SORTED clouds (three_nn_boxes_kernel, rf_threenn_boxes): both sets in the spatial order of the Chamfer sweep's
// sort (nn_pruned.hip: 64-record superblocks and 16-record blocks with their boxes).  A wave takes 64 consecutive sorted
// unknown points -- a compact cell -- and visits only the candidate blocks whose box can still hold a point at or inside some
// lane's third-best distance.  The bound is the SAME unfused fp32 expression evaluated on the per-axis gaps to the box: rounding
// is monotone, so bound <= distance holds in fp32 exactly and nothing that the full scan would insert is skipped.  The scan
// visits candidates in index order and inserts on strict '<': its result is the three smallest by (distance, index); the boxed
// form visits them in any order and inserts by that pair -- the same triple, ties included.
#include "common.hpp"
#include "scatter_rows.hpp"
#include "nn_pruned.hpp"

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


// ---- three_nn over sorted clouds ----------------------------------------------------------------------------------------
#ifndef RFI_TB_WAVES
#define RFI_TB_WAVES 4
#endif
constexpr int TB_WAVES = RFI_TB_WAVES;  // waves per workgroup, each on its own (no barrier)

#define TB_ROW(OP, N) asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float tb_wave_max(float v) {  // uniform result; inputs not NaN
    TB_ROW("v_max_f32_dpp", 8);
    TB_ROW("v_max_f32_dpp", 4);
    TB_ROW("v_max_f32_dpp", 2);
    TB_ROW("v_max_f32_dpp", 1);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float tb_wave_min(float v) {
    TB_ROW("v_min_f32_dpp", 8);
    TB_ROW("v_min_f32_dpp", 4);
    TB_ROW("v_min_f32_dpp", 2);
    TB_ROW("v_min_f32_dpp", 1);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
#undef TB_ROW

// squared distance from a point (or, with plo != phi, a box) to a box, per axis the gap max(lo - phi, plo - hi, 0): the
// unfused expression of the op itself, so that bound <= d in fp32 (header).  An empty box (lo = +inf, hi = -inf) is at +inf.
__device__ __forceinline__ float tb_gap(float a, float b) { return fmaxf(fmaxf(a, b), 0.f); }
__device__ __forceinline__ float tb_bound(float lx, float ly, float lz, float hx, float hy, float hz, float pxl, float pyl,
                                          float pzl, float pxh, float pyh, float pzh) {
    const float gx = tb_gap(lx - pxh, pxl - hx), gy = tb_gap(ly - pyh, pyl - hy), gz = tb_gap(lz - pzh, pzl - hz);
    return (gx * gx + gy * gy) + gz * gz;
}

__global__ __launch_bounds__(64 * TB_WAVES) void three_nn_boxes_kernel(
    int n, int npq, int npc, const float *__restrict__ qxyz, const int *__restrict__ qorig, const float *__restrict__ qb64,
    const float *__restrict__ cxyz, const int *__restrict__ corig, const float *__restrict__ cb16,
    const float *__restrict__ cb64, float *__restrict__ dist, int *__restrict__ idx) {
    const int lane = threadIdx.x & 63;
    // a sample's workgroups on the XCD that sorted it (rf::xcd_contiguous, as nnp_sort): its records are still in that L2
    const int bpb = ((npq >> 6) + TB_WAVES - 1) / TB_WAVES;  // workgroups per sample
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);  // a sample's workgroups on one XCD
    const int bi = logical / bpb;
    const int group = (logical - bi * bpb) * TB_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (group * 64 >= npq) return;  // (uniform)
    const int p = group * 64 + lane;
    const float *__restrict__ Q = qxyz + ((size_t)bi * npq + p) * 3;
    const float x1 = Q[0], y1 = Q[1], z1 = Q[2];
    const int oq = qorig[(size_t)bi * npq + p];
    // a point with a NaN or infinite coordinate is at a NaN or infinite distance from everything: nothing is ever inserted
    // (tf_interpolate.cpp:78-93: every comparison fails) and it takes no part in the search
    const bool search = oq >= 0 && isfinite(x1) && isfinite(y1) && isfinite(z1);
    // the lane's three best as 64-bit keys (distance bits, index): a squared distance is never negative, so its bit pattern orders
    // as the float does, a NaN above +inf; "smaller key" is then the scan's strict '<' in index order, ties included, in ONE
    // comparison.  Unfilled slots are (+inf, 0) as the op leaves them; a lane that does not search holds zeros -- nothing is
    // below them -- until the end.
    typedef unsigned long long u64;
    const u64 kInf = (u64)0x7f800000u << 32;
    u64 k1 = search ? kInf : 0ull, k2 = k1, k3 = k1;
    const float *__restrict__ CX = cxyz + (size_t)bi * npc * 3;
    const int *__restrict__ CO = corig + (size_t)bi * npc;
    const int nsb = npc >> 6;
    const float *__restrict__ B16 = cb16 + (size_t)bi * nsb * 24;
    const float *__restrict__ B64 = cb64 + (size_t)bi * nsb * 8;
#define TB_B3 __uint_as_float((unsigned)(k3 >> 32))

    // One candidate (uniform coordinates and original index, through scalar registers).  A padding record (index -1 = the largest
    // unsigned, coordinates +inf) never enters: its key is not below (+inf, 0).
#define TB_CONSIDER(cx, cy, cz, oi)                                                                         \
    {                                                                                                       \
        const float dx_ = (cx) - x1, dy_ = (cy) - y1, dz_ = (cz) - z1;                                       \
        const float xx_ = dx_ * dx_, yy_ = dy_ * dy_, zz_ = dz_ * dz_;                                       \
        const float d_ = (xx_ + yy_) + zz_;                                                                  \
        if (__ballot(d_ <= TB_B3) != 0ull) { /* wave-uniform */                                              \
            asm volatile("; some lane may insert");                                                          \
            const u64 key_ = ((u64)__float_as_uint(d_) << 32) | (u64)(unsigned)(oi);                         \
            const bool c3_ = key_ < k3, c2_ = key_ < k2, c1_ = key_ < k1;                                    \
            k3 = c3_ ? (c2_ ? k2 : key_) : k3;                                                               \
            k2 = c2_ ? (c1_ ? k1 : key_) : k2;                                                               \
            k1 = c1_ ? key_ : k1;                                                                            \
        }                                                                                                   \
    }
#ifdef TB_STATS
    int nblk = 0, nsbv = 0;  // (uniform) block scans, superblock visits
#endif
    // one superblock: per 16-record block the lanes' bounds against their third-best (TEST), then the records of the blocks
    // some lane needs, eight at a time through two scalar register sets in turn (the next eight are on their way while
    // these are compared)
    auto visit = [&](int sb, bool test) {
#ifdef TB_STATS
        nsbv++;
#endif
        unsigned hm = 0xFFu;  // the half-blocks to scan
        if (test) {
            const float *bx = B16 + (size_t)sb * 24;  // (uniform -> scalar loads)
            float bb[24];
#pragma unroll
            for (int i = 0; i < 24; i++) bb[i] = bx[i];
            hm = 0u;
#pragma unroll
            for (int blk = 0; blk < 4; blk++) {
                const float lb = tb_bound(bb[blk * 6], bb[blk * 6 + 1], bb[blk * 6 + 2], bb[blk * 6 + 3], bb[blk * 6 + 4],
                                          bb[blk * 6 + 5], x1, y1, z1, x1, y1, z1);
                if (__ballot(lb <= TB_B3) != 0ull) hm |= 3u << (2 * blk);  // (uniform)
            }
            if (hm == 0u) return;
        }
#ifdef TB_STATS
        nblk += __builtin_popcount(hm) >> 1;
#endif
        const float *cb = CX + (size_t)sb * 192;
        const int *ob = CO + sb * 64;
        float ca[24], cc[24];
        int oa[8], oc[8];
#define TB_FETCH(C, O, H)                                           \
    {                                                               \
        const float *cp_ = cb + (H) * 24;                           \
        const int *op_ = ob + (H) * 8;                              \
        _Pragma("unroll") for (int i = 0; i < 24; i++) C[i] = cp_[i]; \
        _Pragma("unroll") for (int i = 0; i < 8; i++) O[i] = op_[i];  \
    }
#define TB_SCAN8(C, O) _Pragma("unroll") for (int u = 0; u < 8; u++) TB_CONSIDER(C[u * 3], C[u * 3 + 1], C[u * 3 + 2], O[u])
        int h = __builtin_ctz(hm);
        hm &= hm - 1u;
        TB_FETCH(ca, oa, h);
        for (;;) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set a has arrived
            __builtin_amdgcn_sched_barrier(0);
            const bool more_b = hm != 0u;
            if (more_b) {
                h = __builtin_ctz(hm);
                hm &= hm - 1u;
                TB_FETCH(cc, oc, h);
            }
            __builtin_amdgcn_sched_barrier(0);
            TB_SCAN8(ca, oa);
            if (!more_b) break;
            __builtin_amdgcn_s_waitcnt(0xC07F);  // set c has arrived
            __builtin_amdgcn_sched_barrier(0);
            const bool more_a = hm != 0u;
            if (more_a) {
                h = __builtin_ctz(hm);
                hm &= hm - 1u;
                TB_FETCH(ca, oa, h);
            }
            __builtin_amdgcn_sched_barrier(0);
            TB_SCAN8(cc, oc);
            if (!more_a) break;
        }
#undef TB_SCAN8
#undef TB_FETCH
    };

    // the wave's own box (its 64 unknown points are one superblock of their sorted set); without a point to search for, no search
    const float *qb = qb64 + ((size_t)bi * (npq >> 6) + group) * 8;
    const float qlx = qb[0], qly = qb[1], qlz = qb[2], qhx = qb[4], qhy = qb[5], qhz = qb[6];
    if (__ballot(search) != 0ull) {
        // 1. lanes <-> candidate superblocks: the one nearest to the wave's box goes first and sets the third-bests
        float best = INFINITY;
        int arg = 0;
        for (int r0 = 0; r0 < nsb; r0 += 64) {
            const int g = r0 + lane;
            float lb = INFINITY;
            if (g < nsb) {
                const float4 lo = *(const float4 *)(B64 + (size_t)g * 8), hi = *(const float4 *)(B64 + (size_t)g * 8 + 4);
                lb = tb_bound(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, qlx, qly, qlz, qhx, qhy, qhz);
            }
            if (lb < best) best = lb, arg = g;
        }
        const float wmin = tb_wave_min(best);
        const unsigned long long at = __ballot(best == wmin);
        const int seed = at != 0ull ? __builtin_amdgcn_readlane(arg, __builtin_ctzll(at)) : 0;
        visit(seed, false);
        // 2. every other superblock whose box is not beyond the wave's largest third-best (which shrinks as the visits go),
        //    64 superblocks at a time
        float w3 = tb_wave_max(TB_B3);  // (a lane that does not search holds 0)
        for (int r0 = 0; r0 < nsb; r0 += 64) {
            const int g = r0 + lane;
            float lb = INFINITY;
            if (g < nsb && g != seed) {
                const float4 lo = *(const float4 *)(B64 + (size_t)g * 8), hi = *(const float4 *)(B64 + (size_t)g * 8 + 4);
                lb = tb_bound(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, qlx, qly, qlz, qhx, qhy, qhz);
            }
            // nearest box first: the third-bests shrink fastest that way, and the first box beyond the wave's largest ends the round
            bool pend = lb <= w3 && lb != INFINITY;
            while (__ballot(pend) != 0ull) {  // (uniform)
                const float wmin = tb_wave_min(pend ? lb : INFINITY);
                if (!(wmin <= w3)) break;
                const int j = __builtin_ctzll(__ballot(pend && lb == wmin));
                pend = pend && lane != j;
                visit(r0 + j, true);
                w3 = tb_wave_max(TB_B3);
            }
        }
    }
#undef TB_CONSIDER
#undef TB_B3
    if (oq >= 0) {
        if (!search) k1 = k2 = k3 = kInf;
        const size_t o = ((size_t)bi * n + oq) * 3;
        dist[o] = __uint_as_float((unsigned)(k1 >> 32));
        dist[o + 1] = __uint_as_float((unsigned)(k2 >> 32));
        dist[o + 2] = __uint_as_float((unsigned)(k3 >> 32));
        idx[o] = (int)(unsigned)k1;
#ifdef TB_STATS
        idx[o + 1] = nsbv, idx[o + 2] = nblk;
#else
        idx[o + 1] = (int)(unsigned)k2;
        idx[o + 2] = (int)(unsigned)k3;
#endif
    }
}

// ---- three_interpolate / its gradient, row-shaped --------------------------------------------------------------------------
// One element per thread (the kernels below) pays a 64-bit division per output element and six index / weight loads for four
// bytes of output: 1.7 TB/s forward, 0.5 TB/s backward at 32 x 16384 x 1024, c = 128.  Here a thread row owns an unknown point:
// its three indices and weights are loaded once, the channels go by as VEC-wide vectors over the row's lanes (32-bit
// arithmetic, no division).  Same expression per element: (p1*w1 + p2*w2) + p3*w3, every product rounded (-ffp-contract=off).
constexpr int TI_TPB = 256;
constexpr int TI_PP = 4;  // points per thread row and block

typedef float ti_v4f __attribute__((ext_vector_type(4)));
template <int VEC>
struct TiVec;
template <>
struct TiVec<4> {
    typedef ti_v4f T;
    static __device__ __forceinline__ T mul(T a, float w) { return a * w; }  // (per component, every product rounded)
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ float at(T a, int v) { return a[v]; }
};
template <>
struct TiVec<1> {
    typedef float T;
    static __device__ __forceinline__ T mul(T a, float w) { return a * w; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ float at(T a, int) { return a; }
};

template <int VEC>
__global__ __launch_bounds__(TI_TPB) void three_interpolate_rows_kernel(int m, int c, int n, int tx_log2, int bpb /* blocks per sample */,
                                                                        const float *__restrict__ points,
                                                                        const int *__restrict__ idx,
                                                                        const float *__restrict__ weight,
                                                                        float *__restrict__ out) {
    typedef typename TiVec<VEC>::T V;
    // (a sample's blocks on ONE XCD: its rows of `points` then cross the fabric once instead of eight times -- FETCH_SIZE 71 -> 14 MB, 68 -> 57 us)
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);  // a sample's workgroups on one XCD
    const int bi = logical / bpb, bx = logical - bi * bpb;
    const int TX = 1 << tx_log2, TY = TI_TPB >> tx_log2;
    const int lx = threadIdx.x & (TX - 1), ly = threadIdx.x >> tx_log2;
    const int cv = c / VEC;
    const V *__restrict__ P = (const V *)(points + (size_t)bi * m * c);
    V *__restrict__ O = (V *)(out + (size_t)bi * n * c);
    const int *__restrict__ I = idx + (size_t)bi * n * 3;
    const float *__restrict__ W = weight + (size_t)bi * n * 3;
    int r[TI_PP][3];
    float w[TI_PP][3];
    int jj[TI_PP];
#pragma unroll
    for (int u = 0; u < TI_PP; u++) {
        jj[u] = (bx * TI_PP + u) * TY + ly;
        const int j = min(jj[u], n - 1);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            r[u][t] = I[j * 3 + t] * cv;
            w[u][t] = W[j * 3 + t];
        }
    }
    for (int l = lx; l < cv; l += TX) {
        V a[TI_PP], b2[TI_PP], d[TI_PP];
#pragma unroll
        for (int u = 0; u < TI_PP; u++) a[u] = P[r[u][0] + l], b2[u] = P[r[u][1] + l], d[u] = P[r[u][2] + l];
#pragma unroll
        for (int u = 0; u < TI_PP; u++) {
            if (jj[u] < n)  // (written once, read by another kernel: past the caches)
                __builtin_nontemporal_store(TiVec<VEC>::add(TiVec<VEC>::add(TiVec<VEC>::mul(a[u], w[u][0]), TiVec<VEC>::mul(b2[u], w[u][1])),
                                                            TiVec<VEC>::mul(d[u], w[u][2])), &O[(size_t)jj[u] * cv + l]);
        }
    }
}

// The gradient: a workgroup owns a SLICE of cs channels of one sample's (m, c) gradient as an LDS tile of doubles -- ds_add_f64
// runs at 18 lane-operations per ns and CU against 0.8 for ds_add_f32 (tools/ubench/lds_atomic_rate.hip), and L2 atomics from
// every element (the kernel below) reach 0.3 per ns and CU -- and walks a part of the unknown points: grad_out is read once,
// cs * 4 bytes per point and workgroup; the tile leaves as plain stores (one part) or atomic adds (several).
#ifndef RFI_TG_WGS
#define RFI_TG_WGS 256
#endif
constexpr int TG_TPB = 1024;  // (128 KiB of LDS: one workgroup per CU -- its 16 waves are all the loads in flight there are)
constexpr int TG_U = 4;  // points per thread row in flight

template <int VEC, bool POW2>  // POW2: cs is a power of two (shifts instead of multiplications and a division)
__global__ __launch_bounds__(TG_TPB) void three_interpolate_grad_tile_kernel(int m, int c, int n, int cs, int tx_log2,
                                                                             int nslices, int parts,
                                                                             const float *__restrict__ grad_out,
                                                                             const int *__restrict__ idx,
                                                                             const float *__restrict__ weight,
                                                                             float *__restrict__ grad_points) {
    typedef typename TiVec<VEC>::T V;
    extern __shared__ __attribute__((aligned(16))) double ti_tile[];  // [m * cs]; cs = VEC << tx_log2, or (VEC == 1) any cs <= 1 << tx_log2
    // (a sample's slices and parts on ONE XCD: neighbouring slices share the cache lines of grad_out's rows)
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);  // a sample's workgroups on one XCD
    const int wps = nslices * parts;  // workgroups per sample
    const int bi = logical / wps, bx = logical - bi * wps;
    const int slice = bx % nslices, part = bx / nslices;
    const int TX = 1 << tx_log2, TY = TG_TPB >> tx_log2;
    const int lx = threadIdx.x & (TX - 1), ly = threadIdx.x >> tx_log2;
    const bool lane_on = POW2 || lx * VEC < cs;  // (a slice of 3 or 13 channels leaves the last lanes of its rows idle)
    const int csl = POW2 ? 31 - __clz(cs) : 0;
    for (int e = threadIdx.x; e < m * cs; e += TG_TPB) ti_tile[e] = 0.0;
    const int per = (n + parts - 1) / parts;
    const int jbeg = part * per, jend = min(n, jbeg + per);
    const float *__restrict__ G = grad_out + (size_t)bi * n * c + slice * cs + (lane_on ? lx * VEC : 0);
    const int *__restrict__ I = idx + (size_t)bi * n * 3;
    const float *__restrict__ W = weight + (size_t)bi * n * 3;
    __syncthreads();
    // a point's cs sums sit v-major in the tile (channel lx * VEC + v at v * TX + lx): the lanes of a row then add to consecutive
    // doubles in every instruction; the next batch of points is loaded before this one is added (two waves per SIMD at 128 KiB
    // of LDS: nothing else hides the loads)
    const int rot = VEC == 4 ? (ly & 3) : 0;  // this row's first channel of four
    int voff[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) voff[k] = ((k + rot) & (VEC - 1)) << tx_log2;
    V g[TG_U], gn[TG_U];
    int r[TG_U][3], rn[TG_U][3];
    float w[TG_U][3], wn[TG_U][3];
#define TG_LOAD(GG, RR, WW, J0)                                                   \
    _Pragma("unroll") for (int u = 0; u < TG_U; u++) {                            \
        const int j = min((J0) + u * TY, jend - 1);                               \
        GG[u] = *(const V *)(G + (size_t)j * c);                                  \
        _Pragma("unroll") for (int t = 0; t < 3; t++) {                           \
            RR[u][t] = (POW2 ? I[j * 3 + t] << csl : I[j * 3 + t] * cs) + lx;     \
            WW[u][t] = W[j * 3 + t];                                              \
        }                                                                         \
    }
    if (jbeg + ly < jend) {
        TG_LOAD(g, r, w, jbeg + ly)
    }
    for (int j0 = jbeg + ly; j0 < jend; j0 += TY * TG_U) {
        const bool more = j0 + TY * TG_U < jend;
        if (more) {
            TG_LOAD(gn, rn, wn, j0 + TY * TG_U)
        }
        // (neighbouring rows start at different channels: in one instruction the rows of a wave would otherwise all add to the
        // same 8 banks of their points' 32)
        float gr[TG_U][VEC];
#pragma unroll
        for (int u = 0; u < TG_U; u++)
#pragma unroll
            for (int k = 0; k < VEC; k++) {
                float x = TiVec<VEC>::at(g[u], k);
                if (VEC == 4) {
                    const float y = TiVec<VEC>::at(g[u], (k + 1) & 3), z = TiVec<VEC>::at(g[u], (k + 2) & 3), q = TiVec<VEC>::at(g[u], (k + 3) & 3);
                    x = rot == 0 ? x : (rot == 1 ? y : (rot == 2 ? z : q));
                }
                gr[u][k] = x;
            }
#pragma unroll
        for (int u = 0; u < TG_U; u++) {
            if (j0 + u * TY < jend && lane_on) {
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int k = 0; k < VEC; k++) atomicAdd(&ti_tile[r[u][t] + voff[k]], (double)(gr[u][k] * w[u][t]));
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < TG_U; u++) {
                g[u] = gn[u];
#pragma unroll
                for (int t = 0; t < 3; t++) r[u][t] = rn[u][t], w[u][t] = wn[u][t];
            }
        }
    }
#undef TG_LOAD
    __syncthreads();
    float *__restrict__ GP = grad_points + (size_t)bi * m * c + slice * cs;
    for (int e = threadIdx.x; e < m * cs; e += TG_TPB) {
        const int i = POW2 ? e >> csl : e / cs, ch = e - i * cs;
        const float v = (float)ti_tile[i * cs + (ch % VEC << tx_log2) + ch / VEC];
        if (parts == 1) {
            GP[(size_t)i * c + ch] = v;
        } else {
            atomicAdd(&GP[(size_t)i * c + ch], v);
        }
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

// ---- the boxed form: needs scratch (the sorted copies of the two sets unless the caller hands rf_nn_sort handles over)
size_t rf_threenn_boxes_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || b > 65535 || !rfp::pruned_supported(b, n, m)) return 0;
    return rfp::sorted_bytes(b, n) + rfp::sorted_bytes(b, m);
}

int rf_threenn_boxes(int b, int n, int m, const float *xyz1, const float *xyz2, const void *sorted1, const void *sorted2,
                     float *dist, int *idx, void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || n == 0) return RF_OK;
    if (b > 65535 || !rfp::pruned_supported(b, n, m)) return RF_EINVAL;  // (m = 0 and huge clouds: rf_threenn)
    if (!xyz1 || !xyz2 || !dist || !idx || !workspace || !rf::aligned16(workspace)) return RF_EINVAL;
    if ((sorted1 && !rf::aligned16(sorted1)) || (sorted2 && !rf::aligned16(sorted2))) return RF_EINVAL;
    if (workspace_bytes < rf_threenn_boxes_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    rfp::Sorted sv[2];
    sv[0] = rfp::sorted_view(b, n, sorted1 ? sorted1 : workspace);
    sv[1] = rfp::sorted_view(b, m, sorted2 ? sorted2 : (const char *)workspace + rfp::sorted_bytes(b, n));
    {  // the sets that came without a handle, in one launch
        int nn[2];
        const float *src[2];
        rfp::Sorted out[2];
        int k = 0;
        if (!sorted1) nn[k] = n, src[k] = xyz1, out[k] = sv[0], k++;
        if (!sorted2) nn[k] = m, src[k] = xyz2, out[k] = sv[1], k++;
        if (k > 0)
            if (int e = rfp::sort_sets(b, k, nn, src, out, s, nullptr)) return e;
    }
    RF_LAUNCH("three_nn_boxes", three_nn_boxes_kernel, dim3(rf::ceil_div(sv[0].npad / 64, TB_WAVES) * b), dim3(64 * TB_WAVES), 0, s,
              n, sv[0].npad, sv[1].npad, sv[0].xyz, sv[0].orig, sv[0].box64, sv[1].xyz, sv[1].orig, sv[1].box16, sv[1].box64,
              dist, idx);
    return RF_OK;
}

int rf_threeinterpolate(int b, int m, int c, int n, const float *points, const int *idx,
                        const float *weight, float *out, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || c < 0) return RF_EINVAL;
    long total = (long)b * n * c;
    if (total == 0) return RF_OK;
    if (!points || !idx || !weight || !out) return RF_EINVAL;
    if (b <= 65535 && (long)n * c < (1L << 31) && (long)m * c < (1L << 31) && (long)n * 3 < (1L << 31)) {
        const bool vec = c % 4 == 0 && rf::aligned16(points) && rf::aligned16(out);
        const int cv = vec ? c / 4 : c;
        int tx_log2 = 0;
        while ((1 << tx_log2) < cv && tx_log2 < 6) tx_log2++;
        const int ppb = (TI_TPB >> tx_log2) * TI_PP;  // points per block
        const long bpb = rf::ceil_div(n, ppb);
        if (bpb * b <= 0x7FFFFFFF) {
            const dim3 grid((unsigned)(bpb * b));
            if (vec) {
                RF_LAUNCH("three_interpolate", three_interpolate_rows_kernel<4>, grid, dim3(TI_TPB), 0, (hipStream_t)stream, m, c, n,
                          tx_log2, (int)bpb, points, idx, weight, out);
            } else {
                RF_LAUNCH("three_interpolate", three_interpolate_rows_kernel<1>, grid, dim3(TI_TPB), 0, (hipStream_t)stream, m, c, n,
                          tx_log2, (int)bpb, points, idx, weight, out);
            }
            return RF_OK;
        }
    }
    RF_LAUNCH("three_interpolate", three_interpolate_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, m, c, n, total, points, idx, weight, out);
    return RF_OK;
}

// The LDS-tile form's slice: cs channels with m * cs doubles in 128 KiB -- cs a power of two dividing c (8..64), or, for c that has
// none (3, 6, 13 ...), all c <= 64 channels in one slice; 0: the sample's known points do not fit a tile
static int tig_tile_cs(int b, int n, int c, int m) {
    if (!(b <= 65535 && (long)n * c < (1L << 31) && (long)m * c < (1L << 31) && (long)n * 3 < (1L << 31))) return 0;
    for (int k = 6; k >= 3; k--)
        if (c % (1 << k) == 0 && ((long)m << k) <= 16384) return 1 << k;
    if (c <= 64 && (long)m * c <= 16384) return c;
    return 0;
}
// beyond the tile (more than 2048 known points at 8 channels per slice): the sorted-slots form of scatter_rows.hip, from this
// many gradient elements on (below: the atomics)
constexpr long TIG_CSR_MIN_ELEMS = 1L << 22;
static bool tig_csr(int b, int n, int c, int m) {
    return tig_tile_cs(b, n, c, m) == 0 && (long)b * n * 3 * c >= TIG_CSR_MIN_ELEMS && rfs::rows_csr_supported(b, m, c, (long)n * 3, 3);
}

size_t rf_threeinterpolate_grad_workspace_bytes(int b, int n, int c, int m) {
    if (b <= 0 || n <= 0 || c <= 0 || m <= 0) return 0;
    return tig_csr(b, n, c, m) ? rfs::rows_csr_workspace_bytes(b, m, (long)n * 3) : 0;
}

int rf_threeinterpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                             const float *weight, float *grad_points, rf_stream_t stream) {
    return rf_threeinterpolate_grad_ws(b, n, c, m, grad_out, idx, weight, grad_points, nullptr, 0, stream);
}

int rf_threeinterpolate_grad_ws(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points, void *workspace, size_t workspace_bytes,
                                rf_stream_t stream) {

    if (b < 0 || n < 0 || m < 0 || c < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * m * c && !grad_points) return RF_EINVAL;
    long total = (long)b * n * c;
    if (total != 0 && m != 0 && (!grad_out || !idx || !weight)) return RF_EINVAL;
    const int cs = (total != 0 && m != 0) ? tig_tile_cs(b, n, c, m) : 0;
    if (cs == 0 && total != 0 && m != 0 && tig_csr(b, n, c, m) && workspace && rf::aligned16(workspace) &&
        workspace_bytes >= rf_threeinterpolate_grad_workspace_bytes(b, n, c, m))
        return rfs::rows_csr_scatter(b, m, c, (long)n * 3, 3, grad_out, idx, weight, grad_points, workspace, "three_interpolate_grad_sort",
                                     "three_interpolate_grad", s);
    if (cs > 0) {
        const bool pow2 = (cs & (cs - 1)) == 0 && cs >= 8;
        const bool vec = pow2 && rf::aligned16(grad_out);  // (rows of a slice start 16-byte aligned when the tensor does)
        const int nslices = c / cs;
        int tx_log2 = 0;
        while ((1 << tx_log2) < (vec ? cs / 4 : cs)) tx_log2++;
        // parts of the unknown points: a workgroup per CU (the tile leaves room for one), every part still thousands of points
        int parts = 1;
        while ((long)b * nslices * parts < RFI_TG_WGS && n / (parts * 2) >= 2048) parts *= 2;
        if (parts > 1) RF_ZERO(grad_points, sizeof(float) * (size_t)b * m * c, s);
        const size_t lds = sizeof(double) * (size_t)m * cs;
        const dim3 grid((unsigned)(nslices * parts * b));
#define TG_GO(KERNEL)                                                                                                      \
    RF_HIP(hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));                 \
    RF_LAUNCH("three_interpolate_grad", KERNEL, grid, dim3(TG_TPB), lds, s, m, c, n, cs, tx_log2, nslices, parts, grad_out, idx, \
              weight, grad_points)
        if (vec) {
            TG_GO((three_interpolate_grad_tile_kernel<4, true>));
        } else if (pow2) {
            TG_GO((three_interpolate_grad_tile_kernel<1, true>));
        } else {
            TG_GO((three_interpolate_grad_tile_kernel<1, false>));
        }
#undef TG_GO
        return RF_OK;
    }
    if ((size_t)b * m * c) RF_ZERO(grad_points, sizeof(float) * (size_t)b * m * c, s);
    if (total == 0 || m == 0) return RF_OK;
    RF_LAUNCH("three_interpolate_grad", three_interpolate_grad_kernel, dim3(rf::ceil_div(total, 256)),
              dim3(256), 0, s, m, c, n, total, grad_out, idx, weight, grad_points);
    return RF_OK;
}

}  // extern "C"
