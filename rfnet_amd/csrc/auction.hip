// auction.hip -- auction_match (tf_ops/emd) and select_top_k / SelectionSort (tf_ops/grouping):
// the two remaining ops of the reference's import-time surface ("next" row f3).
// Neither is called by the RFNet model (dead code there); built for completeness with the
// reference's exact tie behaviour.
//
// auction_match replaces AuctionMatchKernel (tf_ops/emd/tf_auctionmatch_g.cu:2-294): a sequential
// auction, one bidder per iteration.  WHAT has to be kept is the result: which object a bidder
// takes and at what price depends on how the (best, second best, argmin) of the bidder's values
// v_j = |xyz1_i - xyz2_j| + price_j is reduced over 512 strided partial scans -- the reference's
// pairwise step is neither associative nor commutative (when the upper partner wins, the lower
// one's best is dropped as a second-best candidate, :222-229), so the reduction TREE is part of
// the semantics: offsets 16,8,4,2,1 inside groups of 32 partials, then 8,4,2,1 over the 16 group
// results.  HOW it is computed here is built for gfx950:
//   * no cost matrix.  The reference fills a (n, n) matrix in global scratch (64 MiB per cloud at
//     n = 4096) and streams one row per iteration.  Here thread t keeps "its" objects j = t + 512 s
//     in registers -- coordinates, price, current owner -- and evaluates the row on the fly
//     (8 correctly rounded sqrt per thread per iteration); the bidder's point arrives by scalar
//     load.  Prices and owners are only ever read by the thread that owns the object, so the
//     winner's update is one predicated register write: no LDS arrays, no global traffic at all
//     inside the loop.
//   * the tree on wave64: the 32-partial groups are the two halves of a wave; offset 16 is one
//     ds_swizzle (xor 16), offsets 8..1 are DPP row rotations; the two group results of a wave go
//     to a double-buffered LDS slot -- ONE barrier per iteration -- and then EVERY wave reduces the
//     16 group results itself (DPP inside one row) and reads the winner back with readfirstlane.
//   * all bookkeeping (queue head / length, the 40 n counter, the tolerance schedule, the price
//     increment) is wave-uniform and replicated in every wave's scalar registers: nobody waits for
//     "thread 0" behind a second barrier.  The queue itself lives with wave 0, which publishes the
//     next bidder through the same exchange slot; the winner's previous owner travels through the
//     reduction as a fourth payload, so no wave ever has to look it up.
// Bit-exact with oracle/rfops_oracle.c::orc_auction_match.  Defined only for n < 1024 or n in
// {1024, 2048, 4096}: for other n the reference's strided scans read out of bounds (:148,185).
//
// select_top_k replaces selection_sort_gpu (tf_grouping_g.cu:83-123): one wave per row, the row
// in LDS, wave arg-min with the lowest index on ties (= the reference's strict '<' scan), swaps
// by lane 0; bit-exact incl. the positions beyond k.
#include "common.hpp"

namespace {

constexpr int AT = 512;        // partial scans per bidder (the reference's block size: fixes the tree)
constexpr int AK = 8;          // objects per thread (n <= 4096)

struct Bid {
    float best, best2;
    int bestj;
    int owner;  // current owner of object bestj (-1: free)
};

// the per-thread part: two values -> (lower, its index, higher); two such triples -> one; fold
// into the running result (strict '<' everywhere: on a tie the SECOND operand wins)
__device__ __forceinline__ void bid_pair(float v1, int j1, float v2, int j2, float &lo, int &jlo, float &hi) {
    if (v1 < v2) { lo = v1; jlo = j1; hi = v2; } else { lo = v2; jlo = j2; hi = v1; }
}
__device__ __forceinline__ void bid_merge(float alo, int aj, float ahi, float blo, int bj, float bhi,
                                          float &lo, int &jlo, float &hi) {
    if (alo < blo) { lo = alo; jlo = aj; hi = fminf(ahi, blo); }
    else           { lo = blo; jlo = bj; hi = fminf(alo, bhi); }
}
__device__ __forceinline__ void bid_acc(Bid &r, float lo, int jlo, float hi) {
    if (r.best < lo) { r.best2 = fminf(r.best2, lo); }
    else { r.best2 = fminf(r.best, hi); r.best = lo; r.bestj = jlo; }
}

// One step of the tree: `lo` is the partial with the lower position, `hi` its partner `off` above.
// The lower one keeps the lead only on a strictly smaller value; when the upper one takes over,
// the second best is formed from the UPPER partial alone (the reference assigns best = b1 before
// it evaluates fminf(best, b2)): kept, it decides prices.
__device__ __forceinline__ Bid bid_combine(const Bid &lo, const Bid &hi) {
    Bid r;
    if (lo.best < hi.best) {
        r = lo;
        r.best2 = fminf(hi.best, lo.best2);
    } else {
        r = hi;
        r.best2 = fminf(hi.best, hi.best2);
    }
    return r;
}

template <int CTRL>
__device__ __forceinline__ Bid bid_dpp(const Bid &v) {  // every field from the lane CTRL names
    Bid p;
    p.best = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.best), CTRL, 0xf, 0xf, false));
    p.best2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v.best2), CTRL, 0xf, 0xf, false));
    p.bestj = __builtin_amdgcn_update_dpp(0, v.bestj, CTRL, 0xf, 0xf, false);
    p.owner = __builtin_amdgcn_update_dpp(0, v.owner, CTRL, 0xf, 0xf, false);
    return p;
}
constexpr int ROR8 = 0x128, ROR4 = 0x124, ROR2 = 0x122, ROR1 = 0x121;  // row_ror:N -- lane l reads l+N (mod 16)

// offsets 8,4,2,1 inside every row of 16 lanes: lane 0 of the row ends with the tree's result
__device__ __forceinline__ Bid bid_tree16(Bid v) {
    v = bid_combine(v, bid_dpp<ROR8>(v));
    v = bid_combine(v, bid_dpp<ROR4>(v));
    v = bid_combine(v, bid_dpp<ROR2>(v));
    v = bid_combine(v, bid_dpp<ROR1>(v));
    return v;
}

struct Exchange {  // one iteration's hand-over between the waves (double buffered)
    float best[AT / 32], best2[AT / 32];
    int bestj[AT / 32], owner[AT / 32];
    int next_bidder;
};

__global__ __launch_bounds__(AT) void auction_kernel(int n, const float *__restrict__ xyz1,
                                                     const float *__restrict__ xyz2,
                                                     int *__restrict__ matchl, int *__restrict__ matchr) {
    __shared__ short queue[4096];  // read and written by wave 0 only
    __shared__ Exchange xch[2];
    const int bi = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ A = xyz1 + (size_t)bi * n * 3;
    const float *__restrict__ B = xyz2 + (size_t)bi * n * 3;
    int *ML = matchl + (size_t)bi * n, *MR = matchr + (size_t)bi * n;

    // this thread's objects: j = t + 512 s
    float ox[AK], oy[AK], oz[AK], price[AK];
    int owner[AK];
#pragma unroll
    for (int s = 0; s < AK; s++) {
        const int j = t + AT * s;
        const int jj = j < n ? j : 0;
        ox[s] = B[jj * 3 + 0]; oy[s] = B[jj * 3 + 1]; oz[s] = B[jj * 3 + 2];
        price[s] = 0.f;
        owner[s] = -1;
        if (j < n) ML[j] = -1;
    }
    for (int j = t; j < n; j += AT) queue[j] = (short)j;
    __syncthreads();

    // wave-uniform state, identical in every wave
    int qhead = 0, qlen = n, cnt = 0, bidder = 0;
    float tolerance = 1e-4f;
    int it = 0;
    while (qlen) {
        const float ax = A[bidder * 3 + 0], ay = A[bidder * 3 + 1], az = A[bidder * 3 + 2];  // uniform: scalar loads
        float v[AK];
#pragma unroll
        for (int s = 0; s < AK; s++)
            v[s] = sqrtf(rf::d2_fma(ax - ox[s], ay - oy[s], az - oz[s])) + price[s];
        Bid r{1e38f, 1e38f, 0, -1};
        if (n == AT * 8) {
            float lo[4], hi[4];
            int jl[4];
#pragma unroll
            for (int p = 0; p < 4; p++)
                bid_pair(v[2 * p], t + AT * 2 * p, v[2 * p + 1], t + AT * (2 * p + 1), lo[p], jl[p], hi[p]);
            float qlo, qhi, rlo, rhi;
            int qj, rj;
            bid_merge(lo[0], jl[0], hi[0], lo[1], jl[1], hi[1], qlo, qj, qhi);
            bid_merge(lo[2], jl[2], hi[2], lo[3], jl[3], hi[3], rlo, rj, rhi);
            bid_merge(qlo, qj, qhi, rlo, rj, rhi, r.best, r.bestj, r.best2);
        } else if (n >= AT * 4) {  // n == 2048: one quad per thread
            float l0, h0, l1, h1, ql, qh;
            int j0, j1, qj;
            bid_pair(v[0], t, v[1], t + AT, l0, j0, h0);
            bid_pair(v[2], t + 2 * AT, v[3], t + 3 * AT, l1, j1, h1);
            bid_merge(l0, j0, h0, l1, j1, h1, ql, qj, qh);
            bid_acc(r, ql, qj, qh);
        } else if (n >= AT * 2) {  // n == 1024: one pair per thread
            float l0, h0;
            int j0;
            bid_pair(v[0], t, v[1], t + AT, l0, j0, h0);
            bid_acc(r, l0, j0, h0);
        } else {  // n < 1024: at most two objects, taken one by one (a tie goes to the later one)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (t + AT * s < n) {
                    if (r.best < v[s]) { r.best2 = fminf(r.best2, v[s]); }
                    else { r.best2 = r.best; r.bestj = t + AT * s; r.best = v[s]; }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < AK; s++)
            if (r.bestj == t + AT * s) r.owner = owner[s];

        // groups of 32 partials = the halves of a wave: offset 16 by swizzle (lane ^ 16), then the row tree
        {
            Bid p;
            p.best = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(r.best), 0x401F));
            p.best2 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(r.best2), 0x401F));
            p.bestj = __builtin_amdgcn_ds_swizzle(r.bestj, 0x401F);
            p.owner = __builtin_amdgcn_ds_swizzle(r.owner, 0x401F);
            r = bid_combine(r, p);  // meaningful in lanes 0..15 of each half, the only ones read below
        }
        r = bid_tree16(r);
        Exchange &x = xch[it & 1];
        if ((lane & 31) == 0) {
            const int g = t >> 5;
            x.best[g] = r.best; x.best2[g] = r.best2; x.bestj[g] = r.bestj; x.owner[g] = r.owner;
        }
        if (t == 0) x.next_bidder = queue[qhead + 1 < n ? qhead + 1 : 0];
        __syncthreads();
        // every wave: the 16 group results in lanes 0..15, one more row tree, winner to scalars
        Bid w{1e38f, 1e38f, 0, -1};
        if (lane < AT / 32) { w.best = x.best[lane]; w.best2 = x.best2[lane]; w.bestj = x.bestj[lane]; w.owner = x.owner[lane]; }
        w = bid_tree16(w);
        const float best = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.best)));
        const float best2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.best2)));
        const int bestj = __builtin_amdgcn_readfirstlane(w.bestj);
        const int old = __builtin_amdgcn_readfirstlane(w.owner);
        const int published = __builtin_amdgcn_readfirstlane(x.next_bidder);

        const float delta = best2 - best + tolerance;
        const bool was_last = qlen == 1;
        qhead = qhead + 1 >= n ? qhead + 1 - n : qhead + 1;
        qlen--;
        cnt++;
        if (old != -1) {  // the previous owner goes back to the end of the queue
            int tail = qhead + qlen;
            if (tail >= n) tail -= n;
            qlen++;
            if (wave == 0 && lane == 0) queue[tail] = (short)old;
        }
        if (cnt == 40 * n) {  // no convergence at this tolerance: loosen it, or give up at 1.0
            if (tolerance == 1.0f) qlen = 0;
            tolerance = fminf(1.0f, tolerance * 100);
            cnt = 0;
        }
        // the object's owner thread books the sale
        if ((bestj & (AT - 1)) == t) {
#pragma unroll
            for (int s = 0; s < AK; s++)
                if ((bestj >> 9) == s) { price[s] += delta; owner[s] = bidder; }
        }
        bidder = was_last ? old : published;  // (irrelevant when the queue ran empty)
        it++;
    }
#pragma unroll
    for (int s = 0; s < AK; s++)
        if (t + AT * s < n) MR[t + AT * s] = owner[s];
    __syncthreads();  // the -1 fill of matchl above is complete and visible
#pragma unroll
    for (int s = 0; s < AK; s++)
        if (t + AT * s < n && owner[s] >= 0) ML[owner[s]] = t + AT * s;
}

// ---- select_top_k: one wave per row, row values in LDS
constexpr int SS_MAXN = 16384;
__global__ __launch_bounds__(64) void selection_sort_kernel(int n, int k, long nrows,
                                                            const float *__restrict__ dist,
                                                            int *__restrict__ outi, float *__restrict__ out) {
    __shared__ float rowv[SS_MAXN];
    const long r = blockIdx.x;
    const int lane = threadIdx.x;
    const float *src = dist + r * n;
    int *pi = outi + r * n;
    float *po = out + r * n;
    for (int s = lane; s < n; s += 64) {
        rowv[s] = src[s];
        pi[s] = s;
    }
    __syncthreads();
    const int kk = k < n ? k : n;
    for (int s = 0; s < kk; s++) {
        // first minimum of rowv[s..n): lowest index on ties, like `if (p[t] < p[min]) min = t`
        float bv = INFINITY;
        int bt = 0x7FFFFFFF;
        for (int t = s + lane; t < n; t += 64) {
            const float v = rowv[t];
            if (v < bv) { bv = v; bt = t; }
        }
        // a lane that saw only +inf/NaN keeps bt = INT_MAX; give every lane a valid candidate
        if (bt == 0x7FFFFFFF && s + lane < n) { bv = rowv[s + lane]; bt = s + lane; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int ot = __shfl_xor(bt, off, 64);
            if (ov < bv || (ov == bv && ot < bt) || (bt == 0x7FFFFFFF)) { bv = ov; bt = ot; }
        }
        // the sequential scan starts from min = s and only moves on a strictly smaller value
        int mn = s;
        if (bt != 0x7FFFFFFF && bv < rowv[s]) mn = bt;
        __syncthreads();
        if (lane == 0 && mn != s) {
            const float tv = rowv[mn]; rowv[mn] = rowv[s]; rowv[s] = tv;
            const int ti = pi[mn]; pi[mn] = pi[s]; pi[s] = ti;
        }
        __syncthreads();
    }
    for (int s = lane; s < n; s += 64) po[s] = rowv[s];
}

}  // namespace

extern "C" {

int rf_auctionmatch_supported(int n) { return n > 0 && (n < 1024 || n == 1024 || n == 2048 || n == 4096); }

// The reference's temp tensor is the (b, n, n) cost matrix; this implementation keeps every value it
// needs in registers and needs no scratch: 0 bytes, `workspace` may be NULL.
size_t rf_auctionmatch_workspace_bytes(int b, int n) {
    (void)b; (void)n;
    return 0;
}

int rf_auctionmatch(int b, int n, const float *xyz1, const float *xyz2, int *matchl, int *matchr,
                    void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0) return RF_EINVAL;
    if (b == 0 || n == 0) return RF_OK;
    if (!rf_auctionmatch_supported(n)) return RF_EINVAL;
    if (!xyz1 || !xyz2 || !matchl || !matchr) return RF_EINVAL;
    (void)workspace; (void)workspace_bytes;
    RF_LAUNCH("auction_match", auction_kernel, dim3(b), dim3(AT), 0, (hipStream_t)stream, n, xyz1, xyz2,
              matchl, matchr);
    return RF_OK;
}

int rf_selectionsort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                     rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0 || k <= 0) return RF_EINVAL;
    const long nrows = (long)b * m;
    if (nrows == 0 || n == 0) return RF_OK;
    if (n > SS_MAXN) return RF_EINVAL;
    if (!dist || !outi || !out) return RF_EINVAL;
    RF_LAUNCH("selection_sort", selection_sort_kernel, dim3((unsigned)nrows), dim3(64), 0, (hipStream_t)stream,
              n, k, nrows, dist, outi, out);
    return RF_OK;
}

}  // extern "C"
