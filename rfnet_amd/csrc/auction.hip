// auction.hip -- auction_match (tf_ops/emd) and select_top_k / SelectionSort (tf_ops/grouping):
// the two remaining ops of the reference's import-time surface ("next" row f3).
// Neither is called by the RFNet model (dead code there); they are built for completeness, with
// the reference's exact tie behaviour, not tuned.
//
// auction_match replaces AuctionMatchKernel (tf_ops/emd/tf_auctionmatch_g.cu:2-294): a
// sequential auction, one 512-thread workgroup per batch element, cost matrix in the caller's
// scratch (b,n,n).  The assignment depends on how ties fall in the block-wide (best, second,
// argmin) reduction, which the reference writes for 32-lane warps; it is reproduced here with
// width-32 shuffles inside wave64 (two 32-lane halves reduce independently, exactly like two
// warps) followed by the same 16-entry tree, including the reference's quirk that the shuffle
// step overwrites `best` before taking fminf(best, b2) (:222-229).  Bit-exact with
// oracle/rfops_oracle.c::orc_auction_match.  Defined only for n < 1024 or n in {1024, 2048,
// 4096}: for other n the reference's strided scans read out of bounds (:148,185).
//
// select_top_k replaces selection_sort_gpu (tf_grouping_g.cu:83-123): one wave per row, the row
// in LDS, wave arg-min with the lowest index on ties (= the reference's strict '<' scan), swaps
// by lane 0; bit-exact incl. the positions beyond k.
#include "common.hpp"

namespace {

constexpr int AT = 512;

struct Bid {
    float best, best2;
    int bestj;
};

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
// one shuffle-down step inside a 32-lane segment (tf_auctionmatch_g.cu:219-230, order kept)
__device__ __forceinline__ void bid_shfl(Bid &r, int off) {
    const float b1 = __shfl_down(r.best, off, 32);
    const float b2 = __shfl_down(r.best2, off, 32);
    const int bj = __shfl_down(r.bestj, off, 32);
    if (r.best < b1) { r.best2 = fminf(b1, r.best2); }
    else { r.best = b1; r.best2 = fminf(r.best, b2); r.bestj = bj; }
}

__global__ __launch_bounds__(AT) void auction_kernel(int n, const float *__restrict__ xyz1,
                                                     const float *__restrict__ xyz2,
                                                     int *__restrict__ matchl, int *__restrict__ matchr,
                                                     float *__restrict__ cost_all) {
    __shared__ short queue[4096];
    __shared__ short matchrbuf[4096];
    __shared__ float pricer[4096];
    __shared__ float bests[AT / 32][3];
    __shared__ int qhead, qlen;
    const int bi = blockIdx.x, t = threadIdx.x;
    const float *A = xyz1 + (size_t)bi * n * 3, *B = xyz2 + (size_t)bi * n * 3;
    float *cost = cost_all + (size_t)bi * n * n;
    int *ML = matchl + (size_t)bi * n, *MR = matchr + (size_t)bi * n;
    for (int j = t; j < n; j += AT) {
        ML[j] = -1;
        matchrbuf[j] = -1;
        queue[j] = (short)j;
        pricer[j] = 0.f;
    }
    for (int j = t; j < n; j += AT) {  // cost[k][j] = |xyz1_k - xyz2_j|
        const float x2 = B[j * 3], y2 = B[j * 3 + 1], z2 = B[j * 3 + 2];
        for (int k = 0; k < n; k++)
            cost[(size_t)k * n + j] = sqrtf(rf::d2_fma(A[k * 3] - x2, A[k * 3 + 1] - y2, A[k * 3 + 2] - z2));
    }
    if (t == 0) { qhead = 0; qlen = n; }
    __syncthreads();
    int cnt = 0;               // thread 0 only
    float tolerance = 1e-4f;   // thread 0 only
    while (qlen) {
        const int i = queue[qhead];
        const float *row = cost + (size_t)i * n;
        Bid r{1e38f, 1e38f, 0};
        if (n == AT * 8) {
            float lo[4], hi[4];
            int jl[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int j1 = t + AT * 2 * p, j2 = j1 + AT;
                bid_pair(row[j1] + pricer[j1], j1, row[j2] + pricer[j2], j2, lo[p], jl[p], hi[p]);
            }
            float qlo, qhi, rlo, rhi;
            int qj, rj;
            bid_merge(lo[0], jl[0], hi[0], lo[1], jl[1], hi[1], qlo, qj, qhi);
            bid_merge(lo[2], jl[2], hi[2], lo[3], jl[3], hi[3], rlo, rj, rhi);
            bid_merge(qlo, qj, qhi, rlo, rj, rhi, r.best, r.bestj, r.best2);
        } else if (n >= AT * 4) {
            for (int j = t; j < n; j += AT * 4) {
                float l0, h0, l1, h1, ql, qh;
                int j0, j1, qj;
                bid_pair(row[j] + pricer[j], j, row[j + AT] + pricer[j + AT], j + AT, l0, j0, h0);
                bid_pair(row[j + 2 * AT] + pricer[j + 2 * AT], j + 2 * AT, row[j + 3 * AT] + pricer[j + 3 * AT],
                         j + 3 * AT, l1, j1, h1);
                bid_merge(l0, j0, h0, l1, j1, h1, ql, qj, qh);
                bid_acc(r, ql, qj, qh);
            }
        } else if (n >= AT * 2) {
            for (int j = t; j < n; j += AT * 2) {
                float l0, h0;
                int j0;
                bid_pair(row[j] + pricer[j], j, row[j + AT] + pricer[j + AT], j + AT, l0, j0, h0);
                bid_acc(r, l0, j0, h0);
            }
        } else {
            for (int j = t; j < n; j += AT) {
                const float v = row[j] + pricer[j];
                if (r.best < v) { r.best2 = fminf(r.best2, v); }
                else { r.best2 = r.best; r.bestj = j; r.best = v; }
            }
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) bid_shfl(r, off);
        if ((t & 31) == 0) {
            bests[t >> 5][0] = r.best;
            bests[t >> 5][1] = r.best2;
            bests[t >> 5][2] = __int_as_float(r.bestj);
        }
        __syncthreads();
        if (t < 64) {  // wave 0: lanes 0..15 carry the 16 partial results, the rest neutral
            Bid w{1e38f, 1e38f, 0};
            if (t < AT / 32) {
                w.best = bests[t][0];
                w.best2 = bests[t][1];
                w.bestj = __float_as_int(bests[t][2]);
            }
#pragma unroll
            for (int off = (AT / 32) >> 1; off > 0; off >>= 1) {
                // lanes whose partner lies beyond the 16 entries keep their value (only lane 0's
                // result is used, and it only ever meets valid partners)
                Bid nw = w;
                bid_shfl(nw, off);
                w = (t + off < AT / 32) ? nw : w;
            }
            if (t == 0) {
                const float delta = w.best2 - w.best + tolerance;
                int h = qhead + 1, ql = qlen - 1;
                if (h >= n) h -= n;
                const int old = matchrbuf[w.bestj];
                pricer[w.bestj] += delta;
                cnt++;
                if (old != -1) {
                    int tail = h + ql;
                    ql = ql + 1;
                    if (tail >= n) tail -= n;
                    queue[tail] = (short)old;
                }
                if (cnt == 40 * n) {
                    if (tolerance == 1.0f) ql = 0;
                    tolerance = fminf(1.0f, tolerance * 100);
                    cnt = 0;
                }
                matchrbuf[w.bestj] = (short)i;
                qhead = h;
                qlen = ql;
            }
        }
        __syncthreads();
    }
    for (int j = t; j < n; j += AT) MR[j] = matchrbuf[j];
    __syncthreads();
    for (int j = t; j < n; j += AT) ML[matchrbuf[j]] = j;
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

size_t rf_auctionmatch_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    return (size_t)b * n * n * sizeof(float);
}

int rf_auctionmatch(int b, int n, const float *xyz1, const float *xyz2, int *matchl, int *matchr,
                    void *workspace, size_t workspace_bytes, rf_stream_t stream) {
    if (b < 0 || n < 0) return RF_EINVAL;
    if (b == 0 || n == 0) return RF_OK;
    if (!rf_auctionmatch_supported(n)) return RF_EINVAL;
    if (!xyz1 || !xyz2 || !matchl || !matchr || !workspace) return RF_EINVAL;
    if (workspace_bytes < rf_auctionmatch_workspace_bytes(b, n)) return RF_EWORKSPACE;
    RF_LAUNCH("auction_match", auction_kernel, dim3(b), dim3(AT), 0, (hipStream_t)stream, n, xyz1, xyz2,
              matchl, matchr, (float *)workspace);
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
