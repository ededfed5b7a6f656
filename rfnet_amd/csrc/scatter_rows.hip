// scatter_rows.hip -- the scatter-add gradients of group_point (tf_ops/grouping/tf_grouping_g.cu:61-78) and three_interpolate
// (tf_ops/interpolation/tf_interpolate.cpp:131-153) WITHOUT atomics on memory, for the sizes where the reference's form
// (one atomicAdd per element into a zeroed tensor) is bound by the L2's atomic rate -- 0.29 float atomics per ns on the whole
// chip (profiles/r05_group_point_rate.txt): 0.235 ms for 32 x (1024 x 32 slots) x 64 channels whose data is 400 MB, 50 us of HBM.
//
// Both gradients are   dst[b][idx[b][s]][:] += w[b][s] * src[b][s / K][:]   over the slots s of a sample (K = 1, w = 1: group_point;
// K = 3: three_interpolate).  Turned round: every destination row SUMS the source rows of the slots that name it.
//   rows_csr_build   per sample, the slots in destination order (a counting sort by idx in LDS: integer ds_add at 24 lanes per
//                    ns and CU, against 0.8 for floats): row_start (n + 1) and perm (S).  A sample's key space is cut over H
//                    workgroups, each scanning all the slots and keeping its own range (no exchange: the slots below a range
//                    are counted on the way) -- 128 instead of 32 CUs busy at C3's shapes.
//   rows_csr_gather  a lane row per destination row: the row's slots one after the other, channels as float4 across the lanes,
//                    sums in DOUBLE (the loads bound the kernel; the result is then the same whatever order the sort left
//                    the slots in, to the last bit in all but ties of rounding), every destination row written ONCE --
//                    rows nobody names as zeros: no zero-fill pass, no read-modify-write.
// HBM traffic = the data: src once, dst once, 12 bytes of index per slot.
#include "common.hpp"
#include "scatter_rows.hpp"

namespace {

constexpr int CB_TPB = 1024;
constexpr int CB_MAXBINS = 20480;  // counters per workgroup (80 KiB of LDS; + 64 KiB of staged slot numbers: 144 of the CU's 160)

__device__ __forceinline__ unsigned cb_wave_incl_scan(unsigned v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(v, o, 64);
        if ((int)(threadIdx.x & 63) >= o) v += t;
    }
    return v;
}

// idx (b, S) -> row_start (b, n + 1), perm (b, S): perm[row_start[r] .. row_start[r + 1]) = the slots whose idx is r (any order).
// Slots whose idx is outside [0, n) are in no row (the reference would write outside its tensor for them).
// REG (CB_RPT > 0): S <= CB_TPB * CB_RPT -- a thread's slots are loaded ONCE, all loads in flight together, and both passes (counts, places)
// run from registers; otherwise each pass walks idx again (L2-resident), eight loads in flight.
constexpr int CB_STAGE = 16384;  // slot numbers staged in LDS (REG)
template <int CB_RPT>
__global__ __launch_bounds__(CB_TPB) void rows_csr_build_kernel(int n, int S, int H, int nbl, const int *__restrict__ idx,
                                                                int *__restrict__ row_start, int *__restrict__ perm) {
    extern __shared__ unsigned cb_cnt[];  // [nbl] counters, then [CB_STAGE] staged slot numbers
    __shared__ unsigned wsum[CB_TPB / 64];
    __shared__ unsigned lowsum[CB_TPB / 64];
    __shared__ unsigned seg_end;  // where this workgroup's segment of perm ends
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);
    const int bi = logical / H, h = logical - bi * H;
    const int lo = h * nbl, hi = min(n, lo + nbl);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool REG = CB_RPT > 0;
    const int *__restrict__ I = idx + (size_t)bi * S;
    int kr[REG ? CB_RPT : 1];
    if (REG) {
#pragma unroll
        for (int u = 0; u < CB_RPT; u++) kr[u] = tid + u * CB_TPB < S ? I[tid + u * CB_TPB] : -1;
    }
    for (int i = tid; i < nbl; i += CB_TPB) cb_cnt[i] = 0u;
    __syncthreads();
    unsigned low = 0;
    if (REG) {
#pragma unroll
        for (int u = 0; u < CB_RPT; u++) {
            if (kr[u] >= lo && kr[u] < hi) atomicAdd(&cb_cnt[kr[u] - lo], 1u);
            low += (kr[u] >= 0 && kr[u] < lo) ? 1u : 0u;
        }
    } else {
        for (int s0 = tid; s0 < S; s0 += 8 * CB_TPB) {
            int k[8];
#pragma unroll
            for (int u = 0; u < 8; u++) k[u] = s0 + u * CB_TPB < S ? I[s0 + u * CB_TPB] : -1;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (k[u] >= lo && k[u] < hi) atomicAdd(&cb_cnt[k[u] - lo], 1u);
                low += (k[u] >= 0 && k[u] < lo) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) low += __shfl_xor(low, o, 64);
    if (lane == 0) lowsum[wave] = low;
    __syncthreads();
    // exclusive scan of the nbl counters: thread t owns the `per` consecutive bins [t * per, (t + 1) * per)
    const int per = (nbl + CB_TPB - 1) / CB_TPB;
    unsigned mine = 0;
    for (int q = 0; q < per; q++) {
        const int i = tid * per + q;
        mine += i < nbl ? cb_cnt[i] : 0u;
    }
    const unsigned incl = cb_wave_incl_scan(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < CB_TPB / 64; w++) {
        base += lowsum[w];
        if (w < wave) base += wsum[w];
    }
    unsigned run = base + incl - mine;
    int *__restrict__ RS = row_start + (size_t)bi * (n + 1);
    for (int q = 0; q < per; q++) {
        const int i = tid * per + q;
        if (i < nbl) {
            const unsigned c = cb_cnt[i];
            cb_cnt[i] = run;  // the row's cursor
            if (lo + i < n) RS[lo + i] = (int)run;
            run += c;
        }
    }
    if (h == H - 1 && tid == CB_TPB - 1) RS[n] = (int)run;  // (the last thread's run ends behind the last row)
    if (tid == CB_TPB - 1) seg_end = run;
    __syncthreads();
    int *__restrict__ P = perm + (size_t)bi * S;
    // this workgroup's rows are ONE contiguous segment of perm, [first, last): the slot numbers meet in LDS and leave as coalesced
    // stores (scattered 4-byte stores from every lane were 7 of the kernel's 21 us); what a lopsided index distribution puts
    // beyond the stage goes straight to memory
    unsigned *stage = cb_cnt + nbl;
    unsigned first = 0;
    for (int w = 0; w < CB_TPB / 64; w++) first += lowsum[w];
    const unsigned last = seg_end;
    auto place = [&](int k, int slot) {
        if (k >= lo && k < hi) {
            const unsigned pos = atomicAdd(&cb_cnt[k - lo], 1u) - first;
            if (pos < (unsigned)CB_STAGE) stage[pos] = (unsigned)slot;
            else P[first + pos] = slot;
        }
    };
    if (REG) {
#pragma unroll
        for (int u = 0; u < CB_RPT; u++) place(kr[u], tid + u * CB_TPB);
    } else {
        for (int s0 = tid; s0 < S; s0 += 8 * CB_TPB) {
            int k[8];
#pragma unroll
            for (int u = 0; u < 8; u++) k[u] = s0 + u * CB_TPB < S ? I[s0 + u * CB_TPB] : -1;
#pragma unroll
            for (int u = 0; u < 8; u++) place(k[u], s0 + u * CB_TPB);
        }
    }
    __syncthreads();
    const unsigned cnt = min(last - first, (unsigned)CB_STAGE);
    for (unsigned j = tid; j < cnt; j += CB_TPB) P[first + j] = (int)stage[j];
}

typedef float cg_v4f __attribute__((ext_vector_type(4)));
template <int VEC>
struct CgVec;
template <>
struct CgVec<4> {
    typedef cg_v4f T;
};
template <>
struct CgVec<1> {
    typedef float T;
};

constexpr int CG_TPB = 256;
// (same device, C3's gradient shapes, tools/ab_group_grad.py: 2 rows x 2 slots 84 us; 1 x 4 88, 2 x 4 83, 1 x 8 87, 4 x 2 101; the
// source rows by non-temporal loads 97 -- and 84 instead of 52 for three_interpolate's, whose rows are read three times)
constexpr int CG_RPT = 2;  // destination rows per lane row (their slot chains run side by side)
constexpr int CG_Q = 2;    // slots of a row in flight

// dst (b, n, c) row r = sum over p in [row_start[r], row_start[r + 1]) of w[perm[p]] * src[perm[p] / K][:]
template <int VEC, int K, bool WEIGHTED>
__global__ __launch_bounds__(CG_TPB) void rows_csr_gather_kernel(int n, int c, int S, int tx_log2, int bpb /* blocks per sample */,
                                                                 const float *__restrict__ src, const float *__restrict__ weight,
                                                                 const int *__restrict__ row_start, const int *__restrict__ perm,
                                                                 float *__restrict__ dst) {
    typedef typename CgVec<VEC>::T V;
    const unsigned logical = rf::xcd_contiguous(blockIdx.x, gridDim.x);
    const int bi = logical / bpb, bx = logical - bi * bpb;
    const int TX = 1 << tx_log2, TY = CG_TPB >> tx_log2;
    const int lx = threadIdx.x & (TX - 1), ly = threadIdx.x >> tx_log2;
    const int cv = c / VEC;
    const int *__restrict__ RS = row_start + (size_t)bi * (n + 1);
    const int *__restrict__ P = perm + (size_t)bi * S;
    const float *__restrict__ W = WEIGHTED ? weight + (size_t)bi * S : nullptr;
    const V *__restrict__ G = (const V *)(src + (size_t)bi * (S / K) * c);
    V *__restrict__ O = (V *)(dst + (size_t)bi * n * c);
    int r[CG_RPT], beg[CG_RPT], end[CG_RPT];
#pragma unroll
    for (int u = 0; u < CG_RPT; u++) {
        r[u] = (bx * CG_RPT + u) * TY + ly;
        const int rr = min(r[u], n - 1);
        beg[u] = RS[rr];
        end[u] = r[u] < n ? RS[rr + 1] : beg[u];
    }
    for (int l = lx; l < cv; l += TX) {
        double acc[CG_RPT][VEC];
#pragma unroll
        for (int u = 0; u < CG_RPT; u++)
#pragma unroll
            for (int k = 0; k < VEC; k++) acc[u][k] = 0.0;
        int p[CG_RPT];
#pragma unroll
        for (int u = 0; u < CG_RPT; u++) p[u] = beg[u];
        // the rows' chains side by side, CG_Q slots of each in flight
        while (true) {
            bool any = false;
#pragma unroll
            for (int u = 0; u < CG_RPT; u++) any = any || p[u] < end[u];
            if (!any) break;
            int s[CG_RPT][CG_Q];
            V v[CG_RPT][CG_Q];
            float w[CG_RPT][CG_Q];
#pragma unroll
            for (int u = 0; u < CG_RPT; u++)
#pragma unroll
                for (int q = 0; q < CG_Q; q++) s[u][q] = p[u] + q < end[u] ? P[p[u] + q] : -1;
#pragma unroll
            for (int u = 0; u < CG_RPT; u++)
#pragma unroll
                for (int q = 0; q < CG_Q; q++) {
                    const int ss = max(s[u][q], 0);
                    v[u][q] = G[(size_t)(ss / K) * cv + l];
                    w[u][q] = s[u][q] < 0 ? 0.f : (WEIGHTED ? W[ss] : 1.f);
                }
#pragma unroll
            for (int u = 0; u < CG_RPT; u++)
#pragma unroll
                for (int q = 0; q < CG_Q; q++) {
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            acc[u][k] += s[u][q] < 0 ? 0.0 : (WEIGHTED ? (double)(v[u][q][k] * w[u][q]) : (double)v[u][q][k]);
                    } else {
                        acc[u][0] += s[u][q] < 0 ? 0.0 : (WEIGHTED ? (double)(v[u][q] * w[u][q]) : (double)v[u][q]);
                    }
                }
#pragma unroll
            for (int u = 0; u < CG_RPT; u++) p[u] += CG_Q;
        }
#pragma unroll
        for (int u = 0; u < CG_RPT; u++) {
            if (r[u] >= n) continue;
            V out;
            if constexpr (VEC == 4) {
                out = V{(float)acc[u][0], (float)acc[u][1], (float)acc[u][2], (float)acc[u][3]};
            } else {
                out = (float)acc[u][0];
            }
            __builtin_nontemporal_store(out, &O[(size_t)r[u] * cv + l]);
        }
    }
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace rfs {

bool rows_csr_supported(int b, int n, int c, long S, int K) {
    // (32-bit offsets inside a sample; a sample's key space over at most 8 workgroups of 32768 bins)
    return b > 0 && b <= 65535 && n > 0 && c > 0 && S > 0 && S % K == 0 && n <= 8 * CB_MAXBINS && S < (1L << 30) &&
           (long)n * c < (1L << 31) && (S / K) * (long)c < (1L << 31);
}

size_t rows_csr_workspace_bytes(int b, int n, long S) {
    return align256(sizeof(int) * (size_t)b * ((size_t)n + 1)) + align256(sizeof(int) * (size_t)b * (size_t)S);
}

int rows_csr_scatter(int b, int n, int c, long S, int K, const float *src, const int *idx, const float *weight, float *dst,
                     void *workspace, const char *build_name, const char *gather_name, hipStream_t s) {
    int *row_start = (int *)workspace;
    int *perm = (int *)((char *)workspace + align256(sizeof(int) * (size_t)b * ((size_t)n + 1)));
    // workgroups per sample: enough to put ~128 CUs to work, every range at least 2048 bins, at most 32768
    int H = rf::ceil_div(n, CB_MAXBINS);
    while ((long)b * H < 128 && n / (2 * H) >= 2048 && H < 8) H *= 2;
    const int nbl = rf::ceil_div(n, H);
#define RFS_BUILD(RPT, EXTRA)                                                                                                          \
    RF_HIP(hipFuncSetAttribute((const void *)rows_csr_build_kernel<RPT>, hipFuncAttributeMaxDynamicSharedMemorySize,                   \
                               (CB_MAXBINS + (EXTRA)) * 4));                                                                           \
    RF_LAUNCH(build_name, rows_csr_build_kernel<RPT>, dim3((unsigned)(b * H)), dim3(CB_TPB), sizeof(unsigned) * ((size_t)nbl + (EXTRA)), \
              s, n, (int)S, H, nbl, idx, row_start, perm)
    if (S <= (long)CB_TPB * 32) {
        RFS_BUILD(32, CB_STAGE);
    } else {  // (48 slots per thread -- three_interpolate's 3 x 16384 -- spill: 128 registers is all a 1024-thread workgroup has)
        RFS_BUILD(0, CB_STAGE);
    }
#undef RFS_BUILD
    const bool vec = c % 4 == 0 && rf::aligned16(src) && rf::aligned16(dst);
    const int cv = vec ? c / 4 : c;
    int tx_log2 = 0;
    while ((1 << tx_log2) < cv && tx_log2 < 6) tx_log2++;
    const int rpb = (CG_TPB >> tx_log2) * CG_RPT;
    const int bpb = rf::ceil_div(n, rpb);
    const dim3 grid((unsigned)((long)bpb * b));
#define RFS_GO(VEC, KK, WT)                                                                                                \
    RF_LAUNCH(gather_name, (rows_csr_gather_kernel<VEC, KK, WT>), grid, dim3(CG_TPB), 0, s, n, c, (int)S, tx_log2, bpb, src, weight, \
              (const int *)row_start, (const int *)perm, dst)
    if (K == 1 && !weight) {
        if (vec) { RFS_GO(4, 1, false); } else { RFS_GO(1, 1, false); }
    } else if (K == 3 && weight) {
        if (vec) { RFS_GO(4, 3, true); } else { RFS_GO(1, 3, true); }
    } else {
        return RF_EINVAL;
    }
#undef RFS_GO
    return RF_OK;
}

}  // namespace rfs
