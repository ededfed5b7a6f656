// emd_fgt.hip -- the BROAD levels of approx_match's schedule (tf_approxmatch.cu:36-177: levels -1, -0.25 and 0 of
// -4^7 ... -4^-1, 0) without their n x m sweeps.
//
// Every phase of a level is a row sum S[row] = sum_col w[col] * exp(-a |row - col|^2).  For a <= 1 and clouds of about unit
// extent the kernel is smooth enough for ONE truncated Taylor expansion about the clouds' common centre O (a fast Gauss
// transform with a single box):
//     exp(-a |x - y|^2) = exp(-a |x|^2) exp(-a |y|^2) exp(2a x.y),      x, y relative to O
//     exp(g x.y) = sum over multi-indices (i,j,k) of  g^(i+j+k) / (i! j! k!) * x^(ijk) * y^(ijk),   g = 2a
// so   S[row] = exp(-a |row|^2) * sum_(ijk) coef(ijk) * row^(ijk) * M(ijk),      M(ijk) = sum_col w[col] exp(-a |col|^2) col^(ijk):
// 286 moments (total degree <= 10; 165 -- degree 8 -- at level -0.25, one at level 0) per batch element replace the 2048 x 2048
// exponentials of a sweep.  Everything here is fp64, the state vectors stay fp32 with the dense sweeps' own update formulas.
//
// ACCURACY IS CERTIFIED PER ROW, not assumed.  The degree-P series of exp(t), t = g x.y, misses by at most
// |t|^(P+1) / (P+1)! * max(1, e^t); with |t| <= g |x| |y| and |y| <= R_cols the error of a row's sum is at most
//     E(x) = exp(-a |x|^2) * (g |x|)^(P+1) / (P+1)! * exp(g |x| R_cols) * Q,      Q = sum_col w[col] exp(-a |col|^2) |col|^(P+1)
// -- ONE more scalar moment per set (slot FG_NM of every moment row).  A row whose E(x) exceeds kRowEps (4e-6) of its sum is
// NOT taken from the series: the workgroup forms the sums of its failed rows directly (row_fixup: the dense sweeps' fp32 terms).
// On clouds that fill their box (C4: uniform in a unit cube, g R R = 1.39 at level -1) few pairs are extreme: with the weights
// the schedule really meets there (the mass the sharp levels leave over lives in the corners) the true error peaks at 3.4e-7
// and the bound at 2.5e-6 over C4's 32 samples -- no row fails.  Two clusters in opposite corners put EVERY pair at t ~ -1.5,
// where the degree-10 series is 4e-6 off (bound 1.8e-5): every row fails its certificate and is summed directly
// (tests/test_gpu_emd.py::test_expansion_on_opposite_corner_clusters; numbers: tools/experiments/fgt_row_bound.py
// with flat weights, fgt_row_bound_chain.py with the chain's own).  Against the reference the result moves by LESS than the dense sweeps' segment-order sums do (which sit ~1e-7 from
// the reference's sequential fp32 sums): replacing the sums of levels 6-8 by exact ones leaves the set of `match` entries outside
// the strict bar unchanged, entry for entry (tools/experiments/fgt_parity.py).
//
// A coarse pre-filter runs per batch element: fgt_prep measures R_rows * R_cols and raises the element's `bad` flag when
// g_max * R_rows * R_cols > kBound (clouds much larger than the unit cube) or a coordinate is not finite: the kernels here then
// form that element's row sums directly (direct_sum: the dense sweeps' arithmetic, inside the same launches) -- no host
// synchronisation either way, and nothing to clear before a call.
//
// One launch per phase: a workgroup EVALUATES the phase for its 256 rows from the other cloud's moments (staged in LDS, the chunk
// partials summed in a fixed order: deterministic) and then ACCUMULATES the moments its own rows contribute, as columns, to the
// next phase (their new weights are in its registers) -- so the chain P1 -> P2 -> P3+P1 -> ... alternates between the two clouds
// with no separate moment launches.
#include "common.hpp"
#include "emd_fgt.hpp"
#include "emd_fgt_prep.hpp"

namespace {
using rfe::Geom;
using rfe::kBound;
using rfe::kRowEps;

constexpr int FG_P = 10;                                            // largest total degree of an expansion
constexpr int FG_NM = (FG_P + 1) * (FG_P + 2) * (FG_P + 3) / 6;     // 286 monomials
constexpr int FG_NMX = FG_NM + 2;                                   // a moment row: the monomials, then Q (see above), then padding
constexpr int FG_ROWS = 256;                                        // rows (and new columns) per workgroup
constexpr int FG_TPB = 2 * FG_ROWS;                                 // threads per workgroup: two per row (see fgt_step_kernel)
constexpr int FG_PWS = FG_ROWS + 1;                                 // row stride of the power tables in LDS (doubles)
constexpr size_t FG_LDS = sizeof(double) * (2 * (FG_P + 1) * (FG_P + 2) / 2 * 12 + 2 * FG_ROWS + 4 * (FG_P + 1) * (size_t)FG_PWS);

struct Ijk {
    unsigned char i, j, k, n;
};
struct MonoTable {
    Ijk t[FG_NM];
    constexpr MonoTable() : t{} {
        int idx = 0;
        for (int i = 0; i <= FG_P; i++)
            for (int j = 0; j <= FG_P - i; j++)
                for (int k = 0; k <= FG_P - i - j; k++) {
                    t[idx].i = (unsigned char)i;
                    t[idx].j = (unsigned char)j;
                    t[idx].k = (unsigned char)k;
                    t[idx].n = (unsigned char)(i + j + k);
                    idx++;
                }
    }
};
__device__ constexpr MonoTable kMono{};
// flat index of monomial (i, j, k) in that order
constexpr int mono_index(int i, int j, int k) {
    int idx = 0;
    for (int a = 0; a < i; a++) idx += (FG_P - a + 1) * (FG_P - a + 2) / 2;
    for (int c = 0; c < j; c++) idx += FG_P - i - c + 1;
    return idx + k;
}
// The moment accumulation's work items: two monomials that differ by one power of x -- (i0, j, k) and (i0 + 1, j, k) -- share the
// y^j z^k factor of every point.  161 items cover the 286 monomials.
struct Item {
    unsigned char i0, j, k, cnt;
    unsigned short t0, t1;
};
constexpr int FG_NI = 161;
struct ItemTable {
    Item t[FG_NI];
    constexpr ItemTable() : t{} {
        int n = 0;
        for (int j = 0; j <= FG_P; j++)
            for (int k = 0; k <= FG_P - j; k++)
                for (int i0 = 0; i0 <= FG_P - j - k; i0 += 2) {
                    const bool two = i0 + 1 <= FG_P - j - k;
                    t[n].i0 = (unsigned char)i0;
                    t[n].j = (unsigned char)j;
                    t[n].k = (unsigned char)k;
                    t[n].cnt = two ? 2 : 1;
                    t[n].t0 = (unsigned short)mono_index(i0, j, k);
                    t[n].t1 = (unsigned short)(two ? mono_index(i0 + 1, j, k) : 0);
                    n++;
                }
    }
};
__device__ constexpr ItemTable kItems{};
__device__ constexpr double kInvFact[13] = {1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320,
                                            1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }
int chunks_of(int npts) { return (npts + FG_ROWS - 1) / FG_ROWS; }  // one partial per workgroup

struct FgtWs {
    Geom *geom;
    double *mom[2];  // [side][(set * b + bi) * chunks + chunk][FG_NMX]: side 0 = moments over xyz1 (columns of P2), 1 = over xyz2
    int chmax;
};
FgtWs view(const void *ws, int b, int nmax) {
    FgtWs v;
    char *p = (char *)ws;
    v.geom = (Geom *)p;
    p += align256(sizeof(Geom) * (size_t)b);
    v.chmax = chunks_of(nmax);
    const size_t per = align256(sizeof(double) * 2 * (size_t)b * v.chmax * FG_NMX);
    v.mom[0] = (double *)p;
    v.mom[1] = (double *)(p + per);
    return v;
}

struct Step {
    // rows (this launch evaluates them) and, as columns of the NEXT phase, the same points
    const float *rows;   // (b, nrows, 3)
    int nrows;
    // moments of the other cloud, to evaluate with: up to two sets
    const double *min;   // [(set * b + bi) * chin + chunk][FG_NMX]
    int chin;            // chunks per batch element in `min`
    int nin;             // 1 or 2 sets
    double a_in[2];      // sharpness of each input set
    int deg_in[2];       // its expansion's total degree (<= FG_P: broader levels need fewer terms)
    // the columns behind those moments, for the direct sums of a refused call (`bad`): points and the sets' fp32 weights
    const float *cols;   // (b, ncols, 3)
    int ncols;
    int cols_are_set2;   // the columns are xyz2 (their radius about the centre: Geom::r2), else xyz1 (Geom::r1)
    const float *wcol[2];
    // moments this launch leaves for the next phase (NULL: none)
    double *mout;        // [(set * b + bi) * chout + chunk][FG_NMX]
    int chout;
    int nout;            // 0, 1 or 2 sets
    double a_out[2];
    int deg_out[2];
    // state (fp32, `stride` floats per batch element)
    size_t stride;
    int mode;            // 0: moments only (w = v0 at a_out[0]); 1: P1 (+P3 when nin == 2); 2: P2
    const float *v0;     // mode 0: the weights; mode 1: ratioL_prev (P3); mode 2: unused
    float *rem;          // mode 1: remainL; mode 2: remainR
    float *out;          // mode 1: ratioL_out; mode 2: ratioR_out
    const Geom *geom;
    int b;
};

// sum_(ijk) Ms * x^i y^j z^k by Horner's rule in z inside y inside x.  Ms is staged as one ROW per (i, j) -- the k = 0..FG_P-i-j
// coefficients, zero-padded to FG_RL entries -- so that every row is the same straight-line code: its 14 entries arrive by seven
// 16-byte broadcast reads issued together, then a 13-step fma chain; the rows of one i are independent but for one fma each, so
// the loads of the next row overlap the chain of this one.  (As one pointer walking down the unpadded table every fma waited for
// its own LDS read: 54 us per launch; fully unrolled, the compiler hoisted all the reads and spilled 2.6 KB per lane.)
constexpr int FG_NR = (FG_P + 1) * (FG_P + 2) / 2;  // 91 rows (i, j)
constexpr int FG_RL = 12;                           // entries per row: up to FG_P + 1 = 11 coefficients, padded to an even count
__device__ __forceinline__ int row_of(int i, int j) { return i * (FG_P + 1) - i * (i - 1) / 2 + j; }
// istep = 1: the whole series.  istep = 2: only the powers of x with the parity of `ifirst` (Horner in x^2): a row's two threads
// take one parity each and the series is even(x^2) + x * odd(x^2).
__device__ __forceinline__ double eval_series(const double *__restrict__ Ms, int deg, double x, double y, double z, int ifirst,
                                              int istep) {
    static_assert((FG_P & 1) == 0, "the row polynomial's even / odd split assumes an even top degree");
    double accx = 0.0;
    const double xs = istep == 2 ? x * x : x, z2 = z * z;
#pragma unroll 1
    for (int i = ifirst; i >= 0; i -= istep) {  // (rows beyond the set's degree hold zeros and are not visited)
        double accy = 0.0;
        int r = row_of(i, deg - i);
#pragma unroll 2
        for (int j = deg - i; j >= 0; --j, --r) {
            const double2 *mr = (const double2 *)(Ms + r * FG_RL);
            double m[FG_RL];
#pragma unroll
            for (int q = 0; q < FG_RL / 2; q++) {
                const double2 v = mr[q];
                m[2 * q] = v.x;
                m[2 * q + 1] = v.y;
            }
            // the row's polynomial in z as even(z^2) + z * odd(z^2): two chains of six instead of one of eleven
            double ev = m[FG_P], od = m[FG_P - 1];  // (FG_P is even)
#pragma unroll
            for (int k = FG_P - 2; k >= 0; k -= 2) ev = fma(ev, z2, m[k]);
#pragma unroll
            for (int k = FG_P - 3; k >= 1; k -= 2) od = fma(od, z2, m[k]);
            accy = fma(accy, y, fma(od, z, ev));
        }
        accx = fma(accx, xs, accy);
    }
    return accx;
}

// the direct sum of a refused call: S = sum_col w[col] exp2(d2 * c) over the columns [lo, hi), fp32 as the dense sweeps form their
// terms.  One row per lane; the columns are wave-uniform and come by scalar loads, eight columns (24 + 8 dwords) per batch in two
// register sets: the wait for a batch comes before the next batch's issue (scalar loads return out of order, so the only wait
// there is is "all of them").  About the dense sweeps' own speed -- the path of clouds far outside the unit cube at these levels
// (an untrained network's output) and of non-finite coordinates.
typedef const __attribute__((address_space(4))) float cfloat;
// TWO: both sums of a fused P3 + P1 phase from one d2 per pair (acc[1]: weights w1, multiplier c1), as am_rowk_kernel forms them.
template <bool TWO>
__device__ __forceinline__ void direct_sum(const float *C_, const float *w0_, const float *w1_, int lo, int hi, float c0, float c1,
                                           float x, float y, float z, float (&acc)[2]) {
    cfloat *C = (cfloat *)C_;
    cfloat *w0 = (cfloat *)w0_;
    cfloat *w1 = (cfloat *)w1_;
    acc[0] = acc[1] = 0.f;
    const bool one = c0 == 0.f;
    auto term = [&](float cx, float cy, float cz, float wl0, float wl1) {
        const float d2 = rf::d2_fma(cx - x, cy - y, cz - z);
        acc[0] = fmaf(one ? 1.0f : __builtin_amdgcn_exp2f(d2 * c0), wl0, acc[0]);
        if (TWO) acc[1] = fmaf(__builtin_amdgcn_exp2f(d2 * c1), wl1, acc[1]);
    };
    int l = lo;
    if (!TWO && one) {  // (uniform) level 0 on its own: the sum of the weights, as am_rowl_kernel<.., ZERO> takes it
        for (; l + 8 <= hi; l += 8) {
            float ww[8];
#pragma unroll
            for (int u = 0; u < 8; u++) ww[u] = w0[l + u];
#pragma unroll
            for (int u = 0; u < 8; u++) acc[0] = fmaf(1.0f, ww[u], acc[0]);
        }
        for (; l < hi; l++) acc[0] = fmaf(1.0f, w0[l], acc[0]);
        return;
    }
    float ca[24], wa[8], va[8], cb[24], wb[8], vb[8];
#define RFE_DS_LOAD(c, ww, vv, at)                                                              \
    _Pragma("unroll") for (int u = 0; u < 24; u++) c[u] = C[(size_t)(at) * 3 + u];              \
    _Pragma("unroll") for (int u = 0; u < 8; u++) ww[u] = w0[(at) + u];                         \
    if (TWO) { _Pragma("unroll") for (int u = 0; u < 8; u++) vv[u] = w1[(at) + u]; }
#define RFE_DS_USE(c, ww, vv) \
    _Pragma("unroll") for (int u = 0; u < 8; u++) term(c[3 * u], c[3 * u + 1], c[3 * u + 2], ww[u], TWO ? vv[u] : 0.f);
    if (l + 8 <= hi) {
        RFE_DS_LOAD(ca, wa, va, l)
        for (; l + 24 <= hi; l += 16) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            RFE_DS_LOAD(cb, wb, vb, l + 8)
            __builtin_amdgcn_sched_barrier(0);
            RFE_DS_USE(ca, wa, va)
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            RFE_DS_LOAD(ca, wa, va, l + 16)
            __builtin_amdgcn_sched_barrier(0);
            RFE_DS_USE(cb, wb, vb)
        }
        RFE_DS_USE(ca, wa, va)
        l += 8;
    }
#undef RFE_DS_LOAD
#undef RFE_DS_USE
    for (; l < hi; l++) term(C[(size_t)l * 3], C[(size_t)l * 3 + 1], C[(size_t)l * 3 + 2], w0[l], TWO ? w1[l] : 0.f);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The rows of this workgroup that failed their certificate, FX_G at a time by the WHOLE workgroup: every thread loads its share of
// the columns once per group and forms its terms for all the group's rows (the dense sweeps' fp32 terms -- d2 by the fma chain,
// v_exp_f32 -- products and sums in double); one block reduction per group.  (One wave per failed row was tried first: a lone
// failed row then cost its workgroup -- and so the launch -- a chain of eight dependent load round trips, ~10 us.)
// All FG_TPB threads must call.  red: 8 x 2 FX_G doubles of LDS.
constexpr int FX_G = 8;
__device__ __forceinline__ void row_fixup(const Step &p, int bi, int nf, const int *__restrict__ flist, double *__restrict__ fix,
                                          double *__restrict__ red) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float c0 = (float)(-p.a_in[0] / 0.69314718055994530942), c1 = (float)(-p.a_in[1] / 0.69314718055994530942);
    const float *C = p.cols + (size_t)bi * p.ncols * 3;
    const float *w0 = p.wcol[0] + (size_t)bi * p.stride;
    const float *w1 = p.nin == 2 ? p.wcol[1] + (size_t)bi * p.stride : w0;
    const float *Rb = p.rows + ((size_t)bi * p.nrows + (size_t)blockIdx.x * FG_ROWS) * 3;
    for (int e0 = 0; e0 < nf; e0 += FX_G) {
        const int cnt = min(FX_G, nf - e0);
        float rx[FX_G], ry[FX_G], rz[FX_G];
        double a0[FX_G], a1[FX_G];
#pragma unroll
        for (int u = 0; u < FX_G; u++) {
            const float *R = Rb + (size_t)flist[min(e0 + u, nf - 1)] * 3;
            rx[u] = R[0], ry[u] = R[1], rz[u] = R[2];
            a0[u] = a1[u] = 0.0;
        }
        for (int l0 = tid; l0 < p.ncols; l0 += 4 * FG_TPB) {  // four columns per thread in flight
            float cx[4], cy[4], cz[4], u0[4], u1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int l = min(l0 + FG_TPB * q, p.ncols - 1);
                const bool in = l0 + FG_TPB * q < p.ncols;
                cx[q] = C[(size_t)l * 3], cy[q] = C[(size_t)l * 3 + 1], cz[q] = C[(size_t)l * 3 + 2];
                u0[q] = in ? w0[l] : 0.f;
                u1[q] = in ? w1[l] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int u = 0; u < FX_G; u++) {
                    const float d2 = rf::d2_fma(cx[q] - rx[u], cy[q] - ry[u], cz[q] - rz[u]);
                    a0[u] += (double)(c0 == 0.f ? 1.0f : __builtin_amdgcn_exp2f(d2 * c0)) * (double)u0[q];
                    if (p.nin == 2) a1[u] += (double)__builtin_amdgcn_exp2f(d2 * c1) * (double)u1[q];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FX_G; u++) {
            a0[u] = wave_sum_f64(a0[u]);
            a1[u] = wave_sum_f64(a1[u]);
            if (lane == 0) {
                red[wv * 2 * FX_G + u] = a0[u];
                red[wv * 2 * FX_G + FX_G + u] = a1[u];
            }
        }
        __syncthreads();
        if (tid < 2 * FX_G && (tid & (FX_G - 1)) < cnt) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < FG_TPB / 64; w++) sum += red[w * 2 * FX_G + tid];
            fix[(tid / FX_G) * FG_ROWS + flist[e0 + (tid & (FX_G - 1))]] = sum;
        }
        __syncthreads();
    }
}

// 512 threads for 256 rows: the chip has only b * rows / 64 = 1024 waves of rows at C4, ONE per SIMD, and both halves of this
// kernel are latency-bound at that (fp64 fma chains in the evaluation, LDS reads in the moments: 21-39 us per launch with 256
// threads).  So thread t and t + 256 share row t: they evaluate one input set each, and in the moment accumulation each takes
// half of the workgroup's points (their sums meet in LDS).
__global__ __launch_bounds__(FG_TPB) void fgt_step_kernel(Step p) {
    // dynamic LDS (131 KB of the CU's 160: one workgroup per CU, which is all the grid has at C4):
    //   Ms   [2][FG_NR * FG_RL]        the staged input moments
    //   s1x  [FG_ROWS]                 what the upper half of the threads hands to the lower half
    //   xc   [FG_ROWS]                 the rows' x coordinates
    //   pw   [4][FG_P + 1][FG_PWS]     power tables of ALL the workgroup's rows as columns: x^e (set 0 | set 1, weights folded in) | y^e | z^e;
    //                                  row stride FG_PWS = 257: lanes of one wave that read different exponents hit different banks
    extern __shared__ __attribute__((aligned(16))) double fg_lds[];
    double(*Ms)[FG_NR * FG_RL] = (double(*)[FG_NR * FG_RL])fg_lds;
    double *s1x = fg_lds + 2 * FG_NR * FG_RL;
    double *xc = s1x + FG_ROWS;
    double *pw = xc + FG_ROWS;
    __shared__ int nfail;          // rows of this workgroup whose certificate failed
    __shared__ double qin[2];      // the input sets' Q (slot FG_NM of their moment rows, chunk partials summed)
    __shared__ double qred[2][4];  // the four row-waves' partial Q of the moments this launch leaves
    const int bi = blockIdx.y, tid = threadIdx.x;
    const int rt = tid & (FG_ROWS - 1), half = tid >> 8;  // FG_ROWS == 256
    const int row = blockIdx.x * FG_ROWS + rt;
    const bool live = row < p.nrows;
    const Geom g = p.geom[bi];
    const bool bad = g.bad != 0;  // (uniform) the expansion was refused for this batch element: direct sums, no moments
    const float *R = p.rows + ((size_t)bi * p.nrows + (live ? row : p.nrows - 1)) * 3;
    const double x = (double)R[0] - g.ox, y = (double)R[1] - g.oy, z = (double)R[2] - g.oz;
    const double r2 = x * x + y * y + z * z;
    float wnew[2] = {0.f, 0.f};  // this row's weights as a column of the next phase (lower half of the threads)
    if (p.mode == 0) {
        wnew[0] = live ? p.v0[(size_t)bi * p.stride + row] : 0.f;
    } else {
        double S0 = 0.0, S1 = 0.0;
        if (!bad) {
            // ---- stage the input moments: chunk partials summed in a fixed order (eight loads in flight), times
            //      coef(ijk) = g^(i+j+k) / (i! j! k!); entries beyond a set's degree stay zero
            for (int t = tid; t < 2 * FG_NR * FG_RL; t += FG_TPB) (&Ms[0][0])[t] = 0.0;
            if (tid == FG_TPB - 1) nfail = 0;
            if (tid < p.nin) {  // (fixed order: deterministic)
                const double *src = p.min + ((size_t)(tid * p.b + bi) * p.chin) * FG_NMX + FG_NM;
                double q = 0.0;
                for (int c0 = 0; c0 < p.chin; c0++) q += src[(size_t)c0 * FG_NMX];
                qin[tid] = q;
            }
            __syncthreads();
            for (int s = 0; s < p.nin; s++) {
                const double gam = 2.0 * p.a_in[s];
                for (int t = tid; t < FG_NM; t += FG_TPB) {
                    const Ijk e = kMono.t[t];
                    if (e.n > p.deg_in[s]) continue;
                    const double *src = p.min + ((size_t)(s * p.b + bi) * p.chin) * FG_NMX + t;
                    double sum = 0.0;
                    for (int c0 = 0; c0 < p.chin; c0 += 8) {
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = c0 + u < p.chin ? src[(size_t)(c0 + u) * FG_NMX] : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; u++) sum += v[u];
                    }
                    double gp = 1.0;
                    for (int q = 0; q < e.n; q++) gp *= gam;
                    Ms[s][row_of(e.i, e.j) * FG_RL + e.k] = sum * gp * kInvFact[e.i] * kInvFact[e.j] * kInvFact[e.k];
                }
            }
            __syncthreads();
            if (p.nin == 2) {  // a set each
                if (half == 0) S0 = exp(-p.a_in[0] * r2) * eval_series(Ms[0], p.deg_in[0], x, y, z, p.deg_in[0], 1);
                else s1x[rt] = exp(-p.a_in[1] * r2) * eval_series(Ms[1], p.deg_in[1], x, y, z, p.deg_in[1], 1);
                __syncthreads();
                S1 = s1x[rt];
            } else {  // one set: the even powers of x here, the odd ones there
                const int d = p.deg_in[0];
                const int ifirst = half == 0 ? (d & ~1) : (d >= 1 ? ((d - 1) | 1) : -1);  // largest even / odd i <= d (none: -1)
                const double part = eval_series(Ms[0], d, x, y, z, ifirst, 2);
                if (half == 1) s1x[rt] = part * x;
                __syncthreads();
                S0 = exp(-p.a_in[0] * r2) * (part + s1x[rt]);
            }
            // ---- the certificate (header): a row whose error bound exceeds kRowEps of its sum does not take it from the series
            double *fix = pw;                        // [2][FG_ROWS] (the power tables are not in use yet)
            int *flist = (int *)(pw + 2 * FG_ROWS);  // the workgroup's failed rows
            bool fail = false;
            if (half == 0 && live) {
                const double r = sqrt(r2), rc = p.cols_are_set2 ? g.r2 : g.r1;
                for (int s = 0; s < p.nin; s++) {
                    const double gr = 2.0 * p.a_in[s] * r;
                    double pwr = gr;
                    for (int q = 0; q < p.deg_in[s]; q++) pwr *= gr;  // (g |x|)^(P + 1)
                    const double bound = exp(gr * rc - p.a_in[s] * r2) * pwr * kInvFact[p.deg_in[s] + 1] * qin[s];
                    fail = fail || !(bound <= kRowEps * (s == 0 ? S0 : S1));
                }
            }
            if (fail) flist[atomicAdd(&nfail, 1)] = rt;
            __syncthreads();
            const int nf = nfail;
            if (nf > 0) {  // (uniform)
                row_fixup(p, bi, nf, flist, fix, pw + 2 * FG_ROWS + FG_ROWS / 2);
                __syncthreads();
                if (fail) {
                    S0 = fix[rt];
                    S1 = fix[FG_ROWS + rt];
                }
            }
        } else {
            // refused: direct sums, half of the columns for each of a row's two threads
            const int h = __builtin_amdgcn_readfirstlane(half);  // (a wave lies in one half)
            const int mid = (p.ncols / 2) & ~7;
            const int lo = h == 0 ? 0 : mid, hi = h == 0 ? mid : p.ncols;
            // (a = -c ln 2 in double, so -a / ln 2 rounds back to the sweeps' own float multiplier c = level * kLog2e)
            const float c0 = (float)(-p.a_in[0] / 0.69314718055994530942), c1 = (float)(-p.a_in[1] / 0.69314718055994530942);
            const float *C = p.cols + (size_t)bi * p.ncols * 3;
            const float *w0 = p.wcol[0] + (size_t)bi * p.stride;
            float acc[2];
            if (p.nin == 2) direct_sum<true>(C, w0, p.wcol[1] + (size_t)bi * p.stride, lo, hi, c0, c1, R[0], R[1], R[2], acc);
            else direct_sum<false>(C, w0, w0, lo, hi, c0, c0, R[0], R[1], R[2], acc);
            if (half == 1) {
                s1x[rt] = (double)acc[0];
                xc[rt] = (double)acc[1];  // (xc is not in use before the moment pass, which a refused call never reaches)
            }
            __syncthreads();
            S0 = (double)(acc[0] + (float)s1x[rt]);
            S1 = (double)(acc[1] + (float)xc[rt]);
        }
        if (live && half == 0) {
            const size_t o = (size_t)bi * p.stride + row;
            if (p.mode == 1) {  // P1 of this level (set 0: w = remainR), behind the fused P3 of the previous one (set 1: w = ratioR_prev)
                float rem = p.rem[o];
                if (p.nin == 2) {
                    const float t3 = (float)((double)p.v0[o] * S1);
                    rem = fmaxf(0.0f, rem - t3);
                    p.rem[o] = rem;
                }
                const float t1 = (float)(1e-9 + S0);
                const float ratio = rem / t1;
                p.out[o] = ratio;
                wnew[0] = ratio;  // P2 of this level sums ratioL over these points
            } else {  // P2 (am_rowl_kernel's epilogue)
                const float sumr = (float)S0;
                const float rem = p.rem[o];
                const float t = sumr * rem;
                const float cons = fminf(rem / (t + 1e-9f), 1.0f);
                const float ratioR = rem * cons;
                const float left = fmaxf(0.0f, rem - t);
                p.out[o] = ratioR;
                p.rem[o] = left;
                wnew[0] = left;    // P1 of the next level sums remainR ...
                wnew[1] = ratioR;  // ... and its fused P3 sums this level's ratioR
            }
        }
    }
    if (p.nout == 0 || bad) return;
    // ---- the moments these rows contribute, as columns, to the next phase.  The power tables of the workgroup's 256 points go to
    //      LDS -- x^e once per set with the set's weight and exp(-a |col|^2) folded in, y^e and z^e shared, and the points' x --
    //      then thread <-> (item, half of the points): an item is two monomials one power of x apart, so a point costs 5 LDS
    //      reads for its 4 products (both sets).
    double wt0 = 0.0, wt1 = 0.0;
    if (half == 0 && live) {
        wt0 = (double)wnew[0] * exp(-p.a_out[0] * r2);
        if (p.nout == 2) wt1 = (double)wnew[1] * exp(-p.a_out[1] * r2);
    }
    {  // Q of the sets these rows leave: sum of W |col|^(P + 1) (one partial per row-wave here, one per workgroup below)
        const double r = sqrt(r2);
        double q0 = wt0 * r, q1 = wt1 * r;
        for (int q = 0; q < p.deg_out[0]; q++) q0 *= r;
        if (p.nout == 2)
            for (int q = 0; q < p.deg_out[1]; q++) q1 *= r;
        q0 = wave_sum_f64(q0);
        q1 = wave_sum_f64(q1);
        if (half == 0 && (tid & 63) == 0) {
            qred[0][tid >> 6] = q0;
            qred[1][tid >> 6] = q1;
        }
    }
    const int dmax = p.nout == 2 ? max(p.deg_out[0], p.deg_out[1]) : p.deg_out[0];
    const int item = rt;  // (161 of the 256 are items)
    const Item it = kItems.t[item < FG_NI ? item : 0];
    const bool work = item < FG_NI && (int)it.i0 + it.j + it.k <= dmax;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;  // [monomial of the item][set]
    __syncthreads();  // (Ms / s1x are done with)
    if (half == 0) {
        double p0 = wt0, p1 = wt1, py = 1.0, pz = 1.0;
        xc[rt] = x;
        for (int e = 0; e <= dmax; e++) {
            pw[(0 * (FG_P + 1) + e) * FG_PWS + rt] = p0;
            pw[(1 * (FG_P + 1) + e) * FG_PWS + rt] = p1;
            pw[(2 * (FG_P + 1) + e) * FG_PWS + rt] = py;
            pw[(3 * (FG_P + 1) + e) * FG_PWS + rt] = pz;
            p0 *= x;
            p1 *= x;
            py *= y;
            pz *= z;
        }
    }
    __syncthreads();
    if (work) {
        const int q0 = half * (FG_ROWS / 2);
        const double *b0 = pw + (0 * (FG_P + 1) + it.i0) * FG_PWS + q0, *b1 = pw + (1 * (FG_P + 1) + it.i0) * FG_PWS + q0;
        const double *ay = pw + (2 * (FG_P + 1) + it.j) * FG_PWS + q0, *az = pw + (3 * (FG_P + 1) + it.k) * FG_PWS + q0;
        const double *xq = xc + q0;
        if (p.nout == 2) {
#pragma unroll 8
            for (int q = 0; q < FG_ROWS / 2; q++) {
                const double yz = ay[q] * az[q];
                const double u0 = b0[q] * yz, u1 = b1[q] * yz;
                a00 += u0;
                a01 += u1;
                a10 = fma(u0, xq[q], a10);
                a11 = fma(u1, xq[q], a11);
            }
        } else {
#pragma unroll 8
            for (int q = 0; q < FG_ROWS / 2; q++) {
                const double u0 = b0[q] * (ay[q] * az[q]);
                a00 += u0;
                a10 = fma(u0, xq[q], a10);
            }
        }
    }
    // the two halves' sums meet in LDS (the tables are done with): ONE partial per workgroup and set leaves the CU
    __syncthreads();
    double *xch = pw;  // [4][FG_ROWS]
    if (half == 1) {
        xch[0 * FG_ROWS + rt] = a00;
        xch[1 * FG_ROWS + rt] = a01;
        xch[2 * FG_ROWS + rt] = a10;
        xch[3 * FG_ROWS + rt] = a11;
    }
    __syncthreads();
    if (half == 0 && item < FG_NI) {
        a00 += xch[0 * FG_ROWS + rt];
        a01 += xch[1 * FG_ROWS + rt];
        a10 += xch[2 * FG_ROWS + rt];
        a11 += xch[3 * FG_ROWS + rt];
        double *out0 = p.mout + ((size_t)(0 * p.b + bi) * p.chout + blockIdx.x) * FG_NMX;
        out0[it.t0] = a00;
        if (it.cnt == 2) out0[it.t1] = a10;
        if (p.nout == 2) {
            double *out1 = p.mout + ((size_t)(1 * p.b + bi) * p.chout + blockIdx.x) * FG_NMX;
            out1[it.t0] = a01;
            if (it.cnt == 2) out1[it.t1] = a11;
        }
    }
    if (half == 0 && rt == FG_NI) {  // (an idle lane of the item pass)
        for (int s = 0; s < p.nout; s++)
            p.mout[((size_t)(s * p.b + bi) * p.chout + blockIdx.x) * FG_NMX + FG_NM] = (qred[s][0] + qred[s][1]) + (qred[s][2] + qred[s][3]);
    }
}

int launch(const Step &p, hipStream_t s) {
    // (per call, not once: the attribute belongs to the calling thread's current device)
    RF_HIP(hipFuncSetAttribute((const void *)fgt_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FG_LDS));
    RF_LAUNCH("am_fgt", fgt_step_kernel, dim3(rf::ceil_div(p.nrows, FG_ROWS), p.b), dim3(FG_TPB), FG_LDS, s, p);
    return RF_OK;
}

}  // namespace

namespace rfe {

size_t fgt_workspace_bytes(int b, int nmax) {
    if (b <= 0 || nmax <= 0) return 0;
    return align256(sizeof(Geom) * (size_t)b) + 2 * align256(sizeof(double) * 2 * (size_t)b * chunks_of(nmax) * FG_NMX);
}

void *fgt_geom(void *ws, int b, int nmax) { return (void *)view(ws, b, nmax).geom; }

// total degree for sharpness a.  (Level -0.25: degree 6 met the certificate's 1e-7 only on box-filling clouds -- 1.3e-7 true error on
// corner clusters -- degree 8 holds 3e-10 there: tools/experiments/fgt_row_bound.py.)
static int degree_for(double a) { return a <= 1e-12 ? 0 : (a <= 0.3 ? 8 : FG_P); }

// moments over xyz2 with w = remainR at a: what the first fgt_p3p1 (without a fused P3) evaluates
static int moments_only(int b, int npts, const float *pts, const float *w, size_t stride, double a, double *mout, int chout,
                        const FgtWs &v, hipStream_t s) {
    Step p{};
    p.rows = pts;
    p.nrows = npts;
    p.mode = 0;
    p.v0 = w;
    p.stride = stride;
    p.mout = mout;
    p.chout = chout;
    p.nout = 1;
    p.a_out[0] = a;
    p.deg_out[0] = degree_for(a);
    p.geom = v.geom;
    p.b = b;
    return launch(p, s);
}

// What the workspace holds between the calls of one schedule:
//   after fgt_p3p1(level v):  mom[0] set 0 = over xyz1, w = ratioL_v at a_v                                   (for fgt_p2(v))
//   after fgt_p2(level v):    mom[1] set 0 = over xyz2, w = remainR at a_next; set 1 = w = ratioR_v at a_v      (for fgt_p3p1(v + 1))
int fgt_p3p1(int b, int n, int m, const float *xyz1, const float *xyz2, bool has_p3, double a_prev, double a_cur,
             const float *ratioR_prev, const float *remainR, const float *ratioL_prev, float *remainL, float *ratioL_out,
             size_t stride, void *ws, hipStream_t s) {
    FgtWs v = view(ws, b, n > m ? n : m);
    if (!has_p3) {  // the chain starts here: the moments over xyz2 with w = remainR
        if (int e = moments_only(b, m, xyz2, remainR, stride, a_cur, v.mom[1], chunks_of(m), v, s)) return e;
    }
    Step p{};
    p.rows = xyz1;
    p.nrows = n;
    p.min = v.mom[1];
    p.chin = chunks_of(m);
    p.nin = has_p3 ? 2 : 1;
    p.a_in[0] = a_cur;
    p.a_in[1] = a_prev;
    p.deg_in[0] = degree_for(a_cur);
    p.deg_in[1] = degree_for(a_prev);
    p.cols = xyz2;
    p.ncols = m;
    p.cols_are_set2 = 1;
    p.wcol[0] = remainR;
    p.wcol[1] = ratioR_prev;
    p.mout = v.mom[0];
    p.chout = chunks_of(n);
    p.nout = 1;
    p.a_out[0] = a_cur;
    p.deg_out[0] = degree_for(a_cur);
    p.stride = stride;
    p.mode = 1;
    p.v0 = ratioL_prev;
    p.rem = remainL;
    p.out = ratioL_out;
    p.geom = v.geom;
    p.b = b;
    return launch(p, s);
}

int fgt_p2(int b, int n, int m, const float *xyz1, const float *xyz2, double a_cur, double a_next, const float *ratioL,
           float *remainR, float *ratioR_out, size_t stride, void *ws, hipStream_t s) {
    FgtWs v = view(ws, b, n > m ? n : m);
    Step p{};
    p.rows = xyz2;
    p.nrows = m;
    p.min = v.mom[0];
    p.chin = chunks_of(n);
    p.nin = 1;
    p.a_in[0] = a_cur;
    p.deg_in[0] = degree_for(a_cur);
    p.cols = xyz1;
    p.ncols = n;
    p.cols_are_set2 = 0;
    p.wcol[0] = ratioL;
    p.mout = v.mom[1];
    p.chout = chunks_of(m);
    p.nout = a_next >= 0.0 ? 2 : 0;
    p.a_out[0] = a_next >= 0.0 ? a_next : 0.0;
    p.a_out[1] = a_cur;
    p.deg_out[0] = degree_for(p.a_out[0]);
    p.deg_out[1] = degree_for(a_cur);
    p.stride = stride;
    p.mode = 2;
    p.rem = remainR;
    p.out = ratioR_out;
    p.geom = v.geom;
    p.b = b;
    return launch(p, s);
}

}  // namespace rfe
