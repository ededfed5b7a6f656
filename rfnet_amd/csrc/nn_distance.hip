// nn_distance.hip -- Chamfer nearest-neighbour distance, forward and backward, for gfx950.
//
// Replaces NmDistanceKernel / NmDistanceGradKernel (tf_ops/CD/tf_nndistance_g.cu:4-156).
// Results: bit-identical to oracle/rfops_oracle.c (d2 = fma(dz,dz,fma(dx,dx,dy*dy)),
// lowest index wins ties) -- the reference's two launches compute (b-a)^2 and (a-b)^2, whose
// bits are equal, so ONE evaluation per pair serves both directions.
//
// MI355X design.  The sweep is fp32-VALU bound (20*B*(N+M) bytes of HBM traffic against
// 2*B*N*M directed pairs); measured issue cost on gfx950 is ~2.3 cycles per VOP2
// wave-instruction and packed-fp32 ops buy nothing (tools/ubench/valu_rate.hip), so the only
// lever is instructions per pair.  The reference spends 2 x 9 (3 sub, mul, 2 fma, cmp, 2
// select, per direction).  Here:
//   * ONE SWEEP: each wave owns 64*R points of the larger set ("own", R per lane, in VGPRs)
//     and streams the other set ("candidates") through SGPRs with scalar loads -- they are
//     wave-uniform, so no LDS, no barrier and no VGPR is spent on them, and VALU ops take the
//     coordinate as their SGPR operand.  d2 is computed once (6 ops) and feeds both minima.
//   * own-side minimum: running min VALUE only (v_min3_f32, half an op per pair); one
//     compare+select per 16-candidate CHUNK remembers the chunk that lowered it; the winning
//     chunk is re-scanned once at the end for the first index whose d2 equals the minimum
//     bit-for-bit (same instruction sequence => same bits).
//   * candidate-side minimum: in-lane min3 over the R own points (half an op per pair), then a
//     wave64 reduce-scatter butterfly over 32 candidates at a time (ds_bpermute + 2 select +
//     min: ~3 ops per candidate, i.e. 3/R per pair) leaves candidate c's minimum over the
//     wave's 64*R points in lane 2c.  Lanes own R CONSECUTIVE points, so the lowest matching
//     index lives in the lowest lane whose in-lane minimum equals the wave minimum: one
//     v_readlane + one v_cmp_eq (its SGPR mask IS the ballot) + s_ff1 per candidate finds that
//     lane.  (value, lane) go to a per-(own block) partial array; a resolve kernel takes, per
//     candidate, the first own block with the smallest value (strict '<' in block order =
//     lowest index on ties) and re-scans just that lane's R points for the first exact match.
//   ~7.9 VALU ops per (B*N*M) pair instead of 18.  All merges are order-fixed: no atomics,
//   deterministic.
#include <stdlib.h>

#include "common.hpp"
#include "nn_dense.hpp"
#include "nn_pruned.hpp"

namespace {

constexpr int TPB = 256;  // 4 independent waves per workgroup
constexpr int CH = 16;    // candidates per chunk (own-side argmin granularity)
constexpr int CG = 32;    // candidates per reduce-scatter group (2 chunks)
constexpr int PADQ = 512; // packed arrays are padded to a multiple of this many points

// Buffers (separate __restrict__ kernel parameters so the candidate loads can be scalar loads):
//   own      packed (b, no_pad, 3), padded with -inf
//   cand     packed (b, nc_pad, 3), padded with +inf
//   row_dist/row_idx  [nsplit][b][no] partial (or the final outputs if nsplit == 1)
//   colpart  [oblocks][b][nc] minimum over one own block
struct Sweep {
    int b, no, nc, no_pad, nc_pad;
    int oblocks;  // ceil(no / (64*R))
    int nsplit;   // candidate range splits
    int span;     // candidates per split (multiple of CG)
    int wgm;      // 1: the 4 waves of a workgroup are 4 consecutive splits of one own block and merge
                  //    their own-side results in LDS (row partial slots = nsplit / 4)
    int rslots_final;  // 1: the sweep writes the FINAL own-side outputs (one partial slot: no row merge follows)
};

// Re-pack both clouds in ONE launch: blocks [0, nblk_own) pack the own set (padded with -inf),
// the rest the candidate set (padded with +inf).
__global__ void pack_kernel(int b, int n_own, int n_own_pad, const float *__restrict__ src_own,
                            float *__restrict__ dst_own, int nblk_own, int n_cand, int n_cand_pad,
                            const float *__restrict__ src_cand, float *__restrict__ dst_cand) {
    const bool is_cand = (int)blockIdx.x >= nblk_own;
    const int n = is_cand ? n_cand : n_own, n_pad = is_cand ? n_cand_pad : n_own_pad;
    const float *__restrict__ src = is_cand ? src_cand : src_own;
    float *__restrict__ dst = is_cand ? dst_cand : dst_own;
    const float padval = is_cand ? INFINITY : -INFINITY;
    long g = (long)(blockIdx.x - (is_cand ? nblk_own : 0)) * blockDim.x + threadIdx.x;
    long total = (long)b * n_pad;
    if (g >= total) return;
    long bi = g / n_pad;
    int j = (int)(g - bi * n_pad);
    float x = padval, y = 0.f, z = 0.f;
    if (j < n) {
        const float *p = src + (bi * n + j) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    dst[g * 3 + 0] = x;
    dst[g * 3 + 1] = y;
    dst[g * 3 + 2] = z;
}

// running minimum as ONE v_min3_f32.  Written as asm so that LLVM cannot re-associate the chain
// of minima over a chunk into a tree evaluated at the end of the chunk (minnum is exactly
// associative, so it does -- and then keeps all 16 x R distances of the chunk alive: spills).
__device__ __forceinline__ float min3_acc(float acc, float a, float b) {
    asm("v_min3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

template <int R>
__device__ __forceinline__ float min_over(const float (&d)[R]) {
    float m = d[0];
#pragma unroll
    for (int r = 1; r + 1 < R; r += 2) m = fminf(fminf(m, d[r]), d[r + 1]);
    if ((R & 1) == 0) m = fminf(m, d[R - 1]);
    return m;
}

// reduce-scatter of v[0..31] over the 64 lanes: returns, in lanes 2c and 2c+1, the minimum over
// all lanes of v[c].  Step (HALF, BIT): lanes with BIT clear keep v[0..HALF) and send
// v[HALF..2*HALF) to lane^BIT; the others the opposite.  Template recursion keeps every register
// index a compile-time constant (a runtime-indexed array would go to scratch).
template <int HALF, int BIT>
__device__ __forceinline__ void rs_step(float (&v)[CG], int lane) {
    const bool hi = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        float keep = hi ? v[i + HALF] : v[i];
        float send = hi ? v[i] : v[i + HALF];
        float recv = __shfl_xor(send, BIT, 64);
        v[i] = fminf(keep, recv);
    }
}

// The first step writes into a fresh half-size array, so the caller's v[] (the in-lane minima
// that the winning-lane pass compares against) survives without a 32-register copy.
__device__ __forceinline__ float reduce_scatter32(const float (&v)[CG], int lane) {
    float w[CG];
    {
        const bool hi = (lane & 32) != 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            float keep = hi ? v[i + 16] : v[i];
            float send = hi ? v[i] : v[i + 16];
            w[i] = fminf(keep, __shfl_xor(send, 32, 64));
        }
    }
    rs_step<8, 16>(w, lane);
    rs_step<4, 8>(w, lane);
    rs_step<2, 4>(w, lane);
    rs_step<1, 2>(w, lane);
    return fminf(w[0], __shfl_xor(w[0], 1, 64));
}

// For candidate I of the group (its wave minimum sits in lane 2I of `cmin`): the lowest lane whose
// in-lane minimum orig[I] equals it, written into lane 2I of `wl`.  The minimum is fetched back
// with ds_bpermute (LDS crossbar, no VALU slot and no VALU->SGPR hazard), compared with
// v_cmp_eq_f32 -- whose 64-bit result mask IS the ballot -- then s_ff1 + v_writelane.
// Some lane always matches (the minimum is attained), so the mask is never 0 for finite/inf data.
template <int I>
__device__ __forceinline__ void who_has_it(const float (&orig)[CG], float cmin, int &wl) {
    const float m = __shfl(cmin, 2 * I, 64);
    const unsigned long long mask = __ballot(orig[I] == m);
    const int l = __builtin_ctzll(mask);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(wl) : "s"(l), "n"(2 * I));
    if constexpr (I + 1 < CG) who_has_it<I + 1>(orig, cmin, wl);
}

// COLS = false: only the own side is wanted (rf_nn_distance_dir with one direction): the in-lane
// candidate minima, the reduce-scatter and the winning-lane pass are compiled out (~6.5 VALU per
// pair instead of ~7.9).
template <int R, bool COLS>
__global__ __launch_bounds__(TPB) void nn_sweep_kernel(Sweep a, const float *__restrict__ own_all,
                                                       const float *__restrict__ cand_all,
                                                       float *__restrict__ row_dist,
                                                       int *__restrict__ row_idx,
                                                       float *__restrict__ colpart,
                                                       unsigned char *__restrict__ collane) {
    const int lane = threadIdx.x & 63;
    // wave-uniform work decomposition (readfirstlane => SGPRs => scalar loads of candidates)
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (TPB / 64) + (threadIdx.x >> 6));
    const int split = w % a.nsplit;
    const int ob = (w / a.nsplit) % a.oblocks;
    const int bi = w / (a.nsplit * a.oblocks);
    if (bi >= a.b) return;

    const float *__restrict__ own = own_all + (size_t)bi * a.no_pad * 3;
    const float *__restrict__ cand = cand_all + (size_t)bi * a.nc_pad * 3;

    float ax[R], ay[R], az[R], best[R];
    int bchunk[R];
    const int c_begin = split * a.span;
    const int c_end = min(a.nc, c_begin + a.span);  // exclusive, real candidates
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = (ob * 64 + lane) * R + r;  // lane-major; < no_pad by construction
        ax[r] = own[j * 3 + 0];
        ay[r] = own[j * 3 + 1];
        az[r] = own[j * 3 + 2];
        best[r] = INFINITY;
        bchunk[r] = c_begin / CH;
    }
    float *__restrict__ colp = colpart + ((size_t)ob * a.b + bi) * a.nc;
    unsigned char *__restrict__ collp = collane + ((size_t)ob * a.b + bi) * a.nc;

    // Candidates arrive by scalar loads in sub-chunks of SUB points (3*SUB SGPRs), one sub-chunk
    // ahead of the arithmetic: the s_load for sub-chunk s+1 is issued before the VALU work on
    // sub-chunk s, so its latency is covered by ~SUB*7*R instructions.  (The packed arrays carry
    // one extra group of padding so the last prefetch stays in bounds.)
    constexpr int SUB = 8;
    float nb[3 * SUB];
    {
        const float *cp = cand + (size_t)c_begin * 3;
#pragma unroll
        for (int i = 0; i < 3 * SUB; i++) nb[i] = cp[i];
    }
    for (int c0 = c_begin; c0 < c_end; c0 += CG) {
        float colv[CG];
        float cm[R];
#pragma unroll
        for (int sub = 0; sub < CG / SUB; sub++) {
            float cb[3 * SUB];
            // retire the previous prefetch BEFORE issuing the next one: scalar loads return out of
            // order, so the only wait is lgkmcnt(0), and placed after the new s_load it would
            // wait for that one too (no overlap).  0xC07F = lgkmcnt(0) only.
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3 * SUB; i++) cb[i] = nb[i];
            {
                // (clamped: the prefetch issued by the last sub-chunk is never used and must not
                // run past the array when the clouds are read in place, without the padded copy)
                const float *cp = cand + (size_t)min(c0 + (sub + 1) * SUB, a.nc_pad - SUB) * 3;  // uniform -> s_load
#pragma unroll
                for (int i = 0; i < 3 * SUB; i++) nb[i] = cp[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((sub * SUB) % CH == 0) {
#pragma unroll
                for (int r = 0; r < R; r++) cm[r] = INFINITY;
            }
#pragma unroll
            for (int u = 0; u < SUB; u += 2) {
                const float px = cb[u * 3 + 0], py = cb[u * 3 + 1], pz = cb[u * 3 + 2];
                const float qx = cb[u * 3 + 3], qy = cb[u * 3 + 4], qz = cb[u * 3 + 5];
                float d0[R], d1[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    d0[r] = rf::d2_fma(px - ax[r], py - ay[r], pz - az[r]);
                    d1[r] = rf::d2_fma(qx - ax[r], qy - ay[r], qz - az[r]);
                    cm[r] = min3_acc(cm[r], d0[r], d1[r]);
                }
                if constexpr (COLS) {
                    colv[sub * SUB + u] = min_over<R>(d0);
                    colv[sub * SUB + u + 1] = min_over<R>(d1);
                }
                // keep the scheduler from interleaving candidate pairs (bounds the live set)
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((sub * SUB) % CH == CH - SUB) {
                const int chunk = (c0 + sub * SUB) / CH;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (cm[r] < best[r]) {
                        best[r] = cm[r];
                        bchunk[r] = chunk;
                    }
                }
            }
        }
        if constexpr (COLS) {
            const float cmin = reduce_scatter32(colv, lane);
            // which lane holds it: the lowest lane whose in-lane minimum equals the wave minimum
            int wl = 0;
            who_has_it<0>(colv, cmin, wl);
            const int c = c0 + (lane >> 1);
            if ((lane & 1) == 0 && c < c_end) {
                colp[c] = cmin;
                collp[c] = (unsigned char)wl;
            }
        }
    }

    // own side: first index inside the winning chunk whose d2 equals the minimum.  The loop over
    // the 16 chunk slots is NOT unrolled (descending, so the lowest matching index is kept):
    // unrolling it keeps 48 floats x R in flight and spills.
    const size_t obase = ((size_t)split * a.b + bi) * a.no;
    int k0[R], besti[R];
#pragma unroll
    for (int r = 0; r < R; r++) besti[r] = k0[r] = bchunk[r] * CH;
#pragma unroll 1
    for (int u = CH - 1; u >= 0; u--) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int k = k0[r] + u;  // < nc_pad; padded entries give +inf and are excluded below
            const float *p = cand + (size_t)k * 3;
            float d = rf::d2_fma(p[0] - ax[r], p[1] - ay[r], p[2] - az[r]);
            if (k < c_end && d == best[r]) besti[r] = k;
        }
    }
    if (a.wgm) {
        // The workgroup's 4 waves hold the same 64*R own points against 4 consecutive candidate
        // spans: merge them here, in span order with strict '<' (lowest index wins ties), and write
        // ONE result per point instead of four partials (C2: the own side is final after this).
        static_assert(R % 4 == 0, "float4 staging");
        __shared__ float md[TPB / 64][64 * R];
        __shared__ int mi[TPB / 64][64 * R];
        const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int r = 0; r < R; r += 4) {
            *(float4 *)&md[wib][lane * R + r] = make_float4(best[r], best[r + 1], best[r + 2], best[r + 3]);
            *(int4 *)&mi[wib][lane * R + r] = make_int4(besti[r], besti[r + 1], besti[r + 2], besti[r + 3]);
        }
        __syncthreads();
        const size_t mbase = ((size_t)(split / (TPB / 64)) * a.b + bi) * a.no + (size_t)ob * 64 * R;
        for (int pnt = threadIdx.x; pnt < 64 * R; pnt += TPB) {
            float bd = md[0][pnt];
            int bk = mi[0][pnt];
#pragma unroll
            for (int sp = 1; sp < TPB / 64; sp++) {
                const float d = md[sp][pnt];
                if (d < bd) {
                    bd = d;
                    bk = mi[sp][pnt];
                }
            }
            if (ob * 64 * R + pnt < a.no) {
                if (a.rslots_final) {  // this is the final own-side result: a NaN point gets (NaN, 0), as in the reference
                    const float *q = own + (size_t)(ob * 64 * R + pnt) * 3;
                    if (q[0] != q[0] || q[1] != q[1] || q[2] != q[2]) { bd = NAN; bk = 0; }
                }
                row_dist[mbase + pnt] = bd;
                row_idx[mbase + pnt] = bk;
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int j = (ob * 64 + lane) * R + r;
        if (j < a.no) {
            const bool qnan = a.rslots_final && (ax[r] != ax[r] || ay[r] != ay[r] || az[r] != az[r]);
            row_dist[obase + j] = qnan ? NAN : best[r];
            row_idx[obase + j] = qnan ? 0 : besti[r];
        }
    }
}

// own side: combine the per-split partials in split order; strict '<' keeps the lowest index.
__device__ __forceinline__ void rowmerge_body(long g, const float *__restrict__ pd,
                                              const int *__restrict__ pi, float *__restrict__ dist,
                                              int *__restrict__ idx, int nsplit, long total,
                                              const float *__restrict__ own_all, int no, int no_pad) {
    if (g >= total) return;
    float best = pd[g];
    int besti = pi[g];
    for (int s = 1; s < nsplit; s++) {
        float d = pd[(size_t)s * total + g];
        if (d < best) {
            best = d;
            besti = pi[(size_t)s * total + g];
        }
    }
    // a point with a NaN coordinate: (NaN, 0), as the reference returns (its first candidate is taken
    // unconditionally and nothing compares below NaN, tf_nndistance_g.cu:27-31)
    const long bi = g / no;
    const float *q = own_all + ((size_t)bi * no_pad + (g - bi * no)) * 3;
    if (q[0] != q[0] || q[1] != q[1] || q[2] != q[2]) { best = NAN; besti = 0; }
    dist[g] = best;
    idx[g] = besti;
}

// candidate side: per candidate (one quad of lanes each), the first own block with the smallest
// partial (strict '<' in block order), then the lowest index among that block's winning lane's R
// consecutive points with an exactly equal d2.
// The same launch also finishes the own side: blocks beyond the candidate blocks run rowmerge.
template <int R>
__global__ __launch_bounds__(TPB) void nn_resolve_kernel(Sweep a, const float *__restrict__ own_all,
                                                         const float *__restrict__ cand_all,
                                                         const float *__restrict__ colpart,
                                                         const unsigned char *__restrict__ collane,
                                                         float *__restrict__ dist, int *__restrict__ idx,
                                                         int nblk_col, const float *__restrict__ row_pd,
                                                         const int *__restrict__ row_pi,
                                                         float *__restrict__ row_dist,
                                                         int *__restrict__ row_idx, int rslots) {
    if ((int)blockIdx.x >= nblk_col) {
        rowmerge_body((long)(blockIdx.x - nblk_col) * TPB + threadIdx.x, row_pd, row_pi, row_dist, row_idx,
                      rslots, (long)a.b * a.no, own_all, a.no, a.no_pad);
        return;
    }
    // 4 lanes (a quad) per candidate: lane q scans the own blocks o = q, q+4, ... (its loads are
    // independent and all in flight), the quad combines (value, block) lexicographically -- the
    // first block with the smallest partial -- and shares the R-point re-scan.  One thread per
    // candidate left the kernel latency-bound on a chain of `oblocks` dependent compares.
    constexpr int CPB = TPB / 4;  // candidates per workgroup
    const int cblocks = nblk_col / a.b;
    const int bi = blockIdx.x / cblocks;
    const int q = threadIdx.x & 3;
    const int c = min((int)(blockIdx.x - bi * cblocks) * CPB + (int)(threadIdx.x >> 2), a.nc - 1);
    const bool writer = q == 0 && (int)(blockIdx.x - bi * cblocks) * CPB + (int)(threadIdx.x >> 2) < a.nc;
    const float *own = own_all + (size_t)bi * a.no_pad * 3;
    const float *cand = cand_all + (size_t)bi * a.nc_pad * 3;
    const float *cp = colpart + (size_t)bi * a.nc + c;
    const size_t ostride = (size_t)a.b * a.nc;
    float best = INFINITY;
    int bblk = 0x7fffffff;
    for (int o = q; o < a.oblocks; o += 4) {
        const float v = cp[(size_t)o * ostride];
        if (v < best || bblk == 0x7fffffff) {  // strict '<' in block order; the first one always taken
            best = v;
            bblk = o;
        }
    }
#pragma unroll
    for (int x = 1; x <= 2; x <<= 1) {
        const float ov = __shfl_xor(best, x, 64);
        const int ob = __shfl_xor(bblk, x, 64);
        if (ob != 0x7fffffff && (bblk == 0x7fffffff || ov < best || (ov == best && ob < bblk))) {
            best = ov;
            bblk = ob;
        }
    }
    // lane 0 of the quad holds exactly what the sequential scan would (NaN partials included);
    // the whole quad re-scans for ITS result
    best = __shfl(best, threadIdx.x & ~3, 64);
    bblk = __shfl(bblk, threadIdx.x & ~3, 64);
    // (a NaN candidate leaves no lane matching its NaN minimum: the recorded lane is then 64 -- masked
    // here so that the re-scan stays inside the arrays; its result is overridden below)
    const int wl = collane[(size_t)bblk * ostride + (size_t)bi * a.nc + c] & 63;
    const float cx = cand[c * 3 + 0], cy = cand[c * 3 + 1], cz = cand[c * 3 + 2];
    const int j0 = (bblk * 64 + wl) * R;
    static_assert(R % 4 == 0, "the quad shares the re-scan");
    int found = 0x7fffffff;
#pragma unroll
    for (int r = R / 4 - 1; r >= 0; r--) {
        const int j = j0 + q * (R / 4) + r;
        float d = rf::d2_fma(cx - own[j * 3 + 0], cy - own[j * 3 + 1], cz - own[j * 3 + 2]);
        if (j < a.no && d == best) found = j;
    }
    found = min(found, __shfl_xor(found, 1, 64));
    found = min(found, __shfl_xor(found, 2, 64));
    if (found == 0x7fffffff) found = j0 < a.no ? j0 : 0;  // all-inf case: block 0, lane 0 -> index 0
    if (cx != cx || cy != cy || cz != cz) {  // a NaN point: (NaN, 0), as in the reference
        best = NAN;
        found = 0;
    }
    if (writer) {
        dist[(size_t)bi * a.nc + c] = best;
        idx[(size_t)bi * a.nc + c] = found;
    }
}

// Backward.  grad_own[j] = 2*gd_own[j]*(own_j - other_{idx_own[j]})            (own term)
//                        - sum_{k: idx_other[k]==j} 2*gd_other[k]*(other_k - own_j)   (scatter)
// The reference zero-fills both outputs and issues 6 global atomicAdd per point
// (tf_nndistance_g.cu:131-156).  Scattered float atomics run ~17x below the coalesced atomic
// rate on MI355X (one lane per row), so here the scatter is privatised in LDS: one workgroup
// owns a TILE of destination points of one batch element, sweeps ALL sources of the other set
// (coalesced idx/gd/xyz reads), accumulates the hits with ds_add_f32, adds the own term and
// writes the tile with plain coalesced stores: no memset, no global atomics.  The tile size is
// chosen per direction (64..2048 destination points) so that b * tiles >= 256 workgroups; only
// when even 64-point tiles leave the chip empty (tiny batches) are the sources split over
// `slices` workgroups per tile, which then flush their tiles with coalesced global atomics
// (256 B per wave-instruction: the full atomic rate) onto a zero-filled output.
// (Workgroups stay in dispatch order, i.e. the tiles of one batch element on 8 different XCDs: a
// batch-major, XCD-contiguous order -- what helps the culled sweep -- measured 0.021 -> 0.025 ms here,
// 16 workgroups hammering the same index lines of one L2.)
// Arithmetic as the reference: g = gd+gd; v = (a-b)*g rounded on its own; plain adds.
constexpr int GT = 2048;    // max destination points per tile (24 KiB of LDS)
constexpr int GTPB = 1024;

struct GradDir {
    const float *dst_xyz;   // (b, nd, 3) the set whose gradient this is
    const float *src_xyz;   // (b, ns, 3) the other set
    const float *gd_dst;    // (b, nd) upstream grad of the dst set's distances
    const int *idx_dst;     // (b, nd) nn of each dst point in src
    const float *gd_src;    // (b, ns)
    const int *idx_src;     // (b, ns) nn of each src point in dst
    float *grad;            // (b, nd, 3)
    int nd, ns, gt, tiles, slices;  // gt = destination points per tile (<= GT)
    // LOSS mode (fused Chamfer loss backward): the upstream grads are not arrays but
    // gd[j] = (gl[bi][col] / npts) * 0.5 / sqrt(dist[j]), from the forward's own distances
    const float *dist_dst, *dist_src;  // (b, nd), (b, ns)
    int col_dst, col_src;              // column of gl (b, 2) for the dst / src direction
    int has_own, has_scatter;          // a direction that was not computed contributes nothing
};
struct GradArgs {
    GradDir d[2];
    int b, nblk0;
    const float *gl;  // LOSS mode: (b, 2) upstream grads of the per-sample mean-sqrt losses
};

template <bool LOSS>
__global__ __launch_bounds__(GTPB) void nn_grad_kernel(GradArgs a) {
    // the tile's sums in DOUBLE: ds_add_f64 runs at 18 lane-operations per ns and CU, ds_add_f32 at 0.8 (tools/ubench/lds_atomic_rate.hip)
    typedef double acc_t;
    __shared__ acc_t acc[GT * 3];
    int bid = blockIdx.x;
    const int which = bid >= a.nblk0;
    if (which) bid -= a.nblk0;
    const GradDir &D = a.d[which];
    const int slice = bid % D.slices;
    const int tile = (bid / D.slices) % D.tiles;
    const int bi = bid / (D.slices * D.tiles);
    const int j0 = tile * D.gt;
    const int jn = min(D.gt, D.nd - j0);
    const float *__restrict__ dxyz = D.dst_xyz + (size_t)bi * D.nd * 3;
    const float *__restrict__ sxyz = D.src_xyz + (size_t)bi * D.ns * 3;
    for (int i = threadIdx.x; i < jn * 3; i += GTPB) acc[i] = (acc_t)0;
    // own term first: its idx -> gather chain is independent of the scatter scan below, so the two
    // dependent-load chains overlap instead of running back to back (the kernel is latency-bound)
    constexpr int OWN = GT * 3 / GTPB;
    float own[OWN];
    {
        const int *__restrict__ id = D.idx_dst + (size_t)bi * D.nd;
        const float *__restrict__ gdd = (LOSS ? D.dist_dst : D.gd_dst) + (size_t)bi * D.nd;
        const float sc = LOSS ? a.gl[bi * 2 + D.col_dst] / (float)D.nd : 0.f;
#pragma unroll
        for (int u = 0; u < OWN; u++) {
            const int i = threadIdx.x + u * GTPB;
            own[u] = 0.f;
            if (i < jn * 3 && D.has_own) {
                const int j = i / 3, c = i - j * 3;
                const int k = id[j0 + j];
                const float gd = LOSS ? sc * 0.5f / sqrtf(gdd[j0 + j]) : gdd[j0 + j];
                const float g = gd + gd;
                own[u] = (dxyz[(size_t)(j0 + j) * 3 + c] - sxyz[(size_t)k * 3 + c]) * g;
            }
        }
    }
    const int *__restrict__ is = D.idx_src + (size_t)bi * D.ns;
    const float *__restrict__ gs = (LOSS ? D.dist_src : D.gd_src) + (size_t)bi * D.ns;
    const float scs = LOSS ? a.gl[bi * 2 + D.col_src] / (float)D.ns : 0.f;
    const int per = (D.ns + D.slices - 1) / D.slices;
    const int k_end = D.has_scatter ? min(D.ns, (slice + 1) * per) : 0;
    constexpr int SU = 4;  // sources per thread in flight
    int jj[SU];
    int kb = slice * per + threadIdx.x;
#pragma unroll
    for (int u = 0; u < SU; u++) {  // first batch of source indices: issued before the barrier
        const int k = kb + u * GTPB;
        jj[u] = k < k_end ? is[k] - j0 : -1;
    }
    __syncthreads();
    while (kb < k_end) {
        int nj[SU];
        const int kn = kb + GTPB * SU;
#pragma unroll
        for (int u = 0; u < SU; u++) {  // next batch, in flight while this one scatters
            const int k = kn + u * GTPB;
            nj[u] = k < k_end ? is[k] - j0 : -1;
        }
        // The loads of a hit are issued for every lane, with the address clamped to a valid source
        // for the lanes that have none (they all read the same line): no branch around a load, so
        // the SU x 7 loads of a batch are in flight together instead of one hit after the other.
        float hg[SU], hs[SU][3], hd[SU][3];
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int j = jj[u];
            const bool hit = j >= 0 && j < jn;
            const int k = hit ? kb + u * GTPB : slice * per;
            const int jd = hit ? j0 + j : j0;
            hg[u] = gs[k];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                hs[u][c] = sxyz[(size_t)k * 3 + c];
                hd[u][c] = dxyz[(size_t)jd * 3 + c];
            }
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int j = jj[u];
            if (j >= 0 && j < jn) {
                const float gd = LOSS ? scs * 0.5f / sqrtf(hg[u]) : hg[u];
                const float g = gd + gd;
#pragma unroll
                for (int c = 0; c < 3; c++) atomicAdd(&acc[j * 3 + c], (acc_t)-((hs[u][c] - hd[u][c]) * g));
            }
        }
#pragma unroll
        for (int u = 0; u < SU; u++) jj[u] = nj[u];
        kb = kn;
    }
    __syncthreads();
    float *out = D.grad + ((size_t)bi * D.nd + j0) * 3;
#pragma unroll
    for (int u = 0; u < OWN; u++) {
        const int i = threadIdx.x + u * GTPB;
        if (i < jn * 3) {
            if (D.slices == 1) {
                out[i] = own[u] + (float)acc[i];
            } else {
                atomicAdd(&out[i], slice == 0 ? own[u] + (float)acc[i] : (float)acc[i]);
            }
        }
    }
}

constexpr int RR = 8;  // own points per lane (16 drops to 2 waves/SIMD and is slower)

struct Plan {
    bool swap;  // own = xyz2 (the larger set) when true
    int no, nc, no_pad, nc_pad, oblocks, nsplit, span;
    int wgm, rslots;  // in-workgroup merge of 4 splits; own-side partial slots left for nn_resolve
    size_t off_own, off_cand, off_rowd, off_rowi, off_col, off_coll, bytes;
};

int round_up(long v, int q) { return (int)((v + q - 1) / q * q); }

// dirs: bit 0 = direction 1 (nearest neighbour of every xyz1 point in xyz2 -> dist1/idx1), bit 1 =
// direction 2.  With one direction the wanted set is the "own" side whatever its size (the own side
// is the per-point minimum over all candidates) and the column half of the sweep is compiled out.
Plan make_plan(int b, int n, int m, int dirs = 3) {
    Plan p;
    p.swap = dirs == 3 ? m > n : dirs == 2;
    p.no = p.swap ? m : n;
    p.nc = p.swap ? n : m;
    p.no_pad = round_up(p.no, PADQ);
    p.nc_pad = round_up(p.nc, PADQ);
    p.oblocks = rf::ceil_div(p.no, 64 * RR);
    // Split the candidate range so that (measured in round 1 with an environment-knob A/B tool, one device; the knobs are compile-time now: tools/build_variant.py + tools/ab_variants.py):
    //   * at least one residency round exists: 4096 waves (118 VGPRs -> 4 per SIMD x 1024 SIMDs);
    //   * a wave's span is ~1024 candidates when the set is large (16384 x 16384: 1.28 ms at
    //     span 1024 vs 1.36 at 4096 and 1.42 at 256), but never below 128.
    long base = (long)b * p.oblocks;
    const long target = 4096;
    int want = (int)((target + base - 1) / (base > 0 ? base : 1));
    if (want < rf::ceil_div(p.nc, 1024)) want = rf::ceil_div(p.nc, 1024);
    int maxs = p.nc / 128 > 0 ? p.nc / 128 : 1;
    int s = want < 1 ? 1 : (want > maxs ? maxs : want);
    p.span = round_up(rf::ceil_div(p.nc, s), CG);
    p.nsplit = rf::ceil_div(p.nc, p.span);
    p.wgm = (p.nsplit % (TPB / 64) == 0) ? 1 : 0;
    p.rslots = p.wgm ? p.nsplit / (TPB / 64) : p.nsplit;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    p.off_own = take(((size_t)b * p.no_pad + CG) * 12);
    p.off_cand = take(((size_t)b * p.nc_pad + CG) * 12);  // + one group: prefetch overrun
    p.off_rowd = take(p.rslots > 1 ? (size_t)p.rslots * b * p.no * 4 : 0);
    p.off_rowi = take(p.rslots > 1 ? (size_t)p.rslots * b * p.no * 4 : 0);
    p.off_col = take(dirs == 3 ? (size_t)p.oblocks * b * p.nc * 4 : 0);
    p.off_coll = take(dirs == 3 ? (size_t)p.oblocks * b * p.nc : 0);
    p.bytes = off;
    return p;
}

// When the culled sweep (nn_pruned.hip) beats the dense one.  Its cost grows with b * (n + m) (a
// few hundred instructions per point) plus a sort whose duration depends on the larger cloud
// (13 us at 1024 points, 29 us at 16384 with one workgroup per cloud; beyond 16384 points the
// cloud no longer fits the registers and the sort costs ~4 ns per point); the dense sweep's with
// b*n*m at 2.5-6e12 pairs/s.  Thresholds from tools/ab_modes.py + tools/ab_culled.py on MI355X
// (randn clouds), e.g. 1 x 4096^2 1.2x, 32 x 3000 x 1024 1.1x, 4 x 3000 x 16384 1.5x, 8 x 8192^2
// 2.2x, 32 x 2048 x 16384 2.2x, 32 x 16384^2 8x, 1 x 65536^2 2.8x; dense stays ahead at
// 128 x 1024^2, 16 x 700 x 20000, 2 x 65536 x 4096 (round 4: 32 x 512 x 16384 now goes to the culled sweep).
bool culled_pays(int b, int n, int m) {
    const int lo = n < m ? n : m, hi = n < m ? m : n;
    if (!rfp::pruned_supported(b, n, m) || lo < 512 || (long)n * m < (1L << 21)) return false;  // (round 4: 512, was 1024 -- 32 x 512 x 16384 culled 0.060 vs dense 0.083 ms)
    const long pairs = (long)b * n * m;
    if (hi <= 4096) return pairs >= (1L << 24);
    if (hi <= 16384) return pairs >= (1L << 27);
    return pairs >= 44000L * hi;
}

}  // namespace

namespace rfd {

// RF_NN_AUTO -> the sweep this shape gets (callers pin one with rf_nn_distance_mode).
int resolve_mode(int b, int n, int m, int mode) {
    if (mode != RF_NN_AUTO) return mode;
    return culled_pays(b, n, m) ? RF_NN_CULLED : RF_NN_DENSE;
}

size_t dense_workspace_bytes(int b, int n, int m, int dirs) { return make_plan(b, n, m, dirs).bytes; }

// The dense sweep (every pair evaluated).  dirs as above; outputs of a direction that is not
// wanted may be NULL.
int dense_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                      float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s, int dirs) {
    Plan p = make_plan(b, n, m, dirs);
    if (workspace_bytes < p.bytes) return RF_EWORKSPACE;
    const bool cols = dirs == 3;
    char *w = (char *)workspace;
    float *own_p = (float *)(w + p.off_own), *cand_p = (float *)(w + p.off_cand);
    const float *own_src = p.swap ? xyz2 : xyz1, *cand_src = p.swap ? xyz1 : xyz2;
    float *own_dist = p.swap ? dist2 : dist1, *cand_dist = p.swap ? dist1 : dist2;
    int *own_idx = p.swap ? idx2 : idx1, *cand_idx = p.swap ? idx1 : idx2;

    // Clouds whose sizes already fit the tiling (own a multiple of 64*R, candidates a multiple of
    // the 32-candidate group) are swept in place; otherwise both are re-packed with -inf / +inf
    // padding (pack_kernel).  BASELINE's shapes (2048, 16384) take the in-place path.
    const bool in_place = (p.no % (64 * RR) == 0) && (p.nc % CG == 0);
    if (in_place) {
        own_p = const_cast<float *>(own_src);
        cand_p = const_cast<float *>(cand_src);
        p.no_pad = p.no;
        p.nc_pad = p.nc;
    } else {
        const int nb_own = rf::ceil_div((long)b * p.no_pad, 256), nb_cand = rf::ceil_div((long)b * p.nc_pad, 256);
        RF_LAUNCH("nn_pack", pack_kernel, dim3(nb_own + nb_cand), dim3(256), 0, s, b, p.no, p.no_pad, own_src,
                  own_p, nb_own, p.nc, p.nc_pad, cand_src, cand_p);
    }

    Sweep a;
    float *row_dist = p.rslots > 1 ? (float *)(w + p.off_rowd) : own_dist;
    int *row_idx = p.rslots > 1 ? (int *)(w + p.off_rowi) : own_idx;
    float *colpart = (float *)(w + p.off_col);
    unsigned char *collane = (unsigned char *)(w + p.off_coll);
    a.b = b; a.no = p.no; a.nc = p.nc; a.no_pad = p.no_pad; a.nc_pad = p.nc_pad;
    a.oblocks = p.oblocks; a.nsplit = p.nsplit; a.span = p.span; a.wgm = p.wgm;
    a.rslots_final = p.rslots > 1 ? 0 : 1;
    long waves = (long)b * p.oblocks * p.nsplit;
    if (cols) {
        RF_LAUNCH("nn_sweep", (nn_sweep_kernel<RR, true>), dim3(rf::ceil_div(waves, TPB / 64)), dim3(TPB), 0, s, a,
                  (const float *)own_p, (const float *)cand_p, row_dist, row_idx, colpart, collane);
    } else {
        RF_LAUNCH("nn_sweep_1dir", (nn_sweep_kernel<RR, false>), dim3(rf::ceil_div(waves, TPB / 64)), dim3(TPB), 0, s,
                  a, (const float *)own_p, (const float *)cand_p, row_dist, row_idx, colpart, collane);
    }
    {
        const int cblocks = rf::ceil_div(p.nc, TPB / 4);  // 4 lanes per candidate
        const int nblk_col = cols ? cblocks * b : 0;
        const int nblk_row = p.rslots > 1 ? rf::ceil_div((long)b * p.no, TPB) : 0;
        if (nblk_col + nblk_row > 0) {
            RF_LAUNCH("nn_resolve", nn_resolve_kernel<RR>, dim3(nblk_col + nblk_row), dim3(TPB), 0, s, a,
                      (const float *)own_p, (const float *)cand_p, (const float *)colpart,
                      (const unsigned char *)collane, cand_dist, cand_idx, nblk_col, (const float *)row_dist,
                      (const int *)row_idx, own_dist, own_idx, p.rslots);
        }
    }
    return RF_OK;
}

}  // namespace rfd

extern "C" {

size_t rf_nn_distance_workspace_bytes(int b, int n, int m) {
    return rf_nn_distance_mode_workspace_bytes(b, n, m, RF_NN_AUTO);
}

size_t rf_nn_distance_mode_workspace_bytes(int b, int n, int m, int mode) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    mode = rfd::resolve_mode(b, n, m, mode);
    return mode == RF_NN_CULLED ? rfp::pruned_workspace_bytes(b, n, m) : rfd::dense_workspace_bytes(b, n, m, 3);
}

int rf_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                   int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                   rf_stream_t stream) {
    return rf_nn_distance_mode(b, n, m, xyz1, xyz2, dist1, idx1, dist2, idx2, workspace, workspace_bytes,
                               stream, RF_NN_AUTO, nullptr);
}

int rf_nn_distance_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                        int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                        rf_stream_t stream, int mode, unsigned long long *stats) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (mode != RF_NN_AUTO && mode != RF_NN_DENSE && mode != RF_NN_CULLED) return RF_EINVAL;
    if (b == 0 || (n == 0 && m == 0)) return RF_OK;
    if (n == 0 || m == 0) return RF_EINVAL;  // a nearest neighbour in an empty set is undefined
    if (!xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2 || !workspace) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    mode = rfd::resolve_mode(b, n, m, mode);
    if (mode == RF_NN_CULLED) {
        if (!rfp::pruned_supported(b, n, m)) return RF_EINVAL;
        return rfp::pruned_nn_distance(b, n, m, xyz1, xyz2, dist1, idx1, dist2, idx2, workspace,
                                       workspace_bytes, s, stats, 3);
    }
    return rfd::dense_nn_distance(b, n, m, xyz1, xyz2, dist1, idx1, dist2, idx2, workspace, workspace_bytes, s, 3);
}

int rf_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                        const float *grad_dist1, const int *idx1, const float *grad_dist2,
                        const int *idx2, float *grad_xyz1, float *grad_xyz2, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b > 0 && n > 0 && m > 0 && (!grad_dist1 || !grad_dist2)) return RF_EINVAL;
    const rfd::GradSource g{grad_dist1, grad_dist2, nullptr, nullptr, nullptr};
    return rfd::nn_distance_grad(b, n, m, xyz1, xyz2, g, idx1, idx2, grad_xyz1, grad_xyz2, (hipStream_t)stream);
}

}  // extern "C"

namespace rfd {

int nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const GradSource &src,
                     const int *idx1, const int *idx2, float *grad_xyz1, float *grad_xyz2, hipStream_t s) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || (n == 0 && m == 0)) return RF_OK;
    if (n == 0 || m == 0) {  // no neighbours exist: the gradient of nothing is zero
        if (n) RF_ZERO(grad_xyz1, sizeof(float) * 3 * (size_t)b * n, s);
        if (m) RF_ZERO(grad_xyz2, sizeof(float) * 3 * (size_t)b * m, s);
        return RF_OK;
    }
    const bool loss = src.gl != nullptr;
    // a direction is present when its index array is (loss mode: fidelity_loss has direction 1 only)
    const bool has1 = idx1 != nullptr && (loss ? src.dist1 != nullptr : src.gd1 != nullptr);
    const bool has2 = idx2 != nullptr && (loss ? src.dist2 != nullptr : src.gd2 != nullptr);
    if (!xyz1 || !xyz2 || !grad_xyz1 || !grad_xyz2 || (!has1 && !has2)) return RF_EINVAL;
    if (!loss && (!has1 || !has2)) return RF_EINVAL;
    GradArgs a;
    a.b = b;
    a.gl = src.gl;
    // Tile size per direction: small enough that b * tiles >= 256 workgroups with every workgroup
    // sweeping ALL sources (no slices: plain stores, no memset, no global atomics), down to 64
    // destination points; only when even that leaves the chip empty (tiny batches) are the sources
    // sliced, at least 1024 per slice.
    auto tile_for = [&](int nd) {
        long want = ((long)nd * b + 255) / 256;
        int gt = (int)((want + 63) / 64 * 64);
        return gt < 64 ? 64 : (gt > GT ? GT : gt);
    };
    const int g0 = tile_for(n), g1 = tile_for(m);
    const int t0 = rf::ceil_div(n, g0), t1 = rf::ceil_div(m, g1);
    auto slices_for = [&](int tiles, int ns) {
        int want = rf::ceil_div(256, (long)b * tiles);
        int maxs = ns / 1024 > 0 ? ns / 1024 : 1;
        return want < 1 ? 1 : (want > maxs ? maxs : want);
    };
    const int s0 = slices_for(t0, m), s1 = slices_for(t1, n);
    if (s0 > 1) RF_ZERO(grad_xyz1, sizeof(float) * 3 * (size_t)b * n, s);
    if (s1 > 1) RF_ZERO(grad_xyz2, sizeof(float) * 3 * (size_t)b * m, s);
    a.d[0] = GradDir{xyz1, xyz2, src.gd1, idx1, src.gd2, idx2, grad_xyz1, n, m, g0, t0, s0,
                     src.dist1, src.dist2, 0, 1, has1, has2};
    a.d[1] = GradDir{xyz2, xyz1, src.gd2, idx2, src.gd1, idx1, grad_xyz2, m, n, g1, t1, s1,
                     src.dist2, src.dist1, 1, 0, has2, has1};
    a.nblk0 = b * t0 * s0;
    const int nblk1 = b * t1 * s1;
    if (loss) {
        RF_LAUNCH("nn_grad_loss", nn_grad_kernel<true>, dim3(a.nblk0 + nblk1), dim3(GTPB), 0, s, a);
    } else {
        RF_LAUNCH("nn_grad", nn_grad_kernel<false>, dim3(a.nblk0 + nblk1), dim3(GTPB), 0, s, a);
    }
    return RF_OK;
}

}  // namespace rfd
