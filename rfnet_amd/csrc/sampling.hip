// sampling.hip -- farthest_point_sample, gather_point and its gradient for gfx950.
//
// Replaces farthestpointsamplingKernel / gatherpointKernel / scatteraddpointKernel
// (tf_ops/sampling/tf_sampling_g.cu:105-192).  Indices are bit-exact with
// oracle/rfops_oracle.c, including the reference's tie order, which its 512-thread launch
// defines: largest running min-distance; among equals the smallest (k mod 512); among those
// the smallest k (tf_sampling_g.cu:146,158).
//
// MI355X design: FPS is a chain of m-1 dependent block-wide arg-max reductions -- latency
// bound, one workgroup per cloud.  The reference keeps the running min-distances in global
// memory and re-reads points beyond its 3072-point LDS cache from global every iteration.
// Here one 1024-thread workgroup holds the WHOLE cloud (up to 16384 points) in registers:
// 16 points x (x,y,z,running-min) per lane, so an iteration touches no memory except a
// 16-entry LDS exchange and one scalar load: per-lane scan -> wave maximum by DPP row
// rotations -> the lowest lane holding it publishes (d2,k) in the wave's LDS slot
// (double-buffered: ONE barrier per iteration) -> every 16-lane row re-reduces the slots by
// DPP (max d2, then min tie rank) -> the winner's xyz is re-read with a scalar load.
#include "common.hpp"
#include "group_internal.hpp"
#include "nn_pruned.hpp"

namespace {

constexpr int FPS_MAX_REG_POINTS = 16384;
constexpr int FPS_SORTED_MIN_POINTS = 1024;   // fps_sorted_kernel: 1024 threads x 2, 4, 8 or 16 points per lane

// tie rank of point k under the reference's 512-thread layout: lower is preferred
__device__ __forceinline__ unsigned tie_rank(int k) { return ((unsigned)(k & 511) << 22) | (unsigned)(k >> 9); }

__device__ __forceinline__ unsigned long long make_key(float d2, int k) {
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(0xFFFFFFFFu - tie_rank(k));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), o, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
        unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

struct Slot {
    unsigned long long key;
    float x, y, z;
    int k;
};

// ---- single-instruction helpers.  hipcc wraps fminf/fmaxf in canonicalising v_max x,x and
// expands a float DPP reduction step into mov/nop/mov_dpp/max/max; on the serial path of FPS
// every instruction counts, so these are written as the one instruction they are.  (Inputs are
// never NaN here: distances of finite points, or the -1 / 1e38 sentinels.)
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v = max(v, v rotated by N lanes inside each 16-lane row).  The s_nop covers the
// VALU-write -> DPP-read hazard (the assembler/hazard recogniser does not look inside asm).
#define RF_DPP_STEP(OP, N)                                                                          \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float row_allmax(float v) {
    RF_DPP_STEP("v_max_f32_dpp", 8);
    RF_DPP_STEP("v_max_f32_dpp", 4);
    RF_DPP_STEP("v_max_f32_dpp", 2);
    RF_DPP_STEP("v_max_f32_dpp", 1);
    return v;
}
__device__ __forceinline__ unsigned row_allmin_u(unsigned v) {
    RF_DPP_STEP("v_min_u32_dpp", 8);
    RF_DPP_STEP("v_min_u32_dpp", 4);
    RF_DPP_STEP("v_min_u32_dpp", 2);
    RF_DPP_STEP("v_min_u32_dpp", 1);
    return v;
}
#undef RF_DPP_STEP
// maximum over the wave, uniform: row all-reduce, then the four row results
__device__ __forceinline__ float wave_allmax(float v) {
    v = row_allmax(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return vmax3(vmax3(r0, r1, r2), r3, r3);
}

// NT threads (multiple of 512), PPT points per thread, all register resident.
// Thread t owns k = (t & 511) + 512 * (s * (NT/512) + (t >> 9)), s = 0..PPT-1, so all its points
// share (k mod 512) and ascend with s; inside a wave a lower lane has a lower (k mod 512).
// One iteration:
//   scan      7 VALU per point (3 sub, mul, 2 fma, min) + half a v_max3 for the per-lane maximum
//             VALUE -- no per-point compare/select for the arg-max;
//   wave max  DPP row rotations + 4 v_readlane -> wm (uniform);
//   arg-max   per lane the lowest s with td[s] == the lane's maximum (compare + select per
//             point, overlapping the wave reduction); one v_cmp_eq(mx, wm) gives the lanes
//             holding the wave maximum, the lowest of them is the winner under the reference's
//             tie order and a v_readlane fetches its s;
//   exchange  lane 0 publishes (wm, k) in the wave's LDS slot (double-buffered: ONE barrier per
//             iteration); every 16-lane row re-reduces the <=16 slots by DPP (max d2, then
//             min tie rank); the winner's coordinates come back by a scalar load.
// EMIT: the samples' coordinates go out with their indices (new_xyz = gather_point(inp, out), tf_sampling_g.cu:172-181,
// fused: the winner's coordinates are in scalar registers at that point of every iteration).
template <int NT, int PPT, bool EMIT>
__global__ __launch_bounds__(NT) void fps_reg_kernel(int n, int m, const float *__restrict__ inp,
                                                     int *__restrict__ out, float *__restrict__ new_xyz) {
    constexpr int NW = NT / 64;
    constexpr int HALVES = NT / 512;
    static_assert(NW <= 16, "slot reduction is one 16-lane DPP row");
    __shared__ float slot_d[2][16];
    __shared__ int slot_k[2][16];
    const int bi = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ P = inp + (size_t)bi * n * 3;
    int *__restrict__ O = out + (size_t)bi * m;

    float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        int k = (t & 511) + 512 * (s * HALVES + (t >> 9));
        // (these loads are PPT branches with a wait each -- ~15 us at the head of a 1.12 ms launch at C3.  As clamped loads with selects
        // they are one batch, and the KERNEL is slower, 1.208 vs 1.124 ms: the register allocation of the iteration loop changes)
        if (k < n) {
            px[s] = P[k * 3 + 0];
            py[s] = P[k * 3 + 1];
            pz[s] = P[k * 3 + 2];
            td[s] = 1e38f;
        } else {
            px[s] = py[s] = pz[s] = 0.f;
            td[s] = -1.0f;  // min(d,-1) = -1: below every real distance, like the reference's best=-1
        }
    }
    if (t < 32) {  // unused slots never win
        slot_d[t >> 4][t & 15] = -2.0f;
        slot_k[t >> 4][t & 15] = 0;
    }
    float *__restrict__ NX = EMIT ? new_xyz + (size_t)bi * m * 3 : nullptr;
    if (t == 0) O[0] = 0;
    __syncthreads();
    float ox = P[0], oy = P[1], oz = P[2];  // old = 0
    if (EMIT && t == 0) {
        NX[0] = ox;
        NX[1] = oy;
        NX[2] = oz;
    }
    for (int j = 1; j < m; j++) {
        float mx = -1.0f;
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            td[s] = vmin(rf::d2_fma(px[s] - ox, py[s] - oy, pz[s] - oz), td[s]);
            if (PPT == 1) {
                mx = td[0];
            } else if (s & 1) {
                mx = vmax3(mx, td[s - 1], td[s]);
            }
        }
        // The wave's arg-max under the tie order (lowest lane, then lowest s).  Per lane, the lowest
        // s attaining the lane's OWN maximum: 2 VALU per point, independent of the wave reduction
        // (so it overlaps the DPP latency); then ONE compare finds the lanes holding the wave
        // maximum and a v_readlane fetches the winner's s.  (16 v_cmp_eq + a 48-deep scalar
        // select chain per iteration cost 25 % of the kernel: ablation in DESIGN.md 5.3.)
        int sidx = 0;
#pragma unroll
        for (int s = PPT - 1; s >= 1; s--) sidx = (td[s] == mx) ? s : sidx;
        if (PPT > 1) sidx = (td[0] == mx) ? 0 : sidx;
        const float wm = wave_allmax(mx);
        const unsigned long long hl = __ballot(mx == wm);
        const int wl = hl ? __builtin_ctzll(hl) : 0;
        const int bs = __builtin_amdgcn_readlane(sidx, wl);
        const int wt = wave * 64 + wl;
        const int wk = (wt & 511) + 512 * (bs * HALVES + (wt >> 9));
        const int buf = j & 1;
        if (lane == 0) {
            slot_d[buf][wave] = wm;
            slot_k[buf][wave] = wm >= 0.f ? wk : 0;
        }
        __syncthreads();
        const float sd = slot_d[buf][lane & 15];
        const int sk = slot_k[buf][lane & 15];
        const float gm = row_allmax(sd);
        const unsigned rank = sd == gm ? tie_rank(sk) : 0xFFFFFFFFu;
        const unsigned gr = __builtin_amdgcn_readfirstlane(row_allmin_u(rank));
        const int gk = (int)(((gr & 0x3FFFFFu) << 9) | (gr >> 22));
        ox = P[gk * 3 + 0];  // uniform address: scalar loads
        oy = P[gk * 3 + 1];
        oz = P[gk * 3 + 2];
        if (t == 0) {
            O[j] = gk;
            if (EMIT) {
                NX[j * 3 + 0] = ox;
                NX[j * 3 + 1] = oy;
                NX[j * 3 + 2] = oz;
            }
        }
    }
}

// ---- FPS over the spatially sorted cloud: the same samples, most of the cloud left alone in most iterations (round 5) ----
// The cloud arrives in sort-tile-recursive order (rfp::sort_clouds); a lane owns PPT CONSECUTIVE sorted points -- a region with a
// small box.  A new sample s lowers the running minimum td[p] only if |p - s|^2 < td[p]; with `lmx` the largest td of the lane's
// region and `lb` the box's lower bound on the distance (the same instruction sequence as d2 on the per-axis gaps: lb <= the d2
// of every point in the box, by the monotonicity of fp32 rounding), lb >= lmx proves that nothing in the region changes.  A WAVE
// none of whose 64 regions is touched skips its scan and its reductions: its cached (maximum, tie rank) still stands.
// The reference's tie order (largest td, then smallest k mod 512, then smallest k: tf_sampling_g.cu:105-170 as fps_reg_kernel
// restates it) is kept through tie_rank(ORIGINAL index) of every point, minimised among equal maxima at every level.
// Round 3 built this with 8 waves x 32 points per lane and it lost (1.16 vs 1.12 ms: the one wave that must scan issues alone at
// half the rate of two interleaved ones).  Here 16 waves x 16 points: a touched wave's scan is half as long, the regions are
// tighter, and the wave's rank reduction is taken only when two lanes tie for the maximum.
__device__ __forceinline__ unsigned wave_allmin_u(unsigned v) {
    v = row_allmin_u(v);
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}

template <int NT, int PPT, bool EMIT>
__global__ __launch_bounds__(NT) void fps_sorted_kernel(int n, int m, int npad, const float *__restrict__ inp,
                                                        const int *__restrict__ sorig, int *__restrict__ out,
                                                        float *__restrict__ new_xyz) {
    constexpr int NW = NT / 64;
    static_assert(NW <= 16, "slot reduction is one 16-lane DPP row");
    static_assert(PPT >= 2 && (PPT & (PPT - 1)) == 0, "the slots of a lane are ordered by a bitonic network");
    // a wave's candidate: its maximum and the tie rank of the point that holds it (double-buffered: ONE barrier per iteration).
    // (The candidates' coordinates riding along, so that the winner's come back from LDS instead of a scalar re-read of the
    // cloud, measured SLOWER: 0.963 against 0.844 ms at C3 -- five LDS reads per lane behind the barrier and three more indexed
    // register reads before it cost more than the ~300 cycles of the scalar loads.  A second form -- the winning lane writes its
    // candidate's coordinates to the wave's slot, untouched waves copy theirs over, ONE broadcast read of the winner's slot behind
    // the reduction -- was slower still: 1.015 ms.)
    __shared__ float slot_d[2][16];
    __shared__ unsigned slot_r[2][16];
    __shared__ unsigned short cpos[NT * PPT];  // the c-th real record's sorted position (the sort pads every segment to 64 records)
    __shared__ int wcnt[NW + 1];
    const int bi = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const float *__restrict__ P = inp + (size_t)bi * n * 3;
    const int *__restrict__ SO = sorig + (size_t)bi * npad;
    int *__restrict__ O = out + (size_t)bi * m;
    // compaction of the real records (orig >= 0), once
    {
        int run = 0;
        for (int base = 0; base < npad; base += NT) {
            const int p = base + t;
            const bool real = p < npad && SO[p] >= 0;
            const unsigned long long mk = __ballot(real);
            if (lane == 0) wcnt[wave] = __builtin_popcountll(mk);
            __syncthreads();
            int before = run;
            for (int w = 0; w < wave; w++) before += wcnt[w];
            int tot = 0;
            for (int w = 0; w < NW; w++) tot += wcnt[w];
            if (real) cpos[before + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = (unsigned short)p;
            run += tot;
            __syncthreads();
        }
    }
    float px[PPT], py[PPT], pz[PPT], td[PPT];
    unsigned rk[PPT];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        // Which CHUNK of 64 * PPT consecutive sorted points a wave takes (round 6).  An iteration costs what its BUSIEST SIMD scans, not
        // the sum: wave w sits on SIMD w % 4, and with chunk = wave a new sample's 2.9 touched chunks -- spatial neighbours along and
        // across the sort's slabs -- put 1.43 of them on the busiest SIMD; dealt so that the four chunks of a SIMD lie far apart it is
        // 1.18 (tools/experiments/fps_region_model.py: a local search on one uniform cloud, the same table on another 1.20).  Same
        // device, sort + kernel: 16384 points 0.852 -> 0.812 ms, 12000 0.787 -> 0.757, 8192 0.691 -> 0.668, 6000 +-0; clouds of up to
        // 4096 points (2 or 4 points per lane: other slab counts, other neighbours) lose 1-3 % with this table and keep chunk = wave.
        constexpr int kChunkOfWave[16] = {0, 1, 3, 4, 2, 6, 5, 7, 9, 10, 8, 11, 14, 12, 15, 13};
        const int chunk = (NW == 16 && PPT >= 8) ? kChunkOfWave[wave & 15] : wave;
        const int c = ((chunk << 6) | lane) * PPT + s;
        if (c < n) {
            // (the coordinates from the cloud itself, through the original index: the winners' coordinates are re-read from there
            // every iteration, and this pass is what brings the cloud into this XCD's L2 -- read from the sorted copy, a launch that
            // follows other work paid a miss to HBM per iteration: 0.96 against 0.84 ms behind a ball query)
            const int k = SO[cpos[c]];
            px[s] = P[(size_t)k * 3 + 0];
            py[s] = P[(size_t)k * 3 + 1];
            pz[s] = P[(size_t)k * 3 + 2];
            rk[s] = tie_rank(k);
            lo[0] = fminf(lo[0], px[s]), hi[0] = fmaxf(hi[0], px[s]);
            lo[1] = fminf(lo[1], py[s]), hi[1] = fmaxf(hi[1], py[s]);
            lo[2] = fminf(lo[2], pz[s]), hi[2] = fmaxf(hi[2], pz[s]);
        } else {
            px[s] = py[s] = pz[s] = 0.f;
            rk[s] = 0xFFFFFFFFu;
        }
    }
    // the lane's slots in ascending tie rank (bitonic network, static indices, once): the lowest slot that attains the lane's
    // maximum is then the one the reference's tie order picks, as in fps_reg_kernel -- two VALU per point instead of three
#pragma unroll
    for (int k = 2; k <= PPT; k <<= 1) {
#pragma unroll
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
#pragma unroll
            for (int i = 0; i < PPT; i++) {
                const int l = i ^ jj;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const bool sw = up ? rk[i] > rk[l] : rk[i] < rk[l];
                    const unsigned ra = rk[i], rb = rk[l];
                    const float xa = px[i], xb = px[l], ya = py[i], yb = py[l], za = pz[i], zb = pz[l];
                    rk[i] = sw ? rb : ra, rk[l] = sw ? ra : rb;
                    px[i] = sw ? xb : xa, px[l] = sw ? xa : xb;
                    py[i] = sw ? yb : ya, py[l] = sw ? ya : yb;
                    pz[i] = sw ? zb : za, pz[l] = sw ? za : zb;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < PPT; s++) td[s] = rk[s] == 0xFFFFFFFFu ? -1.0f : 1e38f;  // (padding: below every real distance, like the reference's best = -1)
    // the coordinates in PAIRS of points for the scan: v_pk_add / v_pk_mul / v_pk_fma take two points per instruction (the same
    // IEEE operations in the same order: bit-identical distances).  A touched wave scans alone on its SIMD, one instruction every
    // four cycles whatever it is -- there the packed forms halve the distance work (in a full SIMD they are no faster per element).
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f PX[PPT / 2], PY[PPT / 2], PZ[PPT / 2];
#pragma unroll
    for (int h = 0; h < PPT / 2; h++) {
        PX[h] = v2f{px[2 * h], px[2 * h + 1]};
        PY[h] = v2f{py[2 * h], py[2 * h + 1]};
        PZ[h] = v2f{pz[2 * h], pz[2 * h + 1]};
    }
    // cached: the largest td of the lane's region; the wave's candidate (value, tie rank, coordinates); -2 never wins
    float lmx = td[0];
    float wm = -2.0f;
    unsigned wr = 0xFFFFFFFFu;
    if (t < 32) {
        slot_d[t >> 4][t & 15] = -2.0f;
        slot_r[t >> 4][t & 15] = 0xFFFFFFFFu;
    }
    float *__restrict__ NX = EMIT ? new_xyz + (size_t)bi * m * 3 : nullptr;
    // the samples collect in LDS (cpos is free again: m <= n <= NT * PPT entries) and leave together behind the loop -- indices
    // and, EMIT, coordinates as coalesced stores of all threads.  (Stored by thread 0 inside the loop, the coordinates cost
    // 0.13 ms of a 0.84 ms launch: every barrier waits for the previous iteration's stores.)
    __syncthreads();
    if (t == 0) cpos[0] = 0;
    float ox = P[0], oy = P[1], oz = P[2];  // old = 0
    for (int j = 1; j < m; j++) {
        const float gx = fmaxf(fmaxf(lo[0] - ox, ox - hi[0]), 0.f);
        const float gy = fmaxf(fmaxf(lo[1] - oy, oy - hi[1]), 0.f);
        const float gz = fmaxf(fmaxf(lo[2] - oz, oz - hi[2]), 0.f);
        const float lb = rf::d2_fma(gx, gy, gz);
        if (__builtin_amdgcn_ballot_w64(lb < lmx) != 0ull) {  // (uniform; a NaN sample touches nothing: td stays, as min(NaN, td))
            float mx = -1.0f;
#pragma unroll
            for (int h = 0; h < PPT / 2; h++) {
                const v2f dx = PX[h] - ox, dy = PY[h] - oy, dz = PZ[h] - oz;
                const v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));  // (rf::d2_fma's order)
                td[2 * h] = vmin(d2.x, td[2 * h]);
                td[2 * h + 1] = vmin(d2.y, td[2 * h + 1]);
                mx = vmax3(mx, td[2 * h], td[2 * h + 1]);
            }
            int sidx = 0;  // the lowest slot attaining the lane's maximum = its lowest tie rank
#pragma unroll
            for (int s = PPT - 1; s >= 1; s--) sidx = (td[s] == mx) ? s : sidx;
            sidx = (td[0] == mx) ? 0 : sidx;
            lmx = mx;
            wm = wave_allmax(mx);
            const unsigned long long hl = __ballot(mx == wm);
            int wl = hl ? __builtin_ctzll(hl) : 0;
            if (__builtin_popcountll(hl) > 1) {  // (uniform, rare) several lanes tie for the wave's maximum: the lowest rank among their candidates
                unsigned lr = 0xFFFFFFFFu;
#pragma unroll
                for (int s = 0; s < PPT; s++) lr = min(lr, td[s] == mx ? rk[s] : 0xFFFFFFFFu);
                const unsigned best = wave_allmin_u(mx == wm ? lr : 0xFFFFFFFFu);
                const unsigned long long bl = __ballot(mx == wm && lr == best);
                wl = bl ? __builtin_ctzll(bl) : wl;
            }
            const int bs = __builtin_amdgcn_readlane(sidx, wl);
            // the candidate's rank out of lane wl's slot bs: bs is wave-uniform, so the indexed register read is a single move
            // under s_set_gpr_idx
            wr = (unsigned)__builtin_amdgcn_readlane((int)rk[bs], wl);
            if (wm < 0.f) wr = 0xFFFFFFFFu;  // (a wave of padding only)
        }
        const int buf = j & 1;
        if (lane == 0) {
            slot_d[buf][wave] = wm;
            slot_r[buf][wave] = wr;
        }
        __syncthreads();
        const float sd = slot_d[buf][lane & 15];
        const unsigned sr = slot_r[buf][lane & 15];
        const float gm = row_allmax(sd);
        const unsigned rank = sd == gm ? sr : 0xFFFFFFFFu;
        const unsigned gr = __builtin_amdgcn_readfirstlane(row_allmin_u(rank));
        const int gk = gr == 0xFFFFFFFFu ? 0 : (int)(((gr & 0x3FFFFFu) << 9) | (gr >> 22));
        ox = P[gk * 3 + 0];  // uniform address: scalar loads
        oy = P[gk * 3 + 1];
        oz = P[gk * 3 + 2];
        if (t == 0) cpos[j] = (unsigned short)gk;
    }
    __syncthreads();
    for (int j = t; j < m; j += NT) {
        const int gk = cpos[j];
        O[j] = gk;
        if (EMIT) {
            NX[j * 3 + 0] = P[gk * 3 + 0];
            NX[j * 3 + 1] = P[gk * 3 + 1];
            NX[j * 3 + 2] = P[gk * 3 + 2];
        }
    }
}

// Fallback for clouds beyond the register-resident limit: running min-distances in the
// caller's temp buffer (b*n floats), points re-read from global/L2.  Same selection rule.
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, const float *__restrict__ inp,
                                                       float *__restrict__ temp, int *__restrict__ out) {
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ Slot slots[2][NW];
    const int bi = blockIdx.x;
    const int t = threadIdx.x;
    const float *P = inp + (size_t)bi * n * 3;
    float *T = temp + (size_t)bi * n;
    int *O = out + (size_t)bi * m;
    for (int k = t; k < n; k += NT) T[k] = 1e38f;
    if (t == 0) O[0] = 0;
    float ox = P[0], oy = P[1], oz = P[2];
    for (int j = 1; j < m; j++) {
        float best = -1.0f;
        int bk = 0;
        for (int k = (t & 511) + 512 * (t >> 9); k < n; k += NT) {
            float d = rf::d2_fma(P[k * 3] - ox, P[k * 3 + 1] - oy, P[k * 3 + 2] - oz);
            float d2 = fminf(d, T[k]);
            T[k] = d2;
            if (d2 > best) {
                best = d2;
                bk = k;
            }
        }
        const unsigned long long key = best >= 0.f ? make_key(best, bk) : 0ull;
        const unsigned long long wmax = wave_max_u64(key);
        const int buf = j & 1;
        if (key == wmax && (key != 0ull || (t & 63) == 0)) {
            Slot sl;
            sl.key = key;
            int kk = key != 0ull ? bk : 0;
            sl.x = P[kk * 3]; sl.y = P[kk * 3 + 1]; sl.z = P[kk * 3 + 2];
            sl.k = kk;
            slots[buf][t >> 6] = sl;
        }
        __syncthreads();
        unsigned long long gk = slots[buf][0].key;
        int gw = 0;
        for (int w = 1; w < NW; w++) {
            unsigned long long kw = slots[buf][w].key;
            if (kw > gk) {
                gk = kw;
                gw = w;
            }
        }
        ox = slots[buf][gw].x;
        oy = slots[buf][gw].y;
        oz = slots[buf][gw].z;
        if (t == 0) O[j] = slots[buf][gw].k;
    }
}

__global__ void gather_kernel(int n, int m, long total, const float *__restrict__ inp,
                              const int *__restrict__ idx, float *__restrict__ out) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    long bi = g / m;
    int a = idx[g];
    const float *p = inp + (bi * n + a) * 3;
    out[g * 3 + 0] = p[0];
    out[g * 3 + 1] = p[1];
    out[g * 3 + 2] = p[2];
}

__global__ void scatteradd_kernel(int n, int m, long total, const float *__restrict__ out_g,
                                  const int *__restrict__ idx, float *__restrict__ inp_g) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    long bi = g / m;
    int a = idx[g];
    float *p = inp_g + (bi * n + a) * 3;
    atomicAdd(p + 0, out_g[g * 3 + 0]);
    atomicAdd(p + 1, out_g[g * 3 + 1]);
    atomicAdd(p + 2, out_g[g * 3 + 2]);
}

// ---- prob_sample (ProbSample op; reference: cumsumKernel + binarysearchKernel, tf_sampling_g.cu:7-104)
// What is pinned is the VALUE of every cumulative sum: fp32 addition is not associative, so the
// association order of the reference's blocked scan is part of the result (and decides which index
// a uniform draw lands on).  That order, per chunk of 8192 elements (2048 groups of 4):
//   in a group      (a+b), (a+b)+c, (a+b)+(c+d);
//   over the groups balanced pairwise sums of aligned power-of-two blocks B(.), and the prefix up
//                   to group count c is folded from the HIGHEST set bit of c down:
//                   P(c) = B(lowest block of c) + P(c - lowbit(c));
//   an element      (in-group prefix + P(groups before it)) + carry;
//   between chunks  a compensated (two-float) carry.
// How it is computed here (one 512-thread workgroup per row, nothing like the reference's LDS tree):
//   * a lane owns one group (4 consecutive elements, one 16-byte load), a wave 64 consecutive
//     groups: the balanced block sums up to 64 groups are a DPP up-sweep inside the wave (row
//     shifts for 1,2,4,8; two readlanes for 16, 32), no LDS, no barrier;
//   * the 32 wave-block totals of a chunk cross waves through a double-buffered LDS line -- ONE
//     barrier per chunk -- and every wave scans them itself (same in-wave routine);
//   * the fold "from the highest bit down" is the in-wave down-sweep started from the prefix of the
//     blocks before the wave (injected as the value of "lane -1"), so each lane ends with exactly
//     the reference's P(c);
//   * the chunk carry is wave-uniform and kept by every wave in scalar registers;
//   * the row of cumulative sums stays in LDS (up to 32768 entries of the 160 KiB) for the
//     inverse-CDF search of the row's draws in the same launch; `temp` still receives it.
// Bit-exact with oracle/rfops_oracle.c::orc_cumsum / orc_prob_sample.
constexpr int PS_T = 512;                 // threads: 8 waves
constexpr int PS_SLOTS = 4;               // wave-blocks (64 groups) per wave and chunk
constexpr int PS_CHUNK = PS_T * PS_SLOTS * 4;  // 8192 elements
constexpr int PS_BLOCKS = PS_T / 64 * PS_SLOTS;  // 32 wave-blocks per chunk
constexpr int PS_ROW_CAP = 32768;         // cumulative sums kept in LDS for the search

template <int N>
__device__ __forceinline__ float row_shr(float v) {  // value of lane - N inside a row of 16 (0 outside)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, false));
}

// Balanced pairwise sums: afterwards lane l holds the sum of the aligned block of lowbit(l+1) lanes
// ending at l (lane 63: all 64).
__device__ __forceinline__ float ps_upsweep(float v, int lane) {
    float o;
    o = row_shr<1>(v); if ((lane & 1) == 1) v = v + o;
    o = row_shr<2>(v); if ((lane & 3) == 3) v = v + o;
    o = row_shr<4>(v); if ((lane & 7) == 7) v = v + o;
    o = row_shr<8>(v); if ((lane & 15) == 15) v = v + o;
    const float t15 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float t47 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    if (lane == 31) v = v + t15;
    if (lane == 63) v = v + t47;
    const float t31 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    if (lane == 63) v = v + t31;
    return v;
}

// From block sums to prefixes: the lane with count r = lane + 1 an odd multiple of 2^u adds the
// finished prefix at r - 2^u; for r == 2^u that is the prefix of everything BEFORE this wave
// (`base`, present from the second wave-block on).  Lane 63 (r = 64) is a boundary of the next
// level up and is left alone.
__device__ __forceinline__ float ps_downsweep(float v, int lane, bool hasbase, float base) {
    const int r = lane + 1;
#pragma unroll
    for (int u = 5; u >= 0; u--) {
        const int bit = 1 << u;
        float below = __shfl_up(v, bit, 64);
        const bool first = r == bit;
        if (first) below = base;
        if ((r & (2 * bit - 1)) == bit && (!first || hasbase)) v = v + below;
    }
    return v;
}

__global__ __launch_bounds__(PS_T) void prob_sample_kernel(int n, int m, const float *__restrict__ weights,
                                                           const float *__restrict__ draws,
                                                           float *__restrict__ cum, int *__restrict__ picked) {
    __shared__ float row[PS_ROW_CAP];
    __shared__ float blocktot[2][PS_BLOCKS];
    __shared__ float chunk_last[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *__restrict__ in = weights + (size_t)blockIdx.x * n;
    float *__restrict__ out = cum + (size_t)blockIdx.x * n;
    const bool keep = n <= PS_ROW_CAP;

    float carry = 0.f, carry_lo = 0.f;  // the running total as a compensated pair (wave-uniform)
    int par = 0;
    for (int c0 = 0; c0 < n; c0 += PS_CHUNK, par ^= 1) {
        const int cnt = min(n - c0, PS_CHUNK);     // elements in this chunk
        const int ngroups = (cnt + 3) >> 2;
        float e[PS_SLOTS][4], tot[PS_SLOTS];
#pragma unroll
        for (int s = 0; s < PS_SLOTS; s++) {
            const int k = (((wave * PS_SLOTS + s) << 6) + lane) << 2;  // first element of this lane's group
            float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
            if (k + 3 < cnt) {
                a = in[c0 + k]; b = in[c0 + k + 1]; c = in[c0 + k + 2]; d = in[c0 + k + 3];
                const float ab = a + b, cd = c + d;
                e[s][0] = a; e[s][1] = ab; e[s][2] = c + ab; e[s][3] = cd + ab;
            } else {  // the row's ragged last group: a plain running sum from 0, repeated to the end
                float run = 0.f;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    if (k + x < cnt) run = run + in[c0 + k + x];
                    e[s][x] = run;
                }
            }
            tot[s] = ps_upsweep(e[s][3], lane);
            if (lane == 63) blocktot[par][wave * PS_SLOTS + s] = tot[s];
        }
        __syncthreads();
        // every wave: prefixes of the wave-block totals (lane q: blocks 0..q), and the carry from
        // the previous chunk's grand total
        float blk = blocktot[par][lane & (PS_BLOCKS - 1)];
        blk = ps_downsweep(ps_upsweep(blk, lane), lane, false, 0.f);
        if (c0 > 0) {
            const float prev = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(chunk_last[par ^ 1])));
            const float t = prev + carry_lo;
            const float next = carry + t;
            carry_lo = t - (next - carry);
            carry = next;
        }
#pragma unroll
        for (int s = 0; s < PS_SLOTS; s++) {
            const int q = wave * PS_SLOTS + s;
            const bool hasbase = q > 0;
            const float base = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blk), q > 0 ? q - 1 : 0));
            float incl = ps_downsweep(tot[s], lane, hasbase, base);
            if (lane == 63) incl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(blk), q));
            float before = __shfl_up(incl, 1, 64);  // prefix of the groups before this lane's group
            if (lane == 0) before = base;
            const int g = (q << 6) + lane;
            const int k = g << 2;
            if (g == ngroups - 1) chunk_last[par] = incl;  // the chunk's grand total
#pragma unroll
            for (int x = 0; x < 4; x++) {
                if (k + x < cnt) {
                    const float within = g > 0 ? e[s][x] + before : e[s][x];
                    const float val = within + carry;
                    out[c0 + k + x] = val;
                    if (keep) row[c0 + k + x] = val;
                }
            }
        }
    }
    __syncthreads();  // the row is complete (LDS copy, or global for rows beyond the LDS cap)
    // inverse CDF: walk down from the last index in power-of-two strides, largest first; a stride is
    // taken while the entry it lands on still reaches the target
    const float *__restrict__ cdf = keep ? row : out;
    int top = 1;
    while (top < n) top <<= 1;
    const float total = cdf[n - 1];
    for (int j = threadIdx.x; j < m; j += PS_T) {
        const float target = draws[(size_t)blockIdx.x * m + j] * total;
        int pos = n - 1;
        for (int stride = top; stride > 0; stride >>= 1) {
            const int cand = pos - stride;
            if (cand >= 0 && cdf[cand] >= target) pos = cand;
        }
        picked[(size_t)blockIdx.x * m + j] = pos;
    }
}

}  // namespace

namespace rfi {
int fps(int b, int n, int m, const float *inp, float *temp, int *out, float *new_xyz, hipStream_t s) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (n == 0) return RF_EINVAL;  // cannot sample from an empty cloud
    if (!inp || !out) return RF_EINVAL;
#define FPS_CASE(NT, PPT)                                                                                             \
    if (new_xyz) {                                                                                                    \
        RF_LAUNCH("fps_reg", (fps_reg_kernel<NT, PPT, true>), dim3(b), dim3(NT), 0, s, n, m, inp, out, new_xyz);     \
    } else {                                                                                                          \
        RF_LAUNCH("fps_reg", (fps_reg_kernel<NT, PPT, false>), dim3(b), dim3(NT), 0, s, n, m, inp, out, new_xyz);    \
    }                                                                                                                 \
    return RF_OK
    if (n <= 512) { FPS_CASE(512, 1); }
    if (n <= 1024) { FPS_CASE(1024, 1); }
    if (n <= 2048) { FPS_CASE(1024, 2); }
    if (n <= 4096) { FPS_CASE(1024, 4); }
    if (n <= 8192) { FPS_CASE(1024, 8); }
    if (n <= 16384) { FPS_CASE(512, 32); }
#undef FPS_CASE
    if (!temp) return RF_EINVAL;
    RF_LAUNCH("fps_mem", fps_mem_kernel, dim3(b), dim3(1024), 0, s, n, m, inp, temp, out);
    if (new_xyz) {  // (the fallback kernel keeps its form: the coordinates by the gather kernel)
        const long total = (long)b * m;
        RF_LAUNCH("gather_point", gather_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0, s, n, m, total, inp,
                  (const int *)out, new_xyz);
    }
    return RF_OK;
}
// FPS over an already sorted cloud (fps_sorted_kernel).  Pays from a few hundred samples on: the sort is 18-26 us, the kernel's own
// prologue ~10, an iteration 0.1-0.3 us shorter than fps_reg's.
// (same device, batch 32, uniform clouds, tools/ab_fps_sizes.py -- fps_reg against sort + fps_sorted, ms: 16384 points 256 samples
// 0.284 / 0.275, 1024 1.118 / 0.824; 8192: 256 0.212 / 0.210, 1024 0.835 / 0.676; 4096: 256 0.164 / 0.170, 512 0.322 / 0.302, 1024
// 0.635 / 0.563; 2048: 1024 0.538 / 0.519)
bool fps_sorted_pays(int n, int m) {
    if (n > FPS_MAX_REG_POINTS || m > n) return false;
    return (n > 4096 && m >= 256) || (n > 2048 && m >= 512);
}
int fps_sorted(int b, int n, int m, const float *inp, const rfp::Sorted &sv, int *out, float *new_xyz, hipStream_t s) {
    if (b <= 0 || m <= 0) return RF_OK;
    if (n <= FPS_SORTED_MIN_POINTS || n > FPS_MAX_REG_POINTS || !inp || !out) return RF_EINVAL;
    int ppt = 2;
    while (1024 * ppt < n) ppt *= 2;
    if (m > 1024 * ppt) return RF_EINVAL;  // (the samples collect in LDS, in the table of the kernel's prologue)
#define FPSS_CASE(PPT)                                                                                                         \
    if (ppt == PPT) {                                                                                                          \
        if (new_xyz) {                                                                                                         \
            RF_LAUNCH("fps_sorted", (fps_sorted_kernel<1024, PPT, true>), dim3(b), dim3(1024), 0, s, n, m, sv.npad, inp, sv.orig, \
                      out, new_xyz);                                                                                           \
        } else {                                                                                                               \
            RF_LAUNCH("fps_sorted", (fps_sorted_kernel<1024, PPT, false>), dim3(b), dim3(1024), 0, s, n, m, sv.npad, inp, sv.orig, \
                      out, new_xyz);                                                                                           \
        }                                                                                                                      \
        return RF_OK;                                                                                                          \
    }
    FPSS_CASE(2) FPSS_CASE(4) FPSS_CASE(8) FPSS_CASE(16)
#undef FPSS_CASE
    return RF_EINVAL;
}
}  // namespace rfi

extern "C" {

// The op with caller scratch of a stated size: where rfi::fps_sorted_pays (clouds of more than 4096 points from 256 samples, of more
// than 2048 from 512) it sorts the cloud into the workspace and runs fps_sorted_kernel (the same indices), otherwise rf_farthestpointsampling's kernels.
size_t rf_farthestpointsampling_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0) return 0;
    if (rfi::fps_sorted_pays(n, m)) return rfp::sort_workspace_bytes(b, n);
    return sizeof(float) * (n > FPS_MAX_REG_POINTS ? (size_t)b * n : 0);
}

int rf_farthestpointsampling_ws(int b, int n, int m, const float *inp, void *workspace, size_t workspace_bytes, int *out,
                                rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (workspace_bytes < rf_farthestpointsampling_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (n > 0 && rfi::fps_sorted_pays(n, m)) {
        if (!inp || !out || !workspace) return RF_EINVAL;
        rfp::Sorted sv;
        if (int e = rfp::sort_clouds(b, n, inp, workspace, workspace_bytes, s, &sv)) return e;
        return rfi::fps_sorted(b, n, m, inp, sv, out, nullptr, s);
    }
    return rfi::fps(b, n, m, inp, (float *)workspace, out, nullptr, s);
}

size_t rf_farthestpointsampling_temp_floats(int b, int n) {
    return n > FPS_MAX_REG_POINTS ? (size_t)b * n : 0;
}

int rf_farthestpointsampling(int b, int n, int m, const float *inp, float *temp, int *out,
                             rf_stream_t stream) {
    return rfi::fps(b, n, m, inp, temp, out, nullptr, (hipStream_t)stream);
}

// FPS over the sorted cloud (fps_sorted_kernel): workspace = rfp::sorted_bytes(b, n), for clouds of 8193..16384 points.
size_t rf_farthestpointsampling_sorted_workspace_bytes(int b, int n) {
    return (b > 0 && n > FPS_SORTED_MIN_POINTS && n <= FPS_MAX_REG_POINTS) ? rfp::sort_workspace_bytes(b, n) : 0;
}

int rf_farthestpointsampling_sorted(int b, int n, int m, int form, const float *inp, void *workspace, size_t workspace_bytes,
                                    int *out, float *new_xyz, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (n <= FPS_SORTED_MIN_POINTS || n > FPS_MAX_REG_POINTS || !inp || !out || !workspace) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    rfp::Sorted sv;
    if (int e = rfp::sort_clouds(b, n, inp, workspace, workspace_bytes, s, &sv)) return e;
    (void)form;
    return rfi::fps_sorted(b, n, m, inp, sv, out, new_xyz, s);
}

int rf_gatherpoint(int b, int n, int m, const float *inp, const int *idx, float *out,
                   rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    long total = (long)b * m;
    if (total == 0) return RF_OK;
    if (!inp || !idx || !out) return RF_EINVAL;
    RF_LAUNCH("gather_point", gather_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0,
              (hipStream_t)stream, n, m, total, inp, idx, out);
    return RF_OK;
}

int rf_scatteraddpoint(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                       rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) {
        if (!inp_g) return RF_EINVAL;
        RF_ZERO(inp_g, sizeof(float) * 3 * (size_t)b * n, s);
    }
    long total = (long)b * m;
    if (total == 0 || n == 0) return RF_OK;
    if (!out_g || !idx) return RF_EINVAL;
    RF_LAUNCH("scatteradd_point", scatteradd_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0, s, n,
              m, total, out_g, idx, inp_g);
    return RF_OK;
}

int rf_probsample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                  rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || m == 0) return RF_OK;
    if (n == 0) return RF_EINVAL;
    if (!inp_p || !inp_r || !temp || !out) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    RF_LAUNCH("prob_sample", prob_sample_kernel, dim3(b), dim3(PS_T), 0, s, n, m, inp_p, inp_r, temp, out);
    return RF_OK;
}

}  // extern "C"
