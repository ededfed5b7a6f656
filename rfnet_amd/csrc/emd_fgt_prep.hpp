// emd_fgt_prep.hpp -- the geometry pass of emd_fgt.hip (the clouds' common centre, their radii about it, the per-element validity
// flag of the expansion) as a workgroup-level device function: approxmatch.hip runs it inside its own prologue launch (am_init)
// instead of a launch of its own.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace rfe {

// A PRE-FILTER, not the accuracy bound: g * R_rows * R_cols beyond which the expansion is not even attempted (every row of such
// an element would fail its certificate).  Any two clouds inside a unit cube stay below it at level -1.  What the expansion
// delivers is certified ROW BY ROW in emd_fgt.hip (kRowEps): the degree-P series of exp(t) misses by |t|^(P+1)/(P+1)! max(1, e^t),
// i.e. 8.6e-6 of the term at t = -1.5 with P = 10 -- little for clouds that fill their box (few pairs sit there: 8e-8 of a row
// sum on C4 with the weights the schedule really meets, whose leftover mass lives in the corners; 4e-10 with flat weights), NOT
// for two clusters in opposite corners, where every pair of a row sits there (4e-6).
constexpr double kBound = 1.5;
// certified relative error of an expanded row sum.  The dense sweeps these levels replace carry ~1e-7 per term (v_exp_f32) and a
// few 1e-7 of fp32 summation; level -1 moves 0.14 % of C4's mass (14 % for corner clusters), so 4e-6 of its row sums is 6e-9
// (6e-7) of the cost, and 4e-6 of an entry's share from these levels against the entries' rel 1e-4 bar.  Over C4's 32 samples
// the bound peaks at 2.5e-6 (true error 3.4e-7): no row fails.  At 1e-6, 43 of 196 608 rows would -- and one failed row costs
// its launch ~4 us (the workgroup's direct sums sit on the launch's critical path): am_fgt 148 us per call against 121 without
// the certificate; at 1e-7, 3 % of the rows fail (212 us).  tools/experiments/fgt_row_bound_chain.py.
constexpr double kRowEps = 4.0e-6;

struct Geom {  // per batch element
    double ox, oy, oz, rarb;
    double r1, r2;  // the largest distances of xyz1's / xyz2's points from the centre
    int bad, pad;   // bad != 0: this element's extent is beyond the pre-filter, or a coordinate is not finite -- direct sums
};

__device__ __forceinline__ float prep_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float prep_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One workgroup of NT threads (all of them must call) for batch element bi: centre = middle of the bounding box of both clouds;
// R1, R2 = the clouds' largest distances from it; the element's record is written whole -- nothing to clear beforehand.
template <int NT>
__device__ __forceinline__ void fgt_prep_block(int bi, int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                               double a_max, Geom *__restrict__ geom) {
    constexpr int NWV = NT / 64;
    __shared__ float red[NWV][6];
    __shared__ float ctr[3];
    __shared__ int nonfin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *A = xyz1 + (size_t)bi * n * 3, *B = xyz2 + (size_t)bi * m * 3;
    if (tid == 0) nonfin = 0;
    __syncthreads();
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool nonfinite = false;
    for (int i = tid; i < n + m; i += NT) {
        const float *p = i < n ? A + (size_t)i * 3 : B + (size_t)(i - n) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = p[c];
            nonfinite = nonfinite || !(fabsf(v) < INFINITY);
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
    }
    const bool anybad = __ballot(nonfinite) != 0ull;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        lo[c] = prep_wave_min(lo[c]);
        hi[c] = prep_wave_max(hi[c]);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            red[wave][c] = lo[c];
            red[wave][3 + c] = hi[c];
        }
    }
    if (anybad && lane == 0) atomicOr(&nonfin, 1);
    __syncthreads();
    if (tid < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < NWV; w++) {
            l = fminf(l, red[w][tid]);
            h = fmaxf(h, red[w][3 + tid]);
        }
        ctr[tid] = 0.5f * l + 0.5f * h;
    }
    __syncthreads();
    const float ox = ctr[0], oy = ctr[1], oz = ctr[2];
    float r1 = 0.f, r2 = 0.f;
    for (int i = tid; i < n + m; i += NT) {
        const float *p = i < n ? A + (size_t)i * 3 : B + (size_t)(i - n) * 3;
        const float dx = p[0] - ox, dy = p[1] - oy, dz = p[2] - oz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (i < n) r1 = fmaxf(r1, d2); else r2 = fmaxf(r2, d2);
    }
    r1 = prep_wave_max(r1);
    r2 = prep_wave_max(r2);
    __syncthreads();
    if (lane == 0) {
        red[wave][0] = r1;
        red[wave][1] = r2;
    }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < NWV; w++) {
            a = fmaxf(a, red[w][0]);
            c = fmaxf(c, red[w][1]);
        }
        const double rarb = sqrt((double)a) * sqrt((double)c);
        geom[bi] = Geom{(double)ox, (double)oy, (double)oz, rarb, sqrt((double)a), sqrt((double)c),
                        (nonfin != 0 || !(2.0 * a_max * rarb <= kBound)) ? 1 : 0, 0};
    }
}

}  // namespace rfe
