// emd_fgt.hpp -- internal interface of emd_fgt.hip (the broad levels of approx_match's schedule as truncated Taylor expansions)
// to approxmatch.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rfe {

// The broadest level the expansion is built for: exp(-a d2) with a <= kFgtMaxA (the reference schedule's levels -1, -0.25, 0).
constexpr float kFgtMaxA = 1.0f;

// bytes of scratch for b batch elements whose larger cloud has `nmax` points (geometry, validity word, per-chunk moments of
// two sets); 256-byte aligned sizes.
size_t fgt_workspace_bytes(int b, int nmax);

// Once per call, BEFORE the chain below: the clouds' common centre and radii per batch element, and the element's validity flag --
// raised when its extent makes the truncated series miss its error bound for a level of sharpness a_max, or a coordinate is not
// finite (then every fgt_* call below forms THAT element's row sums directly, as the dense sweeps do, inside the same launches).
// The pass is rfe::fgt_prep_block (emd_fgt_prep.hpp), run by the caller inside a launch of its own -- one workgroup per batch
// element writing record bi of the array this returns (an array of rfe::Geom).
void *fgt_geom(void *ws, int b, int nmax);

// The phases of a level (tf_approxmatch.cu:36-177) with their row sums S = sum_col w[col] exp(-a |row - col|^2) from the
// expansion; state vectors as in approxmatch.hip's dense sweeps (per batch element `stride` floats apart, original order).
// a = -level * (kLog2e as float) * ln 2 in double: exactly the function the dense sweeps evaluate with v_exp_f32.
// The caller drives the chain strictly as  fgt_p3p1(v0, has_p3 = false), fgt_p2(v0), fgt_p3p1(v0 + 1, true), fgt_p2(v0 + 1), ...:
// every call leaves in the workspace the moments the next one evaluates with (one launch per phase).
//   fgt_p3p1: [has_p3: remainL[k] = max(0, remainL[k] - ratioL_prev[k] * S3[k]), S3 over xyz2 with w = ratioR of the previous
//             level at a_prev]  then  ratioL_out[k] = remainL[k] / (1e-9 + S1[k]),  S1 over xyz2 with w = remainR at a_cur.
//   fgt_p2:   sumr[l] over xyz1 with w = ratioL of this level at a_cur; ratioR_out / remainR update as am_rowl_kernel's epilogue;
//             a_next: the next level's sharpness (< 0: the schedule ends here).
int fgt_p3p1(int b, int n, int m, const float *xyz1, const float *xyz2, bool has_p3, double a_prev, double a_cur,
             const float *ratioR_prev, const float *remainR, const float *ratioL_prev, float *remainL, float *ratioL_out,
             size_t stride, void *ws, hipStream_t s);
int fgt_p2(int b, int n, int m, const float *xyz1, const float *xyz2, double a_cur, double a_next, const float *ratioL,
           float *remainR, float *ratioR_out, size_t stride, void *ws, hipStream_t s);

}  // namespace rfe
