// valu_rate.hip -- measures the fp32 VALU issue rates that bound the Chamfer / EMD sweeps on
// gfx950: v_fma_f32, v_pk_fma_f32, v_sub_f32, v_pk_add_f32, v_min3_f32, v_exp_f32 and the
// Chamfer pair mix, at 1..8 waves per SIMD.  Also reads the shader clock (s_memtime vs
// s_memrealtime).  Development aid: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ void k(float *out, int iters, unsigned long long *clk) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    f2 p0 = {0, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7}, p4 = {1, 1}, p5 = {2, 2}, p6 = {3, 3}, p7 = {4, 4};
    f2 pa = {a, a}, pb = {b, b};
    const float sa2 = __builtin_amdgcn_readfirstlane(a), sb2 = __builtin_amdgcn_readfirstlane(b);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 1) {  // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                              "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
        } else if (MODE == 2) {  // v_sub_f32 (VOP2)
            REP8(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                              "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 3) {  // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n"
                              "v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));)
        } else if (MODE == 4) {  // v_min3_f32
            REP8(asm volatile("v_min3_f32 %0, %8, %9, %0\n v_min3_f32 %1, %8, %9, %1\n v_min3_f32 %2, %8, %9, %2\n v_min3_f32 %3, %8, %9, %3\n"
                              "v_min3_f32 %4, %8, %9, %4\n v_min3_f32 %5, %8, %9, %5\n v_min3_f32 %6, %8, %9, %6\n v_min3_f32 %7, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 5) {  // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 6) {  // Chamfer pair mix: 3 sub, mul, 2 fmac per pair + min3 per 2 pairs (13 instr per 2 pairs) x4
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_mul_f32 %1, %1, %1\n v_sub_f32 %2, %8, %6\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_sub_f32 %0, %9, %4\n v_sub_f32 %3, %8, %5\n v_mul_f32 %3, %3, %3\n v_sub_f32 %2, %9, %6\n v_fmac_f32 %3, %0, %0\n v_fmac_f32 %3, %2, %2\n"
                "v_min3_f32 %7, %7, %1, %3\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 7) {  // same pair mix with an SGPR candidate operand (scalar-load path)
            float sa = __builtin_amdgcn_readfirstlane(a), sb = __builtin_amdgcn_readfirstlane(b);
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_mul_f32 %1, %1, %1\n v_sub_f32 %2, %8, %6\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_sub_f32 %0, %9, %4\n v_sub_f32 %3, %8, %5\n v_mul_f32 %3, %3, %3\n v_sub_f32 %2, %9, %6\n v_fmac_f32 %3, %0, %0\n v_fmac_f32 %3, %2, %2\n"
                "v_min3_f32 %7, %7, %1, %3\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(sa), "s"(sb));)
        } else if (MODE == 8) {  // EMD column: 3 sub, mul, 2 fmac, mul, exp, fmac -- ONE accumulator (x8 columns)
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_mul_f32 %1, %1, %1\n v_sub_f32 %2, %8, %6\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_mul_f32 %1, %9, %1\n v_exp_f32 %1, %1\n v_fmac_f32 %7, %8, %1\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(sa2), "s"(sb2));)
        } else if (MODE == 9) {  // same, two columns interleaved with two accumulators (x4)
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n v_sub_f32 %2, %8, %6\n v_mul_f32 %1, %1, %1\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_mul_f32 %1, %9, %1\n v_exp_f32 %1, %1\n"
                "v_sub_f32 %0, %9, %4\n v_sub_f32 %2, %8, %5\n v_mul_f32 %2, %2, %2\n v_fmac_f32 %2, %0, %0\n v_sub_f32 %0, %9, %6\n v_fmac_f32 %2, %0, %0\n"
                "v_mul_f32 %2, %9, %2\n v_exp_f32 %2, %2\n v_fmac_f32 %7, %8, %1\n v_fmac_f32 %3, %8, %2\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(sa2), "s"(sb2));)
        } else if (MODE == 10) {  // mode 8 + 2 SALU moves per column: is scalar issue free next to VALU?
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n s_mov_b32 s20, s21\n v_mul_f32 %1, %1, %1\n v_sub_f32 %2, %8, %6\n v_fmac_f32 %1, %0, %0\n s_mov_b32 s22, s23\n v_fmac_f32 %1, %2, %2\n"
                "v_mul_f32 %1, %9, %1\n v_exp_f32 %1, %1\n v_fmac_f32 %7, %8, %1\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(sa2), "s"(sb2) : "s20", "s22");)
        } else if (MODE == 11) {  // chamfer pair mix with an SGPR operand + 1 SALU move per pair
            float sa = __builtin_amdgcn_readfirstlane(a), sb = __builtin_amdgcn_readfirstlane(b);
            REP8(asm volatile(
                "v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %9, %5\n s_mov_b32 s20, s21\n v_mul_f32 %1, %1, %1\n v_sub_f32 %2, %8, %6\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_sub_f32 %0, %9, %4\n v_sub_f32 %3, %8, %5\n s_mov_b32 s22, s23\n v_mul_f32 %3, %3, %3\n v_sub_f32 %2, %9, %6\n v_fmac_f32 %3, %0, %0\n v_fmac_f32 %3, %2, %2\n"
                "v_min3_f32 %7, %7, %1, %3\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(sa), "s"(sb) : "s20", "s22");)
        } else if (MODE == 12) {  // v_permlane32_swap (VOP1), 4 independent register pairs
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));)
        } else if (MODE == 13) {  // v_permlane16_swap
            REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));)
        } else if (MODE == 14) {  // v_min_f32_dpp row_ror:8, 8 independent chains (sources never just-written)
            REP8(asm volatile("v_min_f32_dpp %0, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %1, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %2, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %3, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %4, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %5, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_min_f32_dpp %6, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n v_min_f32_dpp %7, %8, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 15) {  // v_cndmask_b32_e64 with an SGPR-pair mask
            REP8(asm volatile("v_cndmask_b32_e64 %0, %8, %9, s[20:21]\n v_cndmask_b32_e64 %1, %8, %9, s[20:21]\n v_cndmask_b32_e64 %2, %8, %9, s[20:21]\n v_cndmask_b32_e64 %3, %8, %9, s[20:21]\n"
                              "v_cndmask_b32_e64 %4, %8, %9, s[20:21]\n v_cndmask_b32_e64 %5, %8, %9, s[20:21]\n v_cndmask_b32_e64 %6, %8, %9, s[20:21]\n v_cndmask_b32_e64 %7, %8, %9, s[20:21]\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "s20", "s21");)
        } else if (MODE == 16) {  // ds_bpermute_b32, 8 in flight
            REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                              "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 17) {  // v_cndmask_b32_e32 (VCC mask)
            REP8(asm volatile("v_cndmask_b32_e32 %0, %8, %9, vcc\n v_cndmask_b32_e32 %1, %8, %9, vcc\n v_cndmask_b32_e32 %2, %8, %9, vcc\n v_cndmask_b32_e32 %3, %8, %9, vcc\n"
                              "v_cndmask_b32_e32 %4, %8, %9, vcc\n v_cndmask_b32_e32 %5, %8, %9, vcc\n v_cndmask_b32_e32 %6, %8, %9, vcc\n v_cndmask_b32_e32 %7, %8, %9, vcc\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");)
        } else if (MODE == 18) {  // v_readlane_b32 + v_writelane_b32 pairs
            REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_writelane_b32 %1, s22, 5\n v_readlane_b32 s21, %2, 3\n v_writelane_b32 %3, s22, 5\n"
                              "v_readlane_b32 s20, %4, 3\n v_writelane_b32 %5, s22, 5\n v_readlane_b32 s21, %6, 3\n v_writelane_b32 %7, s22, 5\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : : "s20", "s21");)
        } else if (MODE == 19) {  // Chamfer pair mix, candidate coordinates as DPP row_newbcast source operands (sub-wave tile form)
            REP8(asm volatile(
                "v_sub_f32_dpp %0, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %1, %5, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32 %1, %1, %1\n"
                "v_sub_f32_dpp %2, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %1, %2, %2\n"
                "v_sub_f32_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %5, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mul_f32 %3, %3, %3\n"
                "v_sub_f32_dpp %2, %6, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %3, %0, %0\n v_fmac_f32 %3, %2, %2\n"
                "v_min3_f32 %7, %7, %1, %3\n"
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 20) {  // v_sub_f32_dpp row_newbcast alone, 8 independent chains
            REP8(asm volatile("v_sub_f32_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                              "v_sub_f32_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                              "v_sub_f32_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                              "v_sub_f32_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        } else if (MODE == 21) {  // v_mov_b32_dpp quad_perm broadcast (what a quad-shared query costs per exchanged value)
            REP8(asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char *name, int instr_per_iter, int waves_per_simd) {
    const int iters = 2000;
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    float *out; unsigned long long *clk;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    double winstr = (double)blocks * 4 * iters * instr_per_iter;  // wave-instructions
    double per_simd_cyc = ms * 1e-3 * ghz * 1e9 * 1024 / winstr;
    printf("%-14s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD  (%.1f T wave-lane-ops/s)\n",
           name, waves_per_simd, ms, ghz, per_simd_cyc, winstr * 64 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_pk_fma_f32", 64, w);
        run<2>("v_sub_f32", 64, w);
        run<3>("v_pk_add_f32", 64, w);
        run<4>("v_min3_f32", 64, w);
        run<5>("v_exp_f32", 64, w);
        run<6>("chamfer mix", 104, w);
        run<7>("chamfer mix sgpr", 104, w);
        run<8>("emd col 1acc", 72, w);
        run<9>("emd col 2acc", 144, w);
        run<10>("emd col +2salu", 72, w);   // cycles per VALU instruction (the 16 SALU not counted)
        run<11>("chamfer +salu", 104, w);
        run<12>("permlane32_swap", 64, w);
        run<13>("permlane16_swap", 64, w);
        run<14>("v_min_f32_dpp", 64, w);
        run<15>("cndmask_e64 sgpr", 64, w);
        run<16>("ds_bpermute", 64, w);
        run<17>("cndmask_e32 vcc", 64, w);
        run<18>("readlane/writelane", 64, w);
        run<19>("chamfer mix dpp-bcast", 104, w);
        run<20>("v_sub_f32_dpp newbcast", 64, w);
        run<21>("v_mov_dpp quad_perm", 64, w);
    }
    return 0;
}
