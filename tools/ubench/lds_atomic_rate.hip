// lds_atomic_rate.hip -- what a CU's LDS does per cycle with float adds from a wave: ds_add_f32 (no return) to conflict-free
// addresses, to one address per 16-lane row, to ONE address; ds_add_rtn_u32; the same traffic as plain ds_read + ds_write; the
// add as two exchanges; ds_add_u64, ds_add_rtn_f32, ds_add_f64.
// Decides how the gradient of three_interpolate accumulates (LDS tile per channel slice, or not at all).
// hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *clk) {
    __shared__ float tile[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int a = MODE == 1 ? (lane >> 4) + wave * 64 : (MODE == 2 ? wave * 64 : threadIdx.x);  // word index
    if (MODE == 5) a = ((lane * 33) & 63) + wave * 64;                                     // a permutation inside the wave's words
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int w = (a + u * 1024 + i * 64) & 16383;
            if (MODE == 3) {
                acc += (float)atomicAdd((unsigned *)&tile[w], 1u);
            } else if (MODE == 6) {  // float add by two exchanges: take what is there, put the sum back, re-add what someone left meanwhile
                float mine = v;
                for (;;) {
                    const float take = __uint_as_float(atomicExch((unsigned *)&tile[w], 0u));
                    const float left = __uint_as_float(atomicExch((unsigned *)&tile[w], __float_as_uint(take + mine)));
                    if (__ballot(left != 0.f) == 0ull) break;
                    if (left == 0.f) mine = 0.f; else mine = left;  // (a lane that is done keeps adding 0 to a taken value: harmless)
                }
            } else if (MODE == 7) {
                atomicAdd((unsigned long long *)&tile[(w * 2) & 16382], 1ull);
            } else if (MODE == 8) {
                acc += atomicAdd(&tile[w], v);
            } else if (MODE == 9) {
                atomicAdd((double *)&tile[(w * 2) & 16382], (double)v);
            } else if (MODE == 4) {
                tile[w] += v;  // read + write, no atomic
            } else {
                atomicAdd(&tile[w], v);
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tile[threadIdx.x] + acc;
}

template <int MODE>
static void run(const char *name, int tpb) {
    float *out;
    unsigned long long *clk, h[1];
    (void)hipMalloc(&out, 1024 * 1024 * 4);
    (void)hipMalloc(&clk, 1024 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(tpb), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
    const double ops = (double)iters * 8 * tpb;  // lane-operations per workgroup (= per CU)
    printf("%-44s %4d threads: %8llu ticks of 10 ns -> %.2f lane-ops per ns per CU, %.2f ns per wave instruction\n", name, tpb, h[0],
           ops / (h[0] * 10.0), h[0] * 10.0 / (ops / 64));
    (void)hipFree(out);
    (void)hipFree(clk);
}

int main() {
    for (int tpb : {256, 1024}) {
        run<0>("ds_add_f32, conflict-free", tpb);
        run<5>("ds_add_f32, conflict-free, permuted lanes", tpb);
        run<1>("ds_add_f32, 4 addresses per wave", tpb);
        run<2>("ds_add_f32, 1 address per wave", tpb);
        run<3>("ds_add_rtn_u32, conflict-free", tpb);
        run<4>("ds_read + v_add + ds_write, conflict-free", tpb);
        run<6>("float add by 2 x ds_wrxchg_rtn, conflict-free", tpb);
        run<7>("ds_add_u64, conflict-free", tpb);
        run<8>("ds_add_rtn_f32, conflict-free", tpb);
        run<9>("ds_add_f64, conflict-free", tpb);
    }
    return 0;
}
