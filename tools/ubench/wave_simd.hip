// Which SIMD does wave w of a 1024-thread workgroup run on?  (fps_sorted_kernel deals its 16 chunks to the waves by this.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/wave_simd tools/ubench/wave_simd.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned *out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned *d, h[64 * 16];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(64), dim3(1024), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; b++) {
        printf("workgroup %d: wave -> simd (wave slot, cu):", b);
        for (int w = 0; w < 16; w++) {
            const unsigned v = h[b * 16 + w];
            printf(" %d->%u(%u,%u)", w, (v >> 4) & 3u, v & 15u, (v >> 8) & 15u);
        }
        printf("\n");
    }
    int pat[16] = {0};
    int same = 0;
    for (int b = 0; b < 64; b++) {
        bool rr = true;
        for (int w = 0; w < 16; w++) rr = rr && (((h[b * 16 + w] >> 4) & 3u) == ((((h[b * 16] >> 4) & 3u) + w) & 3u));
        same += rr;
    }
    printf("%d of 64 workgroups place wave w on SIMD (simd of wave 0 + w) %% 4\n", same);
    (void)pat;
    return 0;
}
