// gather_scan.hip -- what a PER-LANE GATHERED block scan costs on gfx950 (round 4, sub-wave query tiles).
// The culled Chamfer sweep streams a candidate block through SGPRs to all 64 lanes; a quad-per-query tile would
// instead let every quad scan its OWN 16-record block: each lane gathers a quarter of it (3 x dwordx4 = 48 B) from the
// L2-resident sorted cloud and evaluates 4 pairs.  This measures rounds/s of that loop (and of the 1-query-per-lane form,
// 12 x dwordx4 = 192 B per lane, 16 pairs) at 1..8 waves per SIMD, block ids with the locality of a real traversal
// (a handful of blocks around a home position) or uniformly random inside the cloud.
// Development aid: hipcc --offload-arch=gfx950 -O3 gather_scan.hip -o gather_scan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float d2f(float dx, float dy, float dz) { return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy)); }

// QUAD: 4 lanes per query, 4 records per lane.  !QUAD: 1 lane per query, 16 records per lane.
template <bool QUAD>
__global__ __launch_bounds__(256) void k(const float *__restrict__ tab, int nblk_per_cloud, int nclouds, int rounds, int spread, float *out) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int cloud = gw % nclouds;
    const int qid = QUAD ? (gw * 16 + (lane >> 2)) : (gw * 64 + lane);
    unsigned h = qid * 2654435761u + 12345u;
    const int home = (int)(h % (unsigned)nblk_per_cloud);
    const float qx = (float)(h & 1023) * 1e-3f, qy = (float)((h >> 10) & 1023) * 1e-3f, qz = (float)((h >> 20) & 1023) * 1e-3f;
    const float4 *t4 = (const float4 *)(tab + (size_t)cloud * nblk_per_cloud * 48);
    float best = 1e30f;
    int bblk = 0;
    for (int r = 0; r < rounds; r++) {
        h = h * 1664525u + 1013904223u;
        int blk = spread > 0 ? home + (int)((h >> 8) % (unsigned)spread) - spread / 2 : (int)((h >> 8) % (unsigned)nblk_per_cloud);
        blk = blk < 0 ? 0 : (blk >= nblk_per_cloud ? nblk_per_cloud - 1 : blk);
        float cm = 1e30f;
        if (QUAD) {
            const float4 *p = t4 + (size_t)blk * 12 + (lane & 3) * 3;
            const float4 a = p[0], b = p[1], c = p[2];
            const float d0 = d2f(a.x - qx, a.y - qy, a.z - qz), d1 = d2f(a.w - qx, b.x - qy, b.y - qz);
            const float d3 = d2f(b.z - qx, b.w - qy, c.x - qz), d4 = d2f(c.y - qx, c.z - qy, c.w - qz);
            cm = fminf(fminf(d0, d1), fminf(d3, d4));
            cm = fminf(cm, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cm), 0xB1, 0xf, 0xf, false)));  // quad_perm [1,0,3,2]
            cm = fminf(cm, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cm), 0x4E, 0xf, 0xf, false)));  // quad_perm [2,3,0,1]
        } else {
            const float4 *p = t4 + (size_t)blk * 12;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float4 a = p[u * 3], b = p[u * 3 + 1], c = p[u * 3 + 2];
                const float d0 = d2f(a.x - qx, a.y - qy, a.z - qz), d1 = d2f(a.w - qx, b.x - qy, b.y - qz);
                const float d3 = d2f(b.z - qx, b.w - qy, c.x - qz), d4 = d2f(c.y - qx, c.z - qy, c.w - qz);
                cm = fminf(cm, fminf(fminf(d0, d1), fminf(d3, d4)));
            }
        }
        if (cm < best) {
            best = cm;
            bblk = blk;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best + (float)bblk;
}

template <bool QUAD>
void run(const char *name, const float *tab, int nblk, int nclouds, int waves_per_simd, int spread, float *out) {
    const int rounds = 64;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<QUAD><<<blocks, 256>>>(tab, nblk, nclouds, rounds, spread, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k<QUAD><<<blocks, 256>>>(tab, nblk, nclouds, rounds, spread, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double wave_rounds = (double)blocks * 4 * rounds;
    const double pairs = wave_rounds * 64 * (QUAD ? 4 : 16);
    printf("%-22s spread %5d waves/SIMD=%d  %.4f ms  %.1f ns per round per wave-slot  %.2e pairs/s  %.0f GB/s gathered\n", name, spread, waves_per_simd, ms,
           ms * 1e6 / rounds, pairs / (ms * 1e-3), pairs * 12 / (ms * 1e-3) / 1e9);
}

int main() {
    const int nclouds = 32, nblk = 1024;
    const size_t nfl = (size_t)nclouds * nblk * 48;
    std::vector<float> h(nfl);
    for (size_t i = 0; i < nfl; i++) h[i] = (float)((i * 2654435761u) & 0xFFFF) * 1.5e-5f;
    float *tab, *out;
    (void)hipMalloc(&tab, nfl * sizeof(float));
    (void)hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    (void)hipMemcpy(tab, h.data(), nfl * sizeof(float), hipMemcpyHostToDevice);
    for (int spread : {8, 64, 0})
        for (int w : {1, 2, 4, 8}) {
            run<true>("quad (48 B / lane)", tab, nblk, nclouds, w, spread, out);
            run<false>("lane (192 B / lane)", tab, nblk, nclouds, w, spread, out);
        }
    return 0;
}
