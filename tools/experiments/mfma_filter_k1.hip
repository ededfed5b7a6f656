// Experiment (not part of the product): pass 1 of the split-bf16 MFMA filter costed in DESIGN.md 5.1.
// One v_mfma_f32_32x32x16_bf16 per 32x32 pairs forms d~ = |a|^2 + |b|^2 - 2 a.b (K = 15 of 16:
// hi*hi + hi*mid + mid*hi per coordinate, the norms as three bf16 terms each), then the VALU work
// that any consumer of the tile has to do at least: running minimum per (row, lane) and per column.
// Measures that floor at C2 (B=32, 16384 own rows x 2048 candidate columns) and checks the minima
// against a double-precision host evaluation.   hipcc --offload-arch=gfx950 -O3 mfma_filter_k1.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RB = 4;  // 32-row blocks per wave

__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// A: [b][no][16] bf16 (32 B per point), Bm: [b][nc][16] bf16
__global__ __launch_bounds__(256) void k1(int no, int nc, const uint4 *__restrict__ A,
                                          const uint4 *__restrict__ Bm, float *__restrict__ rowmin,
                                          float *__restrict__ colpart) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wpb = no / (32 * RB);
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int bi = w / wpb, wb = w - bi * wpb;
    const uint4 *Ab = A + (size_t)bi * no * 2, *Bb = Bm + (size_t)bi * nc * 2;
    bf16x8 a[RB];
    float rm[RB][16];
#pragma unroll
    for (int i = 0; i < RB; i++) {
        uint4 v = Ab[(size_t)((wb * RB + i) * 32 + r) * 2 + h];
        memcpy(&a[i], &v, 16);
#pragma unroll
        for (int q = 0; q < 16; q++) rm[i][q] = INFINITY;
    }
    float *cp = colpart + ((size_t)bi * wpb + wb) * nc;
    // candidate fragments are requested 3 tiles ahead (one tile of arithmetic is ~250 cycles, an L2
    // round trip several times that)
    uint4 nv0 = Bb[(size_t)r * 2 + h], nv1 = Bb[(size_t)(min(32, nc - 32) + r) * 2 + h],
          nv2 = Bb[(size_t)(min(64, nc - 32) + r) * 2 + h];
    for (int t = 0; t < nc; t += 32) {
        bf16x8 b;
        memcpy(&b, &nv0, 16);
        nv0 = nv1;
        nv1 = nv2;
        nv2 = Bb[(size_t)(min(t + 96, nc - 32) + r) * 2 + h];
        float cm, cmv[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        // inline asm: the builtin form zero-fills 16 accumulator registers per MFMA, lands in AGPRs
        // (16 v_accvgpr_read each) and fminf() canonicalises both operands: 430 VALU per 4 MFMAs
        // instead of 96.  Here C is the inline constant 0 and D is a VGPR tuple.  All four MFMAs are
        // issued first; an XDL result may be read by the VALU 19 wait states after its issue (the
        // hazard recogniser does not look inside asm): 3 MFMAs + s_nop 15 cover the first tile, the
        // 24 instructions consuming each tile cover the next.
        f32x16 acc[RB];
        static_assert(RB == 4, "four accumulator tuples");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %8, 0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %1, %5, %8, 0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %6, %8, 0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %3, %7, %8, 0\n\t"
                     "s_nop 15"
                     : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b));
#pragma unroll
        for (int i = 0; i < RB; i++) {
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_min_f32 %0, %0, %1" : "+v"(rm[i][q]) : "v"(acc[i][q]));
#pragma unroll
            for (int q = 0; q < 16; q += 2)  // 4 independent chains, not one 32-deep dependent chain
                asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(cmv[(q >> 1) & 3]) : "v"(acc[i][q]), "v"(acc[i][q + 1]));
        }
        cm = fminf(fminf(cmv[0], cmv[1]), fminf(cmv[2], cmv[3]));
        cm = fminf(cm, __shfl_xor(cm, 32, 64));
        if (h == 0) cp[t + r] = cm;
    }
    // row minima: C/D row of register q in half h is (q&3) + 8*(q>>2) + 4*h; reduce over the 32 lanes
#pragma unroll
    for (int i = 0; i < RB; i++)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            float v = rm[i][q];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
            if (r == 0) rowmin[(size_t)bi * no + (wb * RB + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = v;
        }
}

static unsigned short bf16_rne(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// K layout (16): own: [xh yh zh | xh yh zh | xm ym zm | nh nm nl | 1 1 1 | 0], cand: [-2xh.. | -2xm.. | -2xh.. | 1 1 1 | nh nm nl | 0]
static void pack(const float *p, bool own, unsigned short *o) {
    float hi[3], mi[3];
    double n = 0;
    for (int c = 0; c < 3; c++) {
        hi[c] = bf16_f(bf16_rne(p[c]));
        mi[c] = bf16_f(bf16_rne(p[c] - hi[c]));
        n += (double)p[c] * p[c];
    }
    float nf = (float)n, nh = bf16_f(bf16_rne(nf)), nm = bf16_f(bf16_rne(nf - nh)), nl = bf16_f(bf16_rne(nf - nh - nm));
    float v[16] = {0};
    if (own) {
        for (int c = 0; c < 3; c++) { v[c] = hi[c]; v[3 + c] = hi[c]; v[6 + c] = mi[c]; }
        v[9] = nh; v[10] = nm; v[11] = nl; v[12] = v[13] = v[14] = 1.f;
    } else {
        for (int c = 0; c < 3; c++) { v[c] = -2 * hi[c]; v[3 + c] = -2 * mi[c]; v[6 + c] = -2 * hi[c]; }
        v[9] = v[10] = v[11] = 1.f; v[12] = nh; v[13] = nm; v[14] = nl;
    }
    for (int k = 0; k < 16; k++) o[k] = bf16_rne(v[k]);
}

int main() {
    const int b = 32, no = 16384, nc = 2048;
    std::vector<float> own((size_t)b * no * 3), cand((size_t)b * nc * 3);
    srand(1);
    auto rn = []() { float s = 0; for (int i = 0; i < 12; i++) s += (float)rand() / RAND_MAX; return s - 6.f; };
    for (auto &x : own) x = rn();
    for (auto &x : cand) x = rn();
    std::vector<unsigned short> A((size_t)b * no * 16), Bm((size_t)b * nc * 16);
    for (size_t i = 0; i < (size_t)b * no; i++) pack(&own[i * 3], true, &A[i * 16]);
    for (size_t i = 0; i < (size_t)b * nc; i++) pack(&cand[i * 3], false, &Bm[i * 16]);
    uint4 *dA, *dB; float *drow, *dcol;
    const int wpb = no / (32 * RB);
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, Bm.size() * 2 + 4096);
    hipMalloc(&drow, (size_t)b * no * 4); hipMalloc(&dcol, (size_t)b * wpb * nc * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bm.data(), Bm.size() * 2, hipMemcpyHostToDevice);
    const int waves = b * wpb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k1<<<waves / 4, 256>>>(no, nc, dA, dB, drow, dcol);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 20; it++) k1<<<waves / 4, 256>>>(no, nc, dA, dB, drow, dcol);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("k1: %.4f ms per launch (B=%d, %d x %d): %.3g pairs/s\n", ms / 20, b, no, nc, (double)b * no * nc / (ms / 20 * 1e-3));
    // check batch 0: row minima vs double-precision d2 minima, relative to eps = 2^-14 (|a|^2+max|b|^2)
    std::vector<float> hrow(no);
    hipMemcpy(hrow.data(), drow, no * 4, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    double bmax = 0;
    for (int k = 0; k < nc; k++) { double n = 0; for (int c = 0; c < 3; c++) n += (double)cand[k * 3 + c] * cand[k * 3 + c]; bmax = n > bmax ? n : bmax; }
    for (int j = 0; j < no; j += 7) {
        double m = 1e30, na = 0;
        for (int c = 0; c < 3; c++) na += (double)own[j * 3 + c] * own[j * 3 + c];
        for (int k = 0; k < nc; k++) {
            double d = 0;
            for (int c = 0; c < 3; c++) { double t = (double)cand[k * 3 + c] - own[j * 3 + c]; d += t * t; }
            m = d < m ? d : m;
        }
        double e = fabs(hrow[j] - m) / (ldexp(1.0, -14) * (na + bmax));
        worst = e > worst ? e : worst;
        if (e > 1) bad++;
    }
    printf("row minima vs double: worst |err| / eps = %.3f, outside eps: %d\n", worst, bad);
    return bad != 0;
}
