// handoff.hip -- what one all-to-all exchange between the waves of an FPS "cluster" costs on gfx950.
//
// The question: farthest_point_sample is a chain of m-1 dependent arg-max
// reductions; splitting one cloud over W waves on several CUs shortens the scan but every iteration
// then needs each wave's (d2, index) winner at every other wave.  This program times exactly that
// exchange, with nothing else in the loop: W one-wave workgroups (one per CU, forced by an LDS
// request), each iteration a wave stores ONE 8-byte {tag, value} granule into its slot and lane l
// re-reads slot l until all W tags carry the iteration number (MI355X_MICROARCH.md, "Valid forms",
// R2 granules).  Reported: ns per iteration as seen by rank 0, and the XCC id of every member.
//
//   stride 8 -> members blockIdx 0, 8, 16, ...: one XCD under the observed round-robin placement
//   stride 1 -> members on W different XCDs (W <= 8) / spread over all of them
// Store / load flavours: sc1 (= relaxed agent-scope atomic store / load), sc0 sc1, plain store.
// "plain store + sc1 load" is only meaningful inside one XCD (the L2 is the meeting point there) and
// is measured to price an XCD-local fast path; it is NOT a valid cross-XCD form.
//
// Development aid: hipcc --offload-arch=gfx950 -O3 handoff.hip -o handoff && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

enum { ST_SC1 = 0, ST_SC01 = 1, ST_PLAIN = 2, ST_ATOMIC = 3 };
enum { LD_SC1 = 0, LD_SC01 = 1, LD_ATOMIC = 2 };

template <int ST>
__device__ __forceinline__ void st64(u64 *p, u64 v) {
    if (ST == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_SC01) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == ST_ATOMIC) __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int LD>
__device__ __forceinline__ u64 ld64(u64 *p) {
    u64 v;
    if (LD == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == LD_SC01) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == LD_ATOMIC) v = __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

// slots: [2][64] granules (double buffered by iteration parity), zeroed by the host before every launch.
// work: dependent VALU filler per iteration (0 = bare exchange)
template <int ST, int LD>
__global__ __launch_bounds__(64) void xchg(u64 *slots, int W, int stride, int iters, int work, u64 *out, unsigned *xcc,
                                           unsigned *fail) {
    extern __shared__ char lds_pad[];  // the request keeps it at one workgroup per CU
    const int b = blockIdx.x;
    if (b % stride != 0 || b / stride >= W) return;
    const int r = b / stride;
    const int lane = threadIdx.x;
    if (lane == 0) xcc[r] = xcc_id();
    float acc = lane * 1e-3f;
    u64 t0 = 0;
    for (int i = 1; i <= iters; i++) {
        if (i == 17) t0 = wall_clock64();  // the first iterations include the grid's start-up skew
        for (int w = 0; w < work; w++) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(acc));
        u64 *s = slots + (i & 1) * 64;
        if (lane == 0) st64<ST>(s + r, ((u64)(unsigned)i << 32) | (unsigned)r);
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            if (lane < W) ok = (ld64<LD>(s + lane) >> 32) == (unsigned)i;
            if (__all(ok)) break;
            if (++spins > (1u << 22)) {  // bounded: a hang here would take the box down
                if (lane == 0) atomicAdd(fail, 1u);
                return;
            }
        }
    }
    u64 t1 = wall_clock64();
    if (lane == 0 && r == 0) out[0] = t1 - t0;
    if (acc == 12345.f) out[1] = 1;
}

template <int ST, int LD>
static void run(const char *name, int W, int stride, int iters, int work) {
    u64 *slots, *out;
    unsigned *xcc, *fail;
    CK(hipMalloc(&slots, 2 * 64 * 8));
    CK(hipMalloc(&out, 16));
    CK(hipMalloc(&xcc, 64 * 4));
    CK(hipMalloc(&fail, 4));
    double best = 1e30;
    std::vector<unsigned> hx(64);
    unsigned hfail = 0;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(slots, 0, 2 * 64 * 8));
        CK(hipMemset(out, 0, 16));
        CK(hipMemset(fail, 0, 4));
        const int grid = (W - 1) * stride + 1;
        hipLaunchKernelGGL((xchg<ST, LD>), dim3(grid), dim3(64), 96 * 1024, 0, slots, W, stride, iters, work, out, xcc, fail);
        CK(hipDeviceSynchronize());
        u64 h[2];
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), xcc, 64 * 4, hipMemcpyDeviceToHost));
        unsigned f;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        hfail += f;
        double ns = (double)h[0] * 10.0 / (iters - 16);  // wall_clock64: 100 MHz
        if (ns < best) best = ns;
    }
    printf("%-28s W=%2d stride=%d work=%4d : %7.1f ns/iteration  fails=%u  xcc:", name, W, stride, work, best, hfail);
    for (int i = 0; i < W && i < 16; i++) printf(" %u", hx[i]);
    printf("\n");
    fflush(stdout);
    CK(hipFree(slots));
    CK(hipFree(out));
    CK(hipFree(xcc));
    CK(hipFree(fail));
}

int main(int argc, char **argv) {
    const int iters = 20000;
    CK(hipFuncSetAttribute((const void *)xchg<ST_SC1, LD_SC1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)xchg<ST_SC01, LD_SC01>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)xchg<ST_PLAIN, LD_SC1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)xchg<ST_PLAIN, LD_SC01>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)xchg<ST_ATOMIC, LD_SC1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void *)xchg<ST_ATOMIC, LD_ATOMIC>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    const int Ws[] = {2, 4, 8, 16, 32};
    for (int stride : {8, 1}) {
        for (int W : Ws) {
            if (stride == 8 && W > 32) continue;
            run<ST_SC1, LD_SC1>("st sc1 / ld sc1", W, stride, iters, 0);
            run<ST_SC01, LD_SC01>("st sc0sc1 / ld sc0sc1", W, stride, iters, 0);
            run<ST_ATOMIC, LD_SC1>("atomic xchg / ld sc1", W, stride, iters, 0);
            run<ST_ATOMIC, LD_ATOMIC>("atomic xchg / atomic ld", W, stride, iters, 0);
            if (stride == 8) {
                run<ST_PLAIN, LD_SC1>("st plain / ld sc1 (XCD)", W, stride, iters, 0);
                run<ST_PLAIN, LD_SC01>("st plain / ld sc0sc1 (XCD)", W, stride, iters, 0);
            }
        }
    }
    // with some dependent work per iteration (what an FPS member does between exchanges)
    for (int work : {100, 400}) {
        run<ST_SC1, LD_SC1>("st sc1 / ld sc1", 8, 8, iters, work);
        run<ST_SC1, LD_SC1>("st sc1 / ld sc1", 8, 1, iters, work);
        run<ST_PLAIN, LD_SC1>("st plain / ld sc1 (XCD)", 8, 8, iters, work);
    }
    return 0;
}
