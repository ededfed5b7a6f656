// common.hpp -- shared helpers for the gfx950 kernels of librfops.so.
// Compiled with -ffp-contract=off: every FMA in this library is an explicit fmaf(), so the
// fp32 instruction sequence is the one the reference's CUDA ops execute
// and matches oracle/rfops_oracle.c bit for bit where that is required.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/rfops.h"

namespace rf {

constexpr int kWave = 64;  // gfx950 wavefront

// squared distance, the reference CUDA arithmetic: fma(dz,dz, fma(dx,dx, dy*dy))
__device__ __forceinline__ float d2_fma(float dx, float dy, float dz) {
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- profiling hook (rf_profile_enable / rf_profile_collect) --------------------------
struct ProfScope {
    ProfScope(const char *name, hipStream_t s);
    ~ProfScope();
    const char *name;
    hipStream_t stream;
    void *start, *stop;  // hipEvent_t, owned by the scope until it ends
};

// RF_OK iff the current HIP device is a gfx950; RF_ENODEVICE otherwise (runtime.hip)
int require_device();

// Zero `bytes` bytes (a multiple of 4, 4-byte aligned) at p on stream s with a KERNEL, never
// hipMemsetAsync: on ROCm 7 a memset node of a small buffer captured into a HIP graph writes garbage
// from the second replay on (tools/experiments/graph_memset_probe.py), and callers may be capturing.
// Returns an RF_* / hipError status (runtime.hip).
int zero_async(void *p, size_t bytes, hipStream_t s);

// Workspaces and sorted-set handles are read with 16-byte vector loads at offsets that are multiples of 256: the pointer the
// caller hands over must be 16-byte aligned (include/rfops.h; anything hipMalloc returns is).  Checked at the boundary -- a
// misaligned sub-allocation would otherwise be a memory fault inside a kernel.
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Workgroup ids go round the 8 XCDs (id % 8), each with an L2 of its own.  Logical position of workgroup `id` of `total` such that
// every XCD owns a CONTIGUOUS eighth of the logical order: workgroups that read the same sample then share one L2 instead of
// fetching it eight times.  A speed-up only: any mapping is correct.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned total) {
    const unsigned q8 = total >> 3, r8 = total & 7u, xcd = id & 7u;
    return (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + (id >> 3);
}

}  // namespace rf

#define RF_HIP(expr)                            \
    do {                                        \
        hipError_t _e = (expr);                 \
        if (_e != hipSuccess) return (int)_e;   \
    } while (0)

// Launch + per-kernel event bracket (no-op unless profiling is enabled) + launch check.
#define RF_LAUNCH(name, kernel, grid, block, shmem, stream, ...)             \
    do {                                                                     \
        if (int _dv = rf::require_device()) return _dv;                      \
        rf::ProfScope _p(name, stream);                                      \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
    } while (0);                                                             \
    RF_HIP(hipGetLastError())

#define RF_ZERO(ptr, bytes, stream)                                    \
    do {                                                               \
        if (int _zs = rf::zero_async((ptr), (bytes), (stream))) return _zs; \
    } while (0)
