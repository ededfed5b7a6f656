// stream_rate.hip -- what a 512 MiB tensor costs to write and to read on gfx950, by store / load width and cache hint.
//
// The question: am_match writes `match` (512 MiB at C4) with 4-byte non-temporal stores at 4.0 TB/s and match_cost / match_cost_grad
// read it back; are wider accesses or other hints faster?  Each kernel touches every byte of the buffer once, lanes on consecutive
// addresses (4, 8 or 16 bytes per lane and instruction), 8 accesses in flight per lane.
//
// Development aid: hipcc --offload-arch=gfx950 -O3 stream_rate.hip -o stream_rate && ./stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T, bool NT>
__global__ __launch_bounds__(256) void wr(T *__restrict__ p, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    T x;
    for (int c = 0; c < (int)(sizeof(T) / 4); c++) ((float *)&x)[c] = v + c;
#pragma unroll
    for (int u = 0; u < 8; u++, i += 256) {
        if (i < n) {
            if (NT) __builtin_nontemporal_store(x, p + i);
            else p[i] = x;
        }
    }
}
template <typename T, bool NT>
__global__ __launch_bounds__(256) void rd(const T *__restrict__ p, size_t n, float *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    float s = 0.f;
    T x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const size_t j = i + (size_t)u * 256;
        if (j < n) x[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
        else x[u] = T{};
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
        for (int c = 0; c < (int)(sizeof(T) / 4); c++) s += ((float *)&x[u])[c];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
static void run(const char *name, K k, size_t bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) k();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int it = 20;
    for (int i = 0; i < it; i++) k();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s %8.1f us  %6.2f TB/s\n", name, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 512ull << 20;
    void *buf;
    float *out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 256));
    CK(hipMemset(buf, 0, bytes));
#define GRID(T) dim3((unsigned)((bytes / sizeof(T) + 2047) / 2048))
    run("write  4 B/lane", [&] { wr<float, false><<<GRID(float), 256>>>((float *)buf, bytes / 4, 1.f); }, bytes);
    run("write  4 B/lane  nt", [&] { wr<float, true><<<GRID(float), 256>>>((float *)buf, bytes / 4, 1.f); }, bytes);
    run("write  8 B/lane  nt", [&] { wr<v2f, true><<<GRID(v2f), 256>>>((v2f *)buf, bytes / 8, 1.f); }, bytes);
    run("write 16 B/lane", [&] { wr<v4f, false><<<GRID(v4f), 256>>>((v4f *)buf, bytes / 16, 1.f); }, bytes);
    run("write 16 B/lane  nt", [&] { wr<v4f, true><<<GRID(v4f), 256>>>((v4f *)buf, bytes / 16, 1.f); }, bytes);
    run("read   4 B/lane", [&] { rd<float, false><<<GRID(float), 256>>>((const float *)buf, bytes / 4, out); }, bytes);
    run("read   4 B/lane  nt", [&] { rd<float, true><<<GRID(float), 256>>>((const float *)buf, bytes / 4, out); }, bytes);
    run("read  16 B/lane", [&] { rd<v4f, false><<<GRID(v4f), 256>>>((const v4f *)buf, bytes / 16, out); }, bytes);
    run("read  16 B/lane  nt", [&] { rd<v4f, true><<<GRID(v4f), 256>>>((const v4f *)buf, bytes / 16, out); }, bytes);
    return 0;
}
