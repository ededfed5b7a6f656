// nn_pruned.hpp -- internal interface between nn_distance.hip (the C ABI entry points) and
// nn_pruned.hip (the culled exact nearest-neighbour sweep).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rfp {

constexpr int kMaxPoints = 65536;  // per cloud: superblock ids must fit 10 bits

bool pruned_supported(int b, int n, int m);
size_t pruned_workspace_bytes(int b, int n, int m);
// stats_out (host, 16 counters, may be NULL; layout in include/rfops.h); asking for them
// synchronises the stream.
int pruned_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                       float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s,
                       unsigned long long *stats_out);

// The sort on its own (one cloud per batch element), for other operators that want the Hilbert
// order: records (x, y, z packed) and original indices in key order, padded to a multiple of 64
// with +inf / -1, and the boxes of every 16-record block / 64-record superblock (layouts in
// nn_pruned.hip).  The views point into `workspace`.
struct Sorted {
    const float *xyz;
    const int *orig;
    const float *box16;
    const float *box64;
    int npad;
};
size_t sort_workspace_bytes(int b, int n);
int sort_clouds(int b, int n, const float *src, void *workspace, size_t workspace_bytes, hipStream_t s, Sorted *out);

}  // namespace rfp
