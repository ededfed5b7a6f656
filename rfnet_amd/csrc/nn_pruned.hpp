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
// dirs: bit 0 = direction 1 (dist1/idx1), bit 1 = direction 2; outputs of a direction not asked
// for may be NULL.
int pruned_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                       float *dist2, int *idx2, void *workspace, size_t workspace_bytes, hipStream_t s,
                       unsigned long long *stats_out, int dirs = 3);

// The sort on its own (one cloud per batch element), for other operators that want the Hilbert
// order: records (x, y, z packed) and original indices in key order, padded to a multiple of 64
// with +inf / -1, and the boxes of every 16-record block / 64-record superblock (layouts in
// nn_pruned.hip).  The views point into `workspace`.
struct Sorted {
    const float *xyz;
    const int *orig;
    const float *box16;
    const float *box64;
    const int *pos0;  // (b) sorted position of original index 0; behind it (b) the crowded flags and (b) the non-finite flags
                      // (1: the cloud has a NaN or infinite coordinate -- its boxes exclude such points)
    int npad;
};
size_t sort_workspace_bytes(int b, int n);
int sort_clouds(int b, int n, const float *src, void *workspace, size_t workspace_bytes, hipStream_t s, Sorted *out);

// The pieces, for callers that keep a sorted set across several sweeps (a "handle" is a caller-owned
// buffer of sorted_bytes(b, n) bytes whose layout is a pure function of (b, n)).
size_t sorted_bytes(int b, int n);
Sorted sorted_view(int b, int n, const void *buf);
int sort_sets(int b, int nsets, const int *n, const float *const *src, const Sorted *out, hipStream_t s,
              unsigned long long *dbg);
int sweep_sorted(int b, int n, int m, const Sorted &s0, const Sorted &s1, float *dist1, int *idx1, float *dist2,
                 int *idx2, int dirs, hipStream_t s, unsigned long long *stats_dev);

// rf_chamfer_step on the culled path: sort, sweep (which also leaves {winner position, own gradient term}
// records in sorted order) and the backward in sorted index space.  gd1 (b, n), gd2 (b, m) upstream gradients.
size_t pruned_step_workspace_bytes(int b, int n, int m);
int pruned_step(int b, int n, int m, const float *xyz1, const float *xyz2, const float *gd1, const float *gd2,
                float *dist1, int *idx1, float *dist2, int *idx2, float *grad_xyz1, float *grad_xyz2, void *workspace,
                size_t workspace_bytes, hipStream_t s);

}  // namespace rfp
