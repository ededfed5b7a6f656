// group_internal.hpp -- launchers shared by the entry points of sampling.hip / grouping.hip and the one-call
// sample-and-group of sample_group.hip.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "nn_pruned.hpp"

namespace rfi {

// farthest_point_sample (sampling.hip).  new_xyz (b, m, 3): the samples' coordinates, written by the same launch (NULL: not
// wanted) -- gather_point(inp, out) fused: the kernel reloads every winner's coordinates anyway.  temp: b*n floats when
// n exceeds the register-resident limit (rf_farthestpointsampling_temp_floats), NULL otherwise.
int fps(int b, int n, int m, const float *inp, float *temp, int *out, float *new_xyz, hipStream_t s);

// The same op over a cloud that is already sorted (fps_sorted_kernel: clouds of 1025..16384 points): the same indices; a new
// sample only re-scans the regions it can still change.  fps_sorted_pays: from how many samples on the sort is repaid.
bool fps_sorted_pays(int n, int m);
int fps_sorted(int b, int n, int m, const float *inp, const rfp::Sorted &sv, int *out, float *new_xyz, hipStream_t s);

// query_ball_point on a sorted dataset (grouping.hip, query_ball_boxes_kernel).  grouped_xyz (b, m, nsample, 3) or NULL:
// group_point(xyz1, idx) fused; zero_empty: rows of empty balls are written as index 0 instead of being left untouched.
// The caller has checked the domain (64 <= n <= 65536, nsample <= 64, b <= 65535).
int ball_boxes(int b, int n, int m, float radius, const float *radius_dev, int nsample, const float *xyz1, const float *xyz2,
               const rfp::Sorted &so, int *idx, int *pts_cnt, float *grouped_xyz, int zero_empty, hipStream_t s);

}  // namespace rfi
