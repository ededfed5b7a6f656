// scatter_rows.hpp -- internal interface of scatter_rows.hip (the scatter-add gradients of group_point / three_interpolate as a
// counting sort of the slots by destination row + a gather that writes every destination row once).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace rfs {

// dst (b, n, c)[idx (b, S)] += weight (b, S) * src (b, S / K, c)[slot / K]; K = 1 with weight == NULL, or K = 3 with weights.
bool rows_csr_supported(int b, int n, int c, long S, int K);
size_t rows_csr_workspace_bytes(int b, int n, long S);
// dst is fully overwritten (rows no slot names: zeros).  workspace: rows_csr_workspace_bytes, 16-byte aligned.
int rows_csr_scatter(int b, int n, int c, long S, int K, const float *src, const int *idx, const float *weight, float *dst,
                     void *workspace, const char *build_name, const char *gather_name, hipStream_t s);

}  // namespace rfs
