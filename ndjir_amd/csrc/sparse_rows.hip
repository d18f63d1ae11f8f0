// sparse_rows.hip -- the receiving side of the sparse grid-gradient exchange of the ray-sharded step (SURVEY.md §8 e).
// The reference has no distributed code; the 2 GiB dense voxel gradient is exchanged as packed lists of
// (cell id, float4 row) per rank (ndjir_voxel_feature_pack_rows), all-gathered with a common capacity:
//   lists: ids (world, cap) int32, rows (world, cap, 4) fp32, counts (world) int32 -- all in device memory.
#include <hip/hip_runtime.h>

#include "common.h"

namespace ndjir {

// buf[cell] += row for every listed row of every rank but `skip_rank` (this rank's own rows are already in buf)
__global__ void __launch_bounds__(256) k_rows_apply(const int* __restrict__ ids, const float4* __restrict__ rows,
                                                    const int* __restrict__ counts, int world, int cap, int skip_rank,
                                                    float* __restrict__ buf) {
  const long long total = (long long)world * cap;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int r = (int)(t / cap), i = (int)(t - (long long)r * cap);
    if (r == skip_rank || i >= counts[r]) continue;
    const float4 v = rows[t];
    float* p = buf + (long long)ids[t] * 4;
    if (v.x != 0.f) atomicAdd(p, v.x);
    if (v.y != 0.f) atomicAdd(p + 1, v.y);
    if (v.z != 0.f) atomicAdd(p + 2, v.z);
    if (v.w != 0.f) atomicAdd(p + 3, v.w);
  }
}

// buf[cell] = 0 for every listed row of every rank: re-arms the accumulate-in-place buffer for the next step
__global__ void __launch_bounds__(256) k_rows_zero(const int* __restrict__ ids, const int* __restrict__ counts, int world, int cap,
                                                   float* __restrict__ buf) {
  const long long total = (long long)world * cap;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int r = (int)(t / cap), i = (int)(t - (long long)r * cap);
    if (i >= counts[r]) continue;
    *reinterpret_cast<float4*>(buf + (long long)ids[t] * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// clears the bitmap words of the listed cells (every set bit belongs to a listed cell)
__global__ void __launch_bounds__(256) k_rows_clear_bitmap(const int* __restrict__ ids, const int* __restrict__ count, int capacity,
                                                           unsigned* __restrict__ bitmap) {
  int n = *count;
  if (n > capacity) n = capacity;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) bitmap[(unsigned)ids[t] >> 5] = 0u;
}

static int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace ndjir

using namespace ndjir;

extern "C" int ndjir_sparse_rows_apply(const int* ids, const float* rows, const int* counts, int world, int capacity,
                                       int skip_rank, float* buf, int D, hipStream_t stream) {
  if (world <= 0 || capacity <= 0) return NDJIR_OK;
  if (!ids || !rows || !counts || !buf) return NDJIR_ERR_ARG;
  if (D != 4) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_rows_apply, dim3(blocks_for((long long)world * capacity)), dim3(256), 0, stream, ids,
                     reinterpret_cast<const float4*>(rows), counts, world, capacity, skip_rank, buf);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_zero(const int* ids, const int* counts, int world, int capacity, float* buf, int D,
                                      hipStream_t stream) {
  if (world <= 0 || capacity <= 0) return NDJIR_OK;
  if (!ids || !counts || !buf) return NDJIR_ERR_ARG;
  if (D != 4) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_rows_zero, dim3(blocks_for((long long)world * capacity)), dim3(256), 0, stream, ids, counts, world, capacity, buf);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_clear_bitmap(const int* ids, const int* count, int capacity, unsigned* bitmap, hipStream_t stream) {
  if (capacity <= 0) return NDJIR_OK;
  if (!ids || !count || !bitmap) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_rows_clear_bitmap, dim3(blocks_for(capacity)), dim3(256), 0, stream, ids, count, capacity, bitmap);
  return ndjir_check_launch();
}
