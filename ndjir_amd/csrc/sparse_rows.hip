// sparse_rows.hip -- the receiving side of the sparse grid-gradient exchange of the ray-sharded step (SURVEY.md §8 e).
// The reference has no distributed code; a dense grid gradient (2 GiB for the 512^3 x 4 voxel grid, 403 MB for the
// 3 x 2048^2 x 8 tri-plane) is exchanged as packed lists of (cell id, row of D floats) per rank
// (ndjir_grid_pack_rows), all-gathered with a common, fixed number of rows:
//   lists: ids (world, limit) int32, rows (world, limit, D) fp32 -- packed with row stride `limit`, the layout one
//   all_gather_into_tensor of every rank's first `limit` rows leaves behind (no per-rank copies) --, counts (world) int32,
//   all in device memory; `limit` = rows per rank that were actually communicated (a rank whose count exceeds it raised
//   the overflow flag and entered the statistics from which the host grows `limit`).
#include <hip/hip_runtime.h>

#include "common.h"

namespace ndjir {

// Packed wire (counts == nullptr): every rank's list travels with a header of ROWS_HDR ints in front of its ids --
// [count, capacity the rank needs, 0, 0] -- so that the counts need no collective of their own: the id lists are then
// (world, limit + ROWS_HDR) ints, the rows stay (world, limit, D).
constexpr int ROWS_HDR = 4;

// buf[cell] += row for every listed row of every rank but `skip_rank` (this rank's own rows are already in buf)
template <int D4>
__global__ void __launch_bounds__(256) k_rows_apply(const int* __restrict__ ids, const float4* __restrict__ rows,
                                                    const int* __restrict__ counts, int world, int cap, int limit, int skip_rank,
                                                    float* __restrict__ buf) {
  // one lane per FLOAT of a row: the 4 D4 lanes of a row address contiguous bytes -- the memory-side atomic unit serves a request,
  // not a float, at its ~20 G / s (tools/ubench/atomics_shape.hip)
  constexpr int D = 4 * D4;
  const long long total = (long long)world * limit * D;
  const float* rowf = reinterpret_cast<const float*>(rows);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long row = t / D;
    const int d = (int)(t - row * D);
    const int r = (int)(row / limit), i = (int)(row - (long long)r * limit);
    const int cnt = counts ? counts[r] : ids[(long long)r * (cap + ROWS_HDR)];
    if (r == skip_rank || i >= cnt) continue;
    const long long e = (long long)r * cap + i;
    const float v = rowf[e * D + d];
    const int cell = counts ? ids[e] : ids[(long long)r * (cap + ROWS_HDR) + ROWS_HDR + i];
    if (v != 0.f) atomicAdd(buf + (long long)cell * D + d, v);
  }
}

// buf[cell] = 0 for every listed row: the other ranks' communicated rows and ALL of this rank's own rows (own_ids /
// own_count: the local list, which may be longer than `limit`) -- re-arms the accumulate-in-place buffer
// (`limit` is read from device memory: the call may be replayed from a captured HIP graph after the limit has grown)
// `cap`: capacity of the own list (and upper bound of `limit`)
template <int D4>
__global__ void __launch_bounds__(256) k_rows_zero(const int* __restrict__ ids, const int* __restrict__ counts, int world, int cap,
                                                   const int* __restrict__ limit_p, int own_rank, const int* __restrict__ own_ids,
                                                   const int* __restrict__ own_count, float* __restrict__ buf) {
  int limit = *limit_p;
  if (limit > cap) limit = cap;
  const long long remote = (long long)world * limit;
  int own = own_ids ? *own_count : 0;
  if (own > cap) own = cap;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < remote + own; t += (long long)gridDim.x * 256) {
    int cell;
    if (t < remote) {
      const int r = (int)(t / limit), i = (int)(t - (long long)r * limit);
      const int cnt = counts ? counts[r] : ids[(long long)r * (limit + ROWS_HDR)];
      if ((own_ids && r == own_rank) || i >= cnt) continue;
      cell = counts ? ids[(long long)r * limit + i]          // (packed lists: row stride = the communicated size)
                    : ids[(long long)r * (limit + ROWS_HDR) + ROWS_HDR + i];
    } else {
      cell = own_ids[t - remote];
    }
    float4* p = reinterpret_cast<float4*>(buf + (long long)cell * (4 * D4));
#pragma unroll
    for (int c = 0; c < D4; ++c) p[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Dense fall-back of the re-arm: when this rank's list overflowed its capacity (*own_count > capacity), k_pack_rows dropped
// cells that hold gradient in `buf` and that no list names -- k_rows_zero cannot clear them.  The step that dropped them is
// vetoed (its count exceeds every wire size), and the whole buffer is cleared here so that nothing stale is added to the next
// step's sums.  In every other step the kernel reads one int and returns.
__global__ void __launch_bounds__(256) k_rows_zero_if_dropped(const int* __restrict__ own_count, int capacity, float4* __restrict__ buf,
                                                              long long n4) {
  if (*own_count <= capacity) return;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n4; t += (long long)gridDim.x * 256) buf[t] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// clears the bitmap words of the listed cells (every set bit belongs to a listed cell)
__global__ void __launch_bounds__(256) k_rows_clear_bitmap(const int* __restrict__ ids, const int* __restrict__ count, int capacity,
                                                           unsigned* __restrict__ bitmap) {
  int n = *count;
  if (n > capacity) n = capacity;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) bitmap[(unsigned)ids[t] >> 5] = 0u;
}

// *flag |= 1 when a rank listed more rows than were communicated (the step's grid gradient is then incomplete: the
// caller vetoes the optimizer step on the device and grows the communicated size).  stats (may be null): [0] running
// maximum of the counts over all exchanges -- what the host sizes `limit` from at its next look, whichever exchange
// overflowed --, [1] number of exchanges that overflowed.
__global__ void k_rows_overflow(const int* __restrict__ counts, int world, int limit, int* __restrict__ flag,
                                int* __restrict__ stats, int stride) {
  bool over = false;
  int most = 0;
  for (int r = threadIdx.x; r < world; r += 64) {
    const int c = counts[(long long)r * stride];
    over |= c > limit; most = c > most ? c : most;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(most, off); most = o > most ? o : most; }
  const bool any_over = __any(over);
  if (threadIdx.x == 0) {
    if (any_over) *flag = 1;
    if (stats) { if (most > stats[0]) stats[0] = most; if (any_over) stats[1] += 1; }
  }
}

static int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace ndjir

using namespace ndjir;

extern "C" int ndjir_sparse_rows_apply(const int* ids, const float* rows, const int* counts, int world, int capacity, int limit,
                                       int skip_rank, float* buf, int D, hipStream_t stream) {
  if (world <= 0 || capacity <= 0 || limit <= 0) return NDJIR_OK;
  if (!ids || !rows || !buf || limit > capacity) return NDJIR_ERR_ARG;      // (counts == null: packed wire, see ROWS_HDR)
  if (D != 4 && D != 8) return NDJIR_ERR_UNSUPPORTED;
  const int blocks = blocks_for((long long)world * limit * D);
  if (D == 4) hipLaunchKernelGGL(k_rows_apply<1>, dim3(blocks), dim3(256), 0, stream, ids, reinterpret_cast<const float4*>(rows), counts,
                                 world, capacity, limit, skip_rank, buf);
  else hipLaunchKernelGGL(k_rows_apply<2>, dim3(blocks), dim3(256), 0, stream, ids, reinterpret_cast<const float4*>(rows), counts, world,
                          capacity, limit, skip_rank, buf);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_zero(const int* ids, const int* counts, int world, int capacity, const int* limit, int own_rank,
                                      const int* own_ids, const int* own_count, float* buf, int D, hipStream_t stream) {
  if (world <= 0 || capacity <= 0) return NDJIR_OK;
  if (!ids || !buf || !limit || ((own_ids == nullptr) != (own_count == nullptr))) return NDJIR_ERR_ARG;
  if (D != 4 && D != 8) return NDJIR_ERR_UNSUPPORTED;
  const int blocks = blocks_for((long long)(world + 1) * capacity);      // grid-stride over world x *limit (+ own rows)
  if (D == 4) hipLaunchKernelGGL(k_rows_zero<1>, dim3(blocks), dim3(256), 0, stream, ids, counts, world, capacity, limit, own_rank, own_ids,
                                 own_count, buf);
  else hipLaunchKernelGGL(k_rows_zero<2>, dim3(blocks), dim3(256), 0, stream, ids, counts, world, capacity, limit, own_rank, own_ids,
                          own_count, buf);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_zero_if_dropped(const int* own_count, int capacity, float* buf, long long n, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!own_count || !buf || capacity < 0 || (n & 3) != 0) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_rows_zero_if_dropped, dim3(2048), dim3(256), 0, stream, own_count, capacity, reinterpret_cast<float4*>(buf), n / 4);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_clear_bitmap(const int* ids, const int* count, int capacity, unsigned* bitmap, hipStream_t stream) {
  if (capacity <= 0) return NDJIR_OK;
  if (!ids || !count || !bitmap) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_rows_clear_bitmap, dim3(blocks_for(capacity)), dim3(256), 0, stream, ids, count, capacity, bitmap);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_overflow(const int* counts, int world, int limit, int* flag, int* stats, int counts_stride,
                                          hipStream_t stream) {
  if (world <= 0) return NDJIR_OK;
  if (!counts || !flag || counts_stride < 1) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_rows_overflow, dim3(1), dim3(64), 0, stream, counts, world, limit, flag, stats, counts_stride);
  return ndjir_check_launch();
}

extern "C" int ndjir_sparse_rows_header(void) { return ndjir::ROWS_HDR; }
