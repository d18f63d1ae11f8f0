// common.h -- shared declarations of the ndjir_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#define NDJIR_OK 0
#define NDJIR_ERR_LAUNCH 1        // hipGetLastError() != hipSuccess after a launch
#define NDJIR_ERR_UNSUPPORTED 2   // argument combination has no kernel
#define NDJIR_ERR_ARG 3           // invalid argument (null pointer, bad size)

namespace ndjir {

// Unlike the reference (csrc/cuda_common.cuh:24-32 only printf's), launch errors are returned.
static inline int ndjir_check_launch() {
  return hipGetLastError() == hipSuccess ? NDJIR_OK : NDJIR_ERR_LAUNCH;
}

void zero_fill(float* p, long long n, hipStream_t stream);

}  // namespace ndjir

#ifdef __HIPCC__
// (row, column) of a flat element index over rows of W columns, advanced by a constant stride without a division per element: the
// element-wise kernels visit t, t + stride, ... and paid a 64-bit division for every element (k_geo_bwd_begin: 68 -> 3x less).
struct RowCol {
  long long p;
  int c, dq, dr, W;
  __device__ __forceinline__ RowCol(long long t, long long stride, int W_) : W(W_) {
    if (t <= 0x7fffffffLL && stride <= 0x7fffffffLL) {       // the usual case: 32-bit divisions
      const unsigned q = (unsigned)t / (unsigned)W_, sq = (unsigned)stride / (unsigned)W_;
      p = q; c = (int)((unsigned)t - q * (unsigned)W_); dq = (int)sq; dr = (int)((unsigned)stride - sq * (unsigned)W_);
    } else {
      p = t / W_; c = (int)(t - p * W_);
      const long long sq = stride / W_;
      dq = (int)sq; dr = (int)(stride - sq * W_);
    }
  }
  __device__ __forceinline__ void next() {
    p += dq; c += dr;
    if (c >= W) { c -= W; ++p; }
  }
};
#endif
