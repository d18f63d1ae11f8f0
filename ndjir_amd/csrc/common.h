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
