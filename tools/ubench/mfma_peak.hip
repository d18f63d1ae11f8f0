// fp32 MFMA issue-rate calibration on gfx950: NACC independent accumulators per wave, W waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = f32x16{0};
  float av = a + threadIdx.x * 1e-6f, bv = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[q], 0, 0, 0);
  }
  float s = 0;
  for (int q = 0; q < NACC; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int threads) {
  float* out; hipMalloc(&out, 256 * 16 * 1024 * 4);
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<256 * blocks_per_cu, threads>>>(out, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<256 * blocks_per_cu, threads>>>(out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = 256.0 * blocks_per_cu * threads / 64;
  double flops = waves * iters * 8.0 * NACC * 4096.0;
  printf("NACC=%d blocks/CU=%d threads=%d (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, threads,
         blocks_per_cu * threads / 64 / 4.0, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<2>(2, 256); run<2>(1, 512); run<1>(2, 512); run<4>(2, 256);
  return 0;
}
