// Calibrates the two in-kernel clocks against HIP events: s_memtime (shader clock) and s_memrealtime (constant
// reference clock), idle chip vs all CUs busy with MFMA.  build: hipcc -O3 --offload-arch=gfx950 tools/ubench/clocks.hip -o tools/ubench/clocks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void spin(long long cycles, long long* out, int mfma) {
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc = {0};
  f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1};
  long long t;
  do {
    if (mfma) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
    }
    t = __builtin_amdgcn_s_memtime();
  } while (t - t0 < cycles);
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = r1 - r0; }
  if (acc[0] == 12345.f) out[2] = 1;
}

int main() {
  long long* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mfma = 0; mfma < 2; ++mfma)
    for (int blocks : {1, 256, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        spin<<<blocks, 512>>>(2000000, out, mfma);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("mfma=%d blocks=%4d: %.1f us by events; block 0: %lld shader cycles, %lld reference ticks -> shader %.2f GHz, reference %.1f MHz (if the block spans the kernel)\n",
             mfma, blocks, ms * 1e3, h[0], h[1], h[0] / (ms * 1e3) / 1e3, h[1] / (ms * 1e3));
    }
  // sustained MFMA load: 30 back-to-back kernels of ~2M cycles on all CUs, rate of every 5th
  for (int i = 0; i < 30; ++i) {
    hipEventRecord(e0);
    spin<<<1024, 512>>>(2000000, out, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (i % 5 == 4) {
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      printf("sustained mfma kernel %2d: %.1f us; block 0: %lld cycles in %lld ticks -> %.2f GHz\n", i, ms * 1e3, h[0], h[1], h[0] * 100.0 / h[1] / 1e3);
    }
  }
  return 0;
}
