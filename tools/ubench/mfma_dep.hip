// How far apart do two v_mfma_f32_32x32x16_f16 on the SAME accumulator have to be?  NACC accumulators per wave, used round-robin
// (the distance between dependent MFMAs = NACC instructions), 1 or 2 waves per SIMD (256- / 512-thread workgroups, one per CU),
// optionally with GAP independent v_fma_f32 behind every MFMA (the chain kernels' interleave).  Prints shader cycles per MFMA
// PER SIMD (s_memtime of wave 0 of workgroup 0) -- 32 = the matrix pipe's rate.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_dep tools/ubench/mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int GAP>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  f32x16 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = f32x16{0};
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GAP; ++g) v[g & 7] = __builtin_fmaf(v[g & 7], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[q][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;      // (per wave: the older wave of a SIMD wins the arbitration)
}

template <int NACC, int GAP>
void run(int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
  const int iters = 2000;
  k<NACC, GAP><<<256, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  k<NACC, GAP><<<256, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  long long first = c[0], last = c[0];
  for (int w = 0; w < threads / 64; ++w) { if (c[w] < first) first = c[w]; if (c[w] > last) last = c[w]; }
  const double n = iters * 4.0 * NACC;
  printf("accumulators %d  gap %d VALU  waves/SIMD %d: fastest wave %6.1f, slowest wave %6.1f cycles per own MFMA -> %6.1f cycles per MFMA per SIMD\n",
         NACC, GAP, wps, first / n, last / n, last / n / wps);
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("v_mfma_f32_32x32x16_f16, round-robin over N accumulators (dependent distance = N MFMAs)\n");
  run<1, 0>(256); run<2, 0>(256); run<3, 0>(256); run<4, 0>(256); run<8, 0>(256);
  run<1, 0>(512); run<2, 0>(512); run<3, 0>(512); run<4, 0>(512); run<8, 0>(512);
  printf("... with independent v_fma_f32 behind every MFMA\n");
  run<2, 4>(256); run<2, 8>(256); run<4, 4>(256); run<4, 8>(256);
  run<2, 4>(512); run<2, 8>(512); run<4, 4>(512); run<4, 8>(512); run<8, 8>(512);
  return 0;
}
