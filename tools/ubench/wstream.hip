// wstream.hip -- what the L2 -> CU path delivers for the chain kernels' weight stream: every CU reads the SAME packed matrices
// (L2-resident after the first touch), wave w of a workgroup streams column block w: per k-step two 1 KB fragment loads
// (global_load_dwordx4, 64 lanes x 16 B), DEPTH k-steps in flight, nothing else going on.  Prints bytes / clk / CU.
// usage: wstream [workgroups (256)] [layers per pass (8)]      build: hipcc -O3 --offload-arch=gfx950 wstream.hip -o wstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_stream(const f32x4* __restrict__ W, float* __restrict__ out, int layers, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int KS = 16;                            // k-steps per layer (256 input features)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r)
    for (int l = 0; l < layers; ++l) {
      // layer l: 8 column blocks x 16 k-steps x 2 planes x 64 lanes of 16 bytes = 256 KB
      const f32x4* p = W + ((size_t)l * 8 + (wave & 7)) * KS * 2 * 64 + lane;
      f32x4 b[DEPTH][2];
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) { b[s][0] = p[(s * 2) * 64]; b[s][1] = p[(s * 2 + 1) * 64]; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc += b[ks % DEPTH][0] + b[ks % DEPTH][1];
        if (ks + DEPTH < KS) { b[ks % DEPTH][0] = p[((ks + DEPTH) * 2) * 64]; b[ks % DEPTH][1] = p[((ks + DEPTH) * 2 + 1) * 64]; }
      }
    }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[threadIdx.x] = acc[0];
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, layers = argc > 2 ? atoi(argv[2]) : 8, reps = 64;
  const size_t n = (size_t)layers * 8 * 16 * 2 * 64;      // f32x4 elements
  f32x4* W; float* out;
  (void)hipMalloc(&W, n * 16); (void)hipMalloc(&out, 4096); (void)hipMemset(W, 0, n * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int clk_khz = 2400000; (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
#define RUN(D, WV)                                                                                                      \
  {                                                                                                                     \
    hipLaunchKernelGGL((k_stream<D, WV>), dim3(blocks), dim3(WV * 64), 0, 0, W, out, layers, 2);                        \
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);                                                             \
    hipLaunchKernelGGL((k_stream<D, WV>), dim3(blocks), dim3(WV * 64), 0, 0, W, out, layers, reps);                     \
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                            \
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                                   \
    const double bytes = (double)reps * layers * WV * 16 * 2 * 1024, sec = ms * 1e-3;                                   \
    printf("depth %2d k-steps, %2d waves / workgroup: %8.1f us  %6.1f B/clk/CU (at %.2f GHz)  %6.1f TB/s chip\n", D, WV, sec * 1e6, \
           bytes / sec / (clk_khz * 1e3), clk_khz * 1e-6, bytes * blocks / sec / 1e12);                                 \
  }
  RUN(3, 8) RUN(6, 8) RUN(12, 8) RUN(3, 16) RUN(6, 16) RUN(3, 4) RUN(6, 4) RUN(12, 4)
  return 0;
}
