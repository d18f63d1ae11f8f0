// sidepat.hip -- what a CU's vector-memory path delivers for the side traffic of mlp3w.hip's epilogue.
// A 128-point x 256-column fp32 tile-layer (128 KB, row stride 1 KB) per workgroup of 8 waves; a wave owns one 32-column block and four
// 32-row blocks, as in the chain kernel.  Pattern A (the operand-swapped accumulator layout): lane (r = lane & 31, hh = lane >> 5)
// moves 16 bytes at row r, byte 32 g + 16 hh of the column block, g = 0..3 -- every instruction touches 32 rows x 32 bytes.
// Pattern B (full rows): lane l moves 16 bytes at row (l >> 3) + 8 i, byte 16 (l & 7) -- every instruction touches 8 rows x 128 bytes.
// usage: sidepat [workgroups [tiles]] -- fewer workgroups than CUs show what ONE CU can move when the chip is otherwise idle.
// Modes: load only / store only / load + store (other array).  Prints GB/s over the chip and bytes / clk / CU at the measured clock.
// build: hipcc -O3 --offload-arch=gfx950 sidepat.hip -o sidepat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int MODE>
__global__ void __launch_bounds__(512) k_side(const float* __restrict__ in, float* __restrict__ out, int tiles, int layers) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int l = 0; l < layers; ++l) {
      const size_t base = ((size_t)l * tiles + t) * 128 * 256;          // floats
#pragma unroll
      for (int J = 0; J < 4; ++J) {
        f32x4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          size_t off;
          if (PAT == 0) off = base + (size_t)(J * 32 + (lane & 31)) * 256 + wave * 32 + 8 * g + 4 * (lane >> 5);
          else off = base + (size_t)(J * 32 + 8 * g + (lane >> 3)) * 256 + wave * 32 + 4 * (lane & 7);
          if (PAT == 3) off = base + ((size_t)J * 256 + wave * 32 + 8 * g + 4 * (lane >> 5)) * 32 + (lane & 31) * 4;
          if (PAT == 2) {
            // point-blocked layout [row block][feature][32 points]: register q of group g = feature 8 g + 4 hh + q, one dword per lane
            const size_t o2 = base + ((size_t)J * 256 + wave * 32 + 8 * g + 4 * (lane >> 5)) * 32 + (lane & 31);
            if (MODE != 1) { v[g] = f32x4{in[o2], in[o2 + 32], in[o2 + 64], in[o2 + 96]}; }
            else v[g] = f32x4{(float)o2, 1.f, 2.f, 3.f};
          } else {
          if (MODE != 1) v[g] = *reinterpret_cast<const f32x4*>(in + off);
          else v[g] = f32x4{(float)off, 1.f, 2.f, 3.f};
          }
          if (MODE == 0) acc += v[g];
        }
        if (MODE != 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            size_t off;
            if (PAT == 0) off = base + (size_t)(J * 32 + (lane & 31)) * 256 + wave * 32 + 8 * g + 4 * (lane >> 5);
            else off = base + (size_t)(J * 32 + 8 * g + (lane >> 3)) * 256 + wave * 32 + 4 * (lane & 7);
            if (PAT == 3) off = base + ((size_t)J * 256 + wave * 32 + 8 * g + 4 * (lane >> 5)) * 32 + (lane & 31) * 4;
            if (PAT == 2) {
              const size_t o2 = base + ((size_t)J * 256 + wave * 32 + 8 * g + 4 * (lane >> 5)) * 32 + (lane & 31);
              const f32x4 w = v[g] * 1.5f;
              out[o2] = w[0]; out[o2 + 32] = w[1]; out[o2 + 64] = w[2]; out[o2 + 96] = w[3];
            } else
            *reinterpret_cast<f32x4*>(out + off) = v[g] * 1.5f;
          }
        }
      }
    }
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[tid] = acc[0];
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;       // workgroups = CUs in use (one 8-wave workgroup per CU)
  const int tiles = argc > 2 ? atoi(argv[2]) : 512, layers = 8;   // 65536 points x 8 layers x 1 KB = 537 MB per array
  const size_t n = (size_t)tiles * layers * 128 * 256;
  float *in, *out;
  hipMalloc(&in, n * 4);
  hipMalloc(&out, n * 4);
  hipMemset(in, 0, n * 4);
  hipMemset(out, 0, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int clk_khz = 2400000;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
#define RUN(P, M, NAME, BYTES)                                                                                         \
  {                                                                                                                    \
    hipLaunchKernelGGL((k_side<P, M>), dim3(blocks), dim3(512), 0, 0, in, out, tiles, layers);                         \
    hipDeviceSynchronize();                                                                                            \
    hipEventRecord(e0);                                                                                                \
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_side<P, M>), dim3(blocks), dim3(512), 0, 0, in, out, tiles, layers); \
    hipEventRecord(e1);                                                                                                \
    hipEventSynchronize(e1);                                                                                           \
    float ms;                                                                                                          \
    hipEventElapsedTime(&ms, e0, e1);                                                                                  \
    const double bytes = (double)(BYTES) * n * 4, sec = ms / 5 * 1e-3;                                                 \
    printf("%-44s %7.1f us  %6.0f GB/s  %5.1f B/clk/CU (at %.2f GHz)\n", NAME, sec * 1e6, bytes / sec / 1e9,          \
           bytes / sec / blocks / (clk_khz * 1e3), clk_khz * 1e-6);                                                       \
  }
  RUN(0, 0, "A: 32 rows x 32 B per instruction, load", 1)
  RUN(0, 1, "A: 32 rows x 32 B per instruction, store", 1)
  RUN(0, 2, "A: 32 rows x 32 B per instruction, load+store", 2)
  RUN(1, 0, "B: 8 rows x 128 B per instruction, load", 1)
  RUN(1, 1, "B: 8 rows x 128 B per instruction, store", 1)
  RUN(1, 2, "B: 8 rows x 128 B per instruction, load+store", 2)
  RUN(2, 0, "C: point-blocked, dword (2 x 128 B lines), load", 1)
  RUN(2, 1, "C: point-blocked, dword, store", 1)
  RUN(2, 2, "C: point-blocked, dword, load+store", 2)
  RUN(3, 0, "D: quad-blocked, dwordx4 (2 x 512 B), load", 1)
  RUN(3, 1, "D: quad-blocked, dwordx4, store", 1)
  RUN(3, 2, "D: quad-blocked, dwordx4, load+store", 2)
  return 0;
}
