// kres.hip -- feasibility microbenchmark of a "weights resident in registers" f16x3 chain layer on gfx950.
//
// mlp3w.hip runs k-loop, epilogue and split of a layer in lock-step phases: the matrix pipe is busy 23 % of the time and
// the VALU work of the epilogue (18 issue slots per element) is serial to it.  Here a workgroup has FOUR waves (one per
// SIMD, up to 512 registers each).  A wave owns 2 column blocks x 4 row blocks of a 128-point x 256-column layer; its whole
// weight slice (K = 256 x 64 columns x 2 planes = 64 KB) is loaded into registers once per layer, the k-loops run row block
// by row block, and the epilogue of row block j - 1 (one accumulator element of each of its two blocks per k-step) is
// written between the MFMAs of row block j: the VALU work sits in the shadow of the wave's own MFMAs.
//   mode 0: k-loops only            mode 1: + interleaved epilogue (softplus, side stores)
//   mode 2: + row maxima, barrier, 2-way split into the planes (a complete regular forward layer)
// Prints fp32-equivalent TFLOP/s (3 MFMA partial products per product; 833 = the matrix pipe's limit).
// build: hipcc -O3 --offload-arch=gfx950 kres.hip -o kres
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int K = 256, N = 256, KS = K / 16, TM = 128, TMP = TM + 4, LAYERS = 8;
constexpr int PLANE = (K / 8) * TMP;          // 16-byte units per plane
constexpr float LO_INV = 1.f / 2048.f, LO_SCALE = 2048.f;

template <int MODE>
__global__ void __launch_bounds__(256) k_res(const f16x8* __restrict__ W, float* __restrict__ side, float* __restrict__ out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned s_rmax[TM];
  f16x8* act = reinterpret_cast<f16x8*>(lds);
  char* actb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 2 * PLANE * 4; i += 256) lds[i] = 0.f;
  for (int i = tid; i < 2 * PLANE * 8; i += 256) reinterpret_cast<_Float16*>(lds)[i] = (_Float16)(0.01f * ((i * 7) & 15));
  if (tid < TM) s_rmax[tid] = 0u;
  __syncthreads();
  const float kk = 1.0e-9f, bb = 0.1f, ib2sc = 0.00693f;
  float sink = 0.f;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = (long long)t * TM;
    for (int l = 0; l < LAYERS; ++l) {
      // ---- the wave's weight slice -> registers ----
      f16x8 wf[2][KS][2];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[n][ks][p] = W[((long long)((l * 8 + wave * 2 + n) * KS + ks) * 2 + p) * 64 + lane];
      f32x16 res[4][2];
      const f16x8* A0 = act + h * TMP + r;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x16 a0[2] = {f32x16{0}, f32x16{0}}, a1[2] = {f32x16{0}, f32x16{0}};
        f16x8 afh = A0[rb * 32], afl = A0[PLANE + rb * 32];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          f16x8 nh = afh, nl = afl;
          if (ks + 1 < KS) { nh = A0[2 * (ks + 1) * TMP + rb * 32]; nl = A0[PLANE + 2 * (ks + 1) * TMP + rb * 32]; }
#pragma unroll
          for (int n = 0; n < 2; ++n) a1[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][ks][0], afl, a1[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 2; ++n) a0[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][ks][0], afh, a0[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 2; ++n) a1[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[n][ks][1], afh, a1[n], 0, 0, 0);
          if (MODE >= 1 && rb > 0) {
            // epilogue of row block rb - 1, element ks of both column blocks
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              const float u = fmaf(res[rb - 1][n][ks], kk, bb);
              const float l2 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(u)));
              res[rb - 1][n][ks] = (fmaxf(u, 0.f) + l2) * ib2sc;
            }
            if ((ks & 3) == 3) {
#pragma unroll
              for (int n = 0; n < 2; ++n) {
                float* o = side + ((row0 + (rb - 1) * 32 + r) * N + (wave * 2 + n) * 32 + 8 * (ks >> 2) + 4 * h);
                *reinterpret_cast<f32x4*>(o) = f32x4{res[rb - 1][n][ks - 3], res[rb - 1][n][ks - 2], res[rb - 1][n][ks - 1], res[rb - 1][n][ks]};
              }
            }
          }
          afh = nh; afl = nl;
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) res[rb][n][i] = fmaf(a1[n][i], LO_INV, a0[n][i]);
      }
      if (MODE >= 1) {      // the last row block's epilogue has no k-loop to hide under
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float u = fmaf(res[3][n][i], kk, bb);
            const float l2 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(u)));
            res[3][n][i] = (fmaxf(u, 0.f) + l2) * ib2sc;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float* o = side + ((row0 + 3 * 32 + r) * N + (wave * 2 + n) * 32 + 8 * g + 4 * h);
            *reinterpret_cast<f32x4*>(o) = f32x4{res[3][n][4 * g], res[3][n][4 * g + 1], res[3][n][4 * g + 2], res[3][n][4 * g + 3]};
          }
        }
      }
      if (MODE >= 2) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          float m = 0.f;
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(res[rb][n][i]));
          atomicMax(&s_rmax[rb * 32 + r], __float_as_uint(m));
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const unsigned mb = s_rmax[rb * 32 + r];
          const float s = __uint_as_float((268u - (mb >> 23)) << 23);
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 xs = f32x4{res[rb][n][4 * g], res[rb][n][4 * g + 1], res[rb][n][4 * g + 2], res[rb][n][4 * g + 3]} * s;
              const f16x4 ph = __builtin_convertvector(xs, f16x4);
              const f32x4 rs = (xs - __builtin_convertvector(ph, f32x4)) * LO_SCALE;
              const f16x4 pl = __builtin_convertvector(rs, f16x4);
              const int k = (wave * 2 + n) * 32 + 8 * g + 4 * h;
              char* p = actb + ((size_t)((k >> 3) * TMP + rb * 32 + r) * 16 + (k & 7) * 2);
              *reinterpret_cast<f16x4*>(p) = ph;
              *reinterpret_cast<f16x4*>(p + (size_t)PLANE * 16) = pl;
            }
        }
        __syncthreads();
        if (tid < TM) s_rmax[tid] = 0u;
      } else {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) sink += res[rb][n][i];
        __syncthreads();
      }
    }
  }
  if (sink == 123.456f) out[tid] = sink;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 512;      // 65536 points
  const size_t wn = (size_t)LAYERS * 8 * KS * 2 * 64;    // f16x8 units
  std::vector<_Float16> hw(wn * 8);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.001f * (float)((i * 13) % 31));
  f16x8* W;
  float *side, *out;
  hipMalloc(&W, wn * 16);
  hipMemcpy(W, hw.data(), wn * 16, hipMemcpyHostToDevice);
  hipMalloc(&side, (size_t)tiles * TM * N * 4);
  hipMalloc(&out, 4096);
  const size_t lds = (size_t)2 * PLANE * 16;
  const double flops = 2.0 * tiles * TM * (double)K * N * LAYERS;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
#define RUN(M)                                                                                                         \
  {                                                                                                                    \
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_res<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    const int blocks = tiles < 256 ? tiles : 256;                                                                     \
    hipLaunchKernelGGL(k_res<M>, dim3(blocks), dim3(256), lds, 0, W, side, out, tiles);                                \
    hipDeviceSynchronize();                                                                                            \
    hipEventRecord(e0);                                                                                                \
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_res<M>, dim3(blocks), dim3(256), lds, 0, W, side, out, tiles);   \
    hipEventRecord(e1);                                                                                                \
    hipEventSynchronize(e1);                                                                                           \
    float ms;                                                                                                          \
    hipEventElapsedTime(&ms, e0, e1);                                                                                  \
    printf("mode %d: %.1f us per launch (%d tiles x %d layers of 128 x 256 x 256)  %.1f TFLOP/s fp32-equivalent  (%s)\n", M,   \
           1e3 * ms / 10, tiles, LAYERS, flops / (ms / 10 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));       \
  }
  RUN(0) RUN(1) RUN(2)
  return 0;
}
