// kloop6.hip -- feasibility microbenchmark for the bf16 3-way-split ("x6") chain k-loop on gfx950:
// every workgroup (8 waves) runs the k-loops of a stack of 256x256 layers for TM = 64 points:
//   fp32 : 1 KiB weight fragment (global, L2) + 2 ds_read_b128 per 8 v_mfma_f32_32x32x2_f32     (today)
//   x6   : 3 KiB weight fragments            + 6 ds_read_b128 per 12 v_mfma_f32_32x32x16_bf16  (6 products)
// No epilogue, no barriers except one per layer: the question is only whether L2 -> CU weight
// streaming and LDS keep up with the 2.67x shorter matrix time.  Prints fp32-equivalent TFLOP/s.
// build: hipcc -O3 --offload-arch=gfx950 kloop6.hip -o kloop6
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 256, N = 256, TM = 64, LAYERS = 8;

__global__ void __launch_bounds__(512, 2) k_fp32(const float* __restrict__ W, float* __restrict__ out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int GP = TM * 4 + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < (K / 4) * GP; i += 512) lds[i] = 0.001f * (i & 7);
  __syncthreads();
  f32x16 acc[2] = {};
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int l = 0; l < LAYERS; ++l) {
      const f32x4* Bp = reinterpret_cast<const f32x4*>(W) + ((long long)(l * 8 + wave) * (K / 8)) * 64 + lane;
      const float* A0 = lds + h * GP + r * 4;
      f32x4 b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = Bp[i * 64];
      for (int kb = 0; kb < K / 8; kb += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* An = A0 + (kb + i) * 2 * GP;
          f32x4 a0 = *reinterpret_cast<const f32x4*>(An), a1 = *reinterpret_cast<const f32x4*>(An + 32 * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[i][j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[i][j], acc[1], 0, 0, 0);
          }
          if (kb + i + 4 < K / 8) b[i] = Bp[(long long)(kb + i + 4) * 64];
        }
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// LDS: 3 planes [K/8][TMP][8 bf16]
constexpr int TMP = 68;
__global__ void __launch_bounds__(512, 2) k_x6(const bf16x8* __restrict__ W, float* __restrict__ out, int tiles, int pf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* act = reinterpret_cast<bf16x8*>(lds);
  constexpr int PLANE = (K / 8) * TMP;     // in 16-byte units
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 3 * PLANE * 4; i += 512) lds[i] = 0.f;
  __syncthreads();
  f32x16 acc[2] = {};
  constexpr int KS = K / 16;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int l = 0; l < LAYERS; ++l) {
      // packed weights: [layer][nb][ks][plane][lane]
      const bf16x8* Bp = W + ((long long)(l * 8 + wave) * KS) * 3 * 64 + lane;
      bf16x8 b[3][3];          // [prefetch slot][plane]
      bf16x8 a[2][3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][p] = Bp[(long long)(s * 3 + p) * 64];
      for (int ks = 0; ks < KS; ks += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (ks + s < KS) {
            static_assert(true, "");
            if (!(pf & 2) || (ks + s) == 0) {
#pragma unroll
              for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[q][p] = act[p * PLANE + (2 * (ks + s) + h) * TMP + q * 32 + r];
            }
            // 6 products, small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); planes 0 = hi, 1 = mid, 2 = lo
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][2], b[s][0], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][0], b[s][2], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][1], b[s][1], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][1], b[s][0], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][0], b[s][1], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][0], b[s][0], acc[q], 0, 0, 0);
            if (ks + s + 3 < KS && !(pf & 1)) {
#pragma unroll
              for (int p = 0; p < 3; ++p) b[s][p] = Bp[(long long)((ks + s + 3) * 3 + p) * 64];
            }
          }
        }
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// 16 waves (4 per SIMD, <= 128 VGPRs): one 32 x 32 block per wave (one accumulator chain), waves w and w + 8 share a
// column block (their weight-fragment loads coincide -> L1), more waves in flight to hide the accumulate and load latency
__global__ void __launch_bounds__(1024, 4) k_x6_16w(const bf16x8* __restrict__ W, float* __restrict__ out, int tiles, int pf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* act = reinterpret_cast<bf16x8*>(lds);
  constexpr int PLANE = (K / 8) * TMP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nb = wave & 7, rb = wave >> 3;
  for (int i = threadIdx.x; i < 3 * PLANE * 4; i += 1024) lds[i] = 0.f;
  __syncthreads();
  f32x16 acc = {};
  constexpr int KS = K / 16;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int l = 0; l < LAYERS; ++l) {
      const bf16x8* Bp = W + ((long long)(l * 8 + nb) * KS) * 3 * 64 + lane;
      bf16x8 b[3][3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[s][p] = Bp[(long long)(s * 3 + p) * 64];
      for (int ks = 0; ks < KS; ks += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (ks + s < KS) {
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = act[p * PLANE + (2 * (ks + s) + h) * TMP + rb * 32 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[s][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[s][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[s][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[s][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[s][0], acc, 0, 0, 0);
            if (ks + s + 3 < KS) {
#pragma unroll
              for (int p = 0; p < 3; ++p) b[s][p] = Bp[(long long)((ks + s + 3) * 3 + p) * 64];
            }
          }
        }
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 1024;
  float *W, *out;
  const size_t wbytes = (size_t)LAYERS * K * N * 6;
  hipMalloc(&W, wbytes);
  hipMemset(W, 0, wbytes);
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * tiles * TM * (double)K * N * LAYERS;
  for (int pfmode : {0, 1, 2, 3})
  for (int variant = (pfmode ? 1 : 0); variant < (pfmode ? 2 : 3); ++variant) {
    if (pfmode) printf("x6 probe: %s%s\n", (pfmode & 1) ? "[no weight reloads] " : "", (pfmode & 2) ? "[no activation reads]" : "");
    for (int grid : {256}) {
      size_t lds = variant == 0 ? (size_t)(K / 4) * (TM * 4 + 4) * 4 : (size_t)3 * (K / 8) * TMP * 16;
      if (variant == 0) hipFuncSetAttribute((const void*)k_fp32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      else if (variant == 1) hipFuncSetAttribute((const void*)k_x6, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      else hipFuncSetAttribute((const void*)k_x6_16w, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k_fp32, dim3(grid), dim3(512), lds, 0, W, out, tiles);
        else if (variant == 1) hipLaunchKernelGGL(k_x6, dim3(grid), dim3(512), lds, 0, reinterpret_cast<const bf16x8*>(W), out, tiles, pfmode);
        else hipLaunchKernelGGL(k_x6_16w, dim3(grid), dim3(1024), lds, 0, reinterpret_cast<const bf16x8*>(W), out, tiles, 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%s grid %d lds %zu KB: %.1f us  %.1f TFLOP/s (fp32-equivalent), err=%d\n", variant == 0 ? "fp32" : variant == 1 ? "x6  " : "x6 16 waves", grid,
             lds / 1024, best * 1e3, flop / (best * 1e-3) / 1e12, (int)hipGetLastError());
    }
  }
  return 0;
}
