// affine.hip -- y = x W (+ b) for a FEW rows (the per-ray terms of the material / light nets' first layers, python/network.py:
// 438, 528, 619 -- 512 rows at the bench's size -- and their input gradients): one wave per 32 x 32 output tile, exact fp32
// products on v_mfma_f32_32x32x2_f32 (bitwise an fp32 FMA chain over k), operands read where they lie (no packed weights).
// Why: through the chain kernel (mlp3.hip, one layer) such a launch runs 16 tiles of 32 points on 16 CUs, each streaming the
// whole packed matrix and walking its column blocks in rounds -- 17 - 38 us for 17 MFLOP (profiles/r06_bench_kernel_summary.txt:
// four of the seven k_chain3<0, 32> launches of a step).  Here the 16 x 9 tiles of a 512 x 262 output are 144 independent waves.
#include <hip/hip_runtime.h>

#include "../../include/ndjir_hip.h"
#include "common.h"

namespace ndjir {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// TRANSPOSE: y = x W^T with W (N, K) row-major (the input gradient of a layer: g W^T)
template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) k_small_affine(long long P, const float* __restrict__ x, int ldx, int K,
                                                      const float* __restrict__ W, int ldw, int N,
                                                      const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                      int tiles_n, long long tiles) {
  const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per tile
  if (tile >= tiles) return;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const long long tm = tile / tiles_n;
  const int tn = (int)(tile - tm * tiles_n);
  const long long row = tm * 32 + r;
  const int col = tn * 32 + r;
  const bool row_ok = row < P, col_ok = col < N;
  // A operand of step k: x[row][k + h]; B operand: W[k + h][col] (or W[col][k + h])
  const float* xa = x + (row_ok ? row : 0) * (long long)ldx + h;
  const float* wb = TRANSPOSE ? W + (long long)(col_ok ? col : 0) * ldw + h : W + (long long)h * ldw + (col_ok ? col : 0);
  const long long wstep = TRANSPOSE ? 2 : 2LL * ldw;
  f32x16 acc = f32x16{0};
  int k = 0;
  for (; k + 16 <= K; k += 16) {        // eight steps' operands in flight, then eight MFMAs
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = xa[k + 2 * j]; b[j] = wb[(long long)(k / 2 + j) * wstep]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(row_ok ? a[j] : 0.f, col_ok ? b[j] : 0.f, acc, 0, 0, 0);
  }
  for (; k < K; k += 2) {
    const bool k_ok = k + h < K;
    const float a = (row_ok && k_ok) ? xa[k] : 0.f;
    const float b = (col_ok && k_ok) ? wb[(long long)(k / 2) * wstep] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // register i of lane (r, h): row 8 (i / 4) + 4 h + (i % 4) of the tile, column r
  const float bv = (bias && col_ok) ? bias[col] : 0.f;
  if (col_ok) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long m = tm * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
      if (m < P) y[m * ldy + col] = acc[i] + bv;
    }
  }
}

}  // namespace ndjir

using namespace ndjir;

// y (P, N; row stride ldy) = x (P, K; ldx) W (K, N; ldw) + bias, or with transpose != 0: x W^T, W (N, K; ldw).  fp32 throughout.
extern "C" int ndjir_mlp_small_affine(long long P, const float* x, int ldx, int K, const float* W, int ldw, int N, int transpose,
                                      const float* bias, float* y, int ldy, hipStream_t stream) {
  if (P <= 0 || N <= 0) return NDJIR_OK;
  if (!x || !W || !y || K <= 0 || ldx < K || ldy < N || ldw < (transpose ? K : N)) return NDJIR_ERR_ARG;
  const int tiles_n = (N + 31) / 32;
  const long long tiles = ((P + 31) / 32) * tiles_n;
  const unsigned blocks = (unsigned)((tiles + 3) / 4);
  if (transpose) hipLaunchKernelGGL(k_small_affine<true>, dim3(blocks), dim3(256), 0, stream, P, x, ldx, K, W, ldw, N, bias, y, ldy, tiles_n, tiles);
  else hipLaunchKernelGGL(k_small_affine<false>, dim3(blocks), dim3(256), 0, stream, P, x, ldx, K, W, ldw, N, bias, y, ldy, tiles_n, tiles);
  return ndjir_check_launch();
}
