// wgrad.hip -- weight-gradient GEMM  dW (K x N) = A^T B,  A (P x K) activations, B (P x N) deltas.
//
// The reduction runs over P = 10^4..10^5 points while K, N <= ~300: a "tall-skinny^T" GEMM that
// library heuristics serve with tiny tiles (measured 37 TFLOP/s).  Here: 128 x 128 output tiles,
// the point axis split over workgroups (partials + a reduce pass), 32-point chunks staged through
// double-buffered LDS, fp32 MFMA 32x32x2 with both operands read as conflict-free 128-byte rows
// (the contraction index is the row index of both matrices, so no transpose is ever needed).
// Workgroup -> (split, tile) mapping keeps the tiles that share a P-range on one XCD (shared L2).
#include <hip/hip_runtime.h>

#include "common.h"
#include "mlp.h"

namespace ndjir {

constexpr int WG_T = 128;     // output tile edge
constexpr int WG_C = 32;      // points per chunk
constexpr int WG_THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(WG_THREADS, 2) k_wgrad(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                         int ldb, int K, int N, long long P, float* __restrict__ partial,
                                                         int S, int tiles_k, int tiles_n, long long rows_per_split) {
  __shared__ float As[2][WG_C][WG_T];
  __shared__ float Bs[2][WG_C][WG_T];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave >> 1, wn = wave & 1;          // 2 x 2 waves, each 64 x 64
  const int T = tiles_k * tiles_n;
  // XCD-aware: blocks are dealt round-robin to the 8 XCDs; give every XCD whole splits
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int vid = (nblk & 7) == 0 ? xcd * (nblk >> 3) + local : blockIdx.x;
  const int split = vid / T, tile = vid - split * T;
  const int tk = tile / tiles_n, tn = tile - tk * tiles_n;
  const int k0 = tk * WG_T, n0 = tn * WG_T;
  const long long p_begin = (long long)split * rows_per_split;
  long long p_end = p_begin + rows_per_split;
  if (p_end > P) p_end = P;

  f32x16 acc[2][2] = {};
  float ra[16], rb[16];

  // element e = tid + 256 i of a 32 x 128 chunk: column = tid & 127 (fixed), row = (tid >> 7) + 2 i
  const int col = tid & (WG_T - 1), rbase = tid >> 7;
  const bool acol = (k0 + col) < K, bcol = (n0 + col) < N;
  const float* Ap = A + (long long)rbase * lda + k0 + col;
  const float* Bp = B + (long long)rbase * ldb + n0 + col;
  const long long astep = 2LL * lda, bstep = 2LL * ldb;
  auto load_chunk = [&](long long p0) {
    const float* ap = Ap + p0 * lda;
    const float* bp = Bp + p0 * ldb;
    if (p0 + WG_C <= p_end) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        ra[i] = acol ? ap[i * astep] : 0.f;
        rb[i] = bcol ? bp[i * bstep] : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        bool pv = p0 + rbase + 2 * i < p_end;
        ra[i] = (pv && acol) ? ap[i * astep] : 0.f;
        rb[i] = (pv && bcol) ? bp[i * bstep] : 0.f;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      As[buf][rbase + 2 * i][col] = ra[i];
      Bs[buf][rbase + 2 * i][col] = rb[i];
    }
  };

  if (p_begin < p_end) {
    load_chunk(p_begin);
    store_chunk(0);
    __syncthreads();
    int cur = 0;
    for (long long p0 = p_begin; p0 < p_end; p0 += WG_C) {
      const bool more = p0 + WG_C < p_end;
      if (more) load_chunk(p0 + WG_C);
#pragma unroll 4
      for (int s = 0; s < WG_C / 2; ++s) {
        const float* ar = &As[cur][2 * s + h][wk * 64 + r];
        const float* br = &Bs[cur][2 * s + h][wn * 64 + r];
        float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      if (more) store_chunk(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  float* out = partial + (long long)split * K * N;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const int n = n0 + wn * 64 + bj * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + wk * 64 + bi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < K && n < N) out[(long long)k * N + n] = acc[bi][bj][i];
      }
    }
}

__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ out,
                                                      long long KN, int S, int accum) {
  // one float4 of the output per thread; 8 independent loads in flight per thread
  const long long n4 = KN >> 2;
  const bool vec = ((KN & 3) == 0);
  if (vec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* p = reinterpret_cast<const float4*>(partial) + i;
      int k = 0;
      for (; k + 8 <= S; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; k < S; ++k) { float4 v = p[(long long)k * n4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      float4* o = reinterpret_cast<float4*>(out) + i;
      if (accum) { float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
      *o = s;
    }
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < KN; i += (long long)gridDim.x * blockDim.x) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < S; ++k) s += partial[(long long)k * KN + i];
      out[i] = accum ? out[i] + s : s;
    }
  }
}

// workspace floats needed for (K, N): S * K * N with the S chosen below
static inline int pick_splits(int K, int N, long long P) {
  int T = ((K + WG_T - 1) / WG_T) * ((N + WG_T - 1) / WG_T);
  int S = (256 + T - 1) / T;                       // ~1 workgroup (4 waves) per CU keeps every SIMD's matrix pipe busy
  long long max_s = (P + WG_C - 1) / WG_C;
  if (S > max_s) S = (int)max_s;
  if (S < 1) S = 1;
  while ((S * T) & 7) ++S;                         // whole splits per XCD
  return S;
}

long long wgrad_workspace(int K, int N, long long P) { return (long long)pick_splits(K, N, P) * K * N; }

int launch_wgrad(const float* A, int lda, const float* B, int ldb, int K, int N, long long P, float* out, int accum,
                 float* workspace, hipStream_t stream) {
  if (K <= 0 || N <= 0) return NDJIR_OK;
  const int tiles_k = (K + WG_T - 1) / WG_T, tiles_n = (N + WG_T - 1) / WG_T;
  const int S = pick_splits(K, N, P);
  long long rows = (P + S - 1) / S;
  rows = (rows + WG_C - 1) / WG_C * WG_C;
  hipLaunchKernelGGL(k_wgrad, dim3(S * tiles_k * tiles_n), dim3(WG_THREADS), 0, stream, A, lda, B, ldb, K, N, P, workspace, S,
                     tiles_k, tiles_n, rows);
  const long long KN = (long long)K * N;
  int blocks = (int)(((KN + 3) / 4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, stream, workspace, out, KN, S, accum);
  return ndjir_check_launch();
}

}  // namespace ndjir
