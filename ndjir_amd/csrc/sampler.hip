// sampler.hip -- one up-sampling round of the hierarchical SDF-guided importance sampler
// (python/sampler.py:194-240: robust slope, sigmoid CDF, alpha, transmittance weights, inverse-
// transform sampling with deterministic u, clip, merge-sort) as ONE kernel: one wave per ray, lane l
// owns slots l, l+64, ... of the ray's <= 64*H samples (H = 2 up to 128 slots, the training shapes;
// H = 4 up to 256 slots, render_image at renderer.n_samples0 = 128); all intermediates live in LDS.
// The reference runs ~40 small nnabla launches per round.  Arithmetic and scan orders are the
// definitions of include/ndjir_math.h, shared bit-for-bit with the CPU oracle.
#include <hip/hip_runtime.h>

#include "../../include/ndjir_math.h"
#include "common.h"

#pragma clang fp contract(off)

namespace ndjir {

constexpr int RAYS_PER_BLOCK = 4;

// The scan / sum orders of ndjir_math.h do not depend on H: a Kogge-Stone prefix at slot i only
// involves slots <= i, and the lane sum (W[l] + W[l+64]) + (W[l+128] + W[l+192]) adds exact zeros
// when the upper slots are empty, so H = 2 and H = 4 give the same bits wherever both apply.
template <int H>
__global__ void __launch_bounds__(64 * RAYS_PER_BLOCK) k_importance_round(
    int R, int N, int M, float gain, float udenom, const float* __restrict__ t_in, const float* __restrict__ sdf_in,
    const float* __restrict__ t_near, const float* __restrict__ t_far, float* __restrict__ t_out, int* __restrict__ idx_out,
    int* __restrict__ src_out, float* __restrict__ tnew_out) {
  constexpr int SLOTS = 64 * H;
  __shared__ float s_t[RAYS_PER_BLOCK][SLOTS], s_a[RAYS_PER_BLOCK][SLOTS], s_b[RAYS_PER_BLOCK][SLOTS],
      s_w[RAYS_PER_BLOCK][SLOTS], s_c[RAYS_PER_BLOCK][SLOTS], s_new[RAYS_PER_BLOCK][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ray = blockIdx.x * RAYS_PER_BLOCK + w;
  const bool live = ray < R;
  const int NI = N - 1;
  float* T = s_t[w]; float* A = s_a[w]; float* B = s_b[w]; float* W = s_w[w]; float* C = s_c[w]; float* TN = s_new[w];
  const float tn = live ? t_near[ray] : 0.f, tf = live ? t_far[ray] : 0.f;

  // load samples; A = sdf
#pragma unroll
  for (int h = 0; h < H; ++h) {
    int i = lane + 64 * h;
    T[i] = (live && i < N) ? t_in[(long long)ray * N + i] : 0.f;
    A[i] = (live && i < N) ? sdf_in[(long long)ray * N + i] : 0.f;
  }
  __syncthreads();
  // cos1 of every interval -> B
#pragma unroll
  for (int h = 0; h < H; ++h) {
    int i = lane + 64 * h;
    float v = 0.f;
    if (i < NI) v = (A[i + 1] - A[i]) / (T[i + 1] - T[i] + 1e-5f);
    B[i] = v;
  }
  __syncthreads();
  // alpha -> W ; q = 1 - alpha -> C (scan input)
#pragma unroll
  for (int h = 0; h < H; ++h) {
    int i = lane + 64 * h;
    float alpha = 0.f;
    if (i < NI) {
      float d0 = A[i], d1 = A[i + 1], t0 = T[i], t1 = T[i + 1];
      float sdfm = (d0 + d1) * 0.5f;
      float cos1 = B[i];
      float cos0 = (i == 0) ? 1.0f : B[i - 1];
      float cv = fminf(cos0, cos1);
      cv = fminf(fmaxf(cv, -1e3f), 0.f);
      float dist = t1 - t0;
      float hh = cv * dist * 0.5f;
      float c0 = ndjir_sigmoidf((sdfm - hh) * gain);
      float c1 = ndjir_sigmoidf((sdfm + hh) * gain);
      alpha = (c0 - c1 + 1e-5f) / (c0 + 1e-5f);
      alpha = fminf(fmaxf(alpha, 0.f), 1.f);
    }
    W[i] = alpha;
    C[i] = (i < NI) ? 1.f - alpha : 1.f;
  }
  __syncthreads();
  // inclusive cumprod of q, Kogge-Stone
  for (int d = 1; d < SLOTS; d <<= 1) {
    float v[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { int i = lane + 64 * h; v[h] = (i >= d) ? C[i - d] * C[i] : C[i]; }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) C[lane + 64 * h] = v[h];
    __syncthreads();
  }
  // weights = alpha * exclusive cumprod -> W
#pragma unroll
  for (int h = 0; h < H; ++h) {
    int i = lane + 64 * h;
    float ex = (i == 0) ? 1.f : C[i - 1];
    W[i] = (i < NI) ? W[i] * ex : 0.f;
  }
  __syncthreads();
  // lane sums in the order of ndjir_math.h, then the xor butterfly over the 64 lanes
  float s = W[lane] + W[lane + 64];
  if (H == 4) s = s + (W[lane + 128] + W[lane + 192]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s = s + __shfl_xor(s, m);
#pragma unroll
  for (int h = 0; h < H; ++h) { int i = lane + 64 * h; W[i] = W[i] / s; C[i] = W[i]; }
  __syncthreads();
  // inclusive cumsum, Kogge-Stone
  for (int d = 1; d < SLOTS; d <<= 1) {
    float v[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { int i = lane + 64 * h; v[h] = (i >= d) ? C[i - d] + C[i] : C[i]; }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < H; ++h) C[lane + 64 * h] = v[h];
    __syncthreads();
  }
  // inverse transform sampling: lane m < M
  if (lane < M) {
    float u = (float)lane / udenom;
    int idx = 0;
    for (int i = 0; i < NI; ++i) idx += (C[i] < u) ? 1 : 0;
    float lower = (idx == 0) ? 0.f : C[idx - 1];
    int gi = idx < NI - 1 ? idx : NI - 1;
    float ratio = (u - lower) / W[gi];
    float step = (idx < N - 1) ? (T[idx + 1] - T[idx]) : (tf - T[N - 1]);
    float tv = T[idx] + step * ratio;
    tv = fmaxf(fminf(tv, tf), tn);
    TN[lane] = tv;
    if (live) {
      idx_out[(long long)ray * M + lane] = idx;
      if (tnew_out) tnew_out[(long long)ray * M + lane] = tv;
    }
  }
  __syncthreads();
  // merge the two sorted lists by rank
  if (live) {
    float* out = t_out + (long long)ray * (N + M);
    int* src = src_out ? src_out + (long long)ray * (N + M) : nullptr;   // merged position -> source slot
#pragma unroll
    for (int h = 0; h < H; ++h) {
      int i = lane + 64 * h;
      if (i < N) {
        float v = T[i];
        int r = i;
        for (int m = 0; m < M; ++m) r += (TN[m] < v) ? 1 : 0;
        out[r] = v;
        if (src) src[r] = i;
      }
    }
    if (lane < M) {
      float v = TN[lane];
      int r = lane;
      for (int i = 0; i < N; ++i) r += (T[i] <= v) ? 1 : 0;
      out[r] = v;
      if (src) src[r] = N + lane;
    }
  }
}

int launch_importance_round(int R, int N, int M, float gain, const float* t, const float* sdf, const float* t_near,
                            const float* t_far, float* t_out, int* idx_out, int* src_out, float* tnew_out,
                            hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (N < 2 || M < 1 || M > 32 || N + M > NDJIR_SAMPLER_SLOTS) return NDJIR_ERR_UNSUPPORTED;
  float udenom = (float)(M - 1 + 1.0 / M);
  int blocks = (R + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
  if (N + M <= 128)
    hipLaunchKernelGGL(k_importance_round<2>, dim3(blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream, R, N, M, gain, udenom, t,
                       sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out);
  else
    hipLaunchKernelGGL(k_importance_round<4>, dim3(blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream, R, N, M, gain, udenom, t,
                       sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out);
  return ndjir_check_launch();
}

}  // namespace ndjir

extern "C" int ndjir_sampler_importance_round(int R, int N, int M, float gain, const float* t, const float* sdf,
                                              const float* t_near, const float* t_far, float* t_out, int* idx_out,
                                              int* src_out, float* tnew_out, hipStream_t stream) {
  if (R > 0 && (!t || !sdf || !t_near || !t_far || !t_out || !idx_out)) return NDJIR_ERR_ARG;
  return ndjir::launch_importance_round(R, N, M, gain, t, sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out, stream);
}
