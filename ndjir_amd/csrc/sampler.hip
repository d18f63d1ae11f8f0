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
// Optional fused glue of SamplePoints.sample_importance_dists (python/sampler.py:187-193, 236-240):
//   sdf_new / src_prev / n_prev: the SDF of the current samples is gathered here from the previous round's values
//     (sdf_in, n_prev of them) and the values at the samples that round added (sdf_new), by its source map --
//     instead of cat + gather launches; sdf_out (R, N) receives the gathered array for the next round;
//   camloc / raydir / xnew_out: the points x = c + t d of the M new samples (what the next SDF evaluation needs).
struct RoundGlue {
  const float* sdf_new;     // (R, N - n_prev) or null
  const int* src_prev;      // (R, N): merged position -> slot in [sdf_in | sdf_new]
  int n_prev;
  float* sdf_out;           // (R, N) or null
  const float* camloc;      // (B, 3) or null
  const float* raydir;      // (R, 3)
  int rays_per_batch;
  float* xnew_out;          // (R, M, 3)
};

template <int H>
__global__ void __launch_bounds__(64 * RAYS_PER_BLOCK) k_importance_round(
    int R, int N, int M, float gain, float udenom, const float* __restrict__ t_in, const float* __restrict__ sdf_in,
    const float* __restrict__ t_near, const float* __restrict__ t_far, float* __restrict__ t_out, int* __restrict__ idx_out,
    int* __restrict__ src_out, float* __restrict__ tnew_out, RoundGlue glue) {
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
    float a = 0.f;
    if (live && i < N) {
      if (glue.sdf_new) {
        const int sl = glue.src_prev[(long long)ray * N + i];
        a = sl < glue.n_prev ? sdf_in[(long long)ray * glue.n_prev + sl] : glue.sdf_new[(long long)ray * (N - glue.n_prev) + (sl - glue.n_prev)];
        if (glue.sdf_out) glue.sdf_out[(long long)ray * N + i] = a;
      } else {
        a = sdf_in[(long long)ray * N + i];
      }
    }
    A[i] = a;
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
      if (glue.xnew_out) {
        const float* c = glue.camloc + (long long)(ray / glue.rays_per_batch) * 3;
        const float* d = glue.raydir + (long long)ray * 3;
        float* x = glue.xnew_out + ((long long)ray * M + lane) * 3;
        x[0] = c[0] + tv * d[0]; x[1] = c[1] + tv * d[1]; x[2] = c[2] + tv * d[2];
      }
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
                            const float* t_far, float* t_out, int* idx_out, int* src_out, float* tnew_out, const RoundGlue& glue,
                            hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (N < 2 || M < 1 || M > 32 || N + M > NDJIR_SAMPLER_SLOTS) return NDJIR_ERR_UNSUPPORTED;
  float udenom = (float)(M - 1 + 1.0 / M);
  int blocks = (R + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;
  if (N + M <= 128)
    hipLaunchKernelGGL(k_importance_round<2>, dim3(blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream, R, N, M, gain, udenom, t,
                       sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out, glue);
  else
    hipLaunchKernelGGL(k_importance_round<4>, dim3(blocks), dim3(64 * RAYS_PER_BLOCK), 0, stream, R, N, M, gain, udenom, t,
                       sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out, glue);
  return ndjir_check_launch();
}

// ---- the rest of SamplePoints as two launches ---------------------------------------------------------------------------
// k_sampler_begin (python/sampler.py:71-165, 265-273): mask = n_hits > 1, stratified distances
//   t_i = t_near + (t_far - t_near) / N0 (i + u_i) and their points x_i = c + t_i d.
__global__ void __launch_bounds__(256) k_sampler_begin(long long R, int N0, int rays_per_batch, const float* __restrict__ camloc,
                                                       const float* __restrict__ raydir, const float* __restrict__ t_near,
                                                       const float* __restrict__ t_far, const float* __restrict__ n_hits,
                                                       const float* __restrict__ u, float* __restrict__ mask,
                                                       float* __restrict__ t0, float* __restrict__ x0) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * N0) return;
  const long long ray = e / N0;
  const int i = (int)(e - ray * N0);
  const float tn = t_near[ray], tf = t_far[ray];
  const float step = (tf - tn) / (float)N0;
  const float t = tn + step * ((float)i + u[e]);
  t0[e] = t;
  const float* c = camloc + (ray / rays_per_batch) * 3;
  const float* d = raydir + ray * 3;
  x0[e * 3] = c[0] + t * d[0]; x0[e * 3 + 1] = c[1] + t * d[1]; x0[e * 3 + 2] = c[2] + t * d[2];
  if (i == 0 && mask) mask[ray] = n_hits ? (n_hits[ray] > 1.f ? 1.f : 0.f) : 1.f;
}

// k_sampler_finish (python/sampler.py:275-299): x_fg = c + t d, t_fg = [t, t_far]; background (:244-254, 281-290):
//   t_base = t_far mask + (|c| - r)(1 - mask), t_bg = sort(t_base / u_bg), x_bg = inverted-sphere coordinates of the
//   first Nb of them.  One wave per ray; the (Nb + 1 <= 64)-value sort is a rank count.
__global__ void __launch_bounds__(256) k_sampler_finish(long long R, int N, int Nb, int rays_per_batch, float radius,
                                                        const float* __restrict__ camloc, const float* __restrict__ raydir,
                                                        const float* __restrict__ t, const float* __restrict__ t_far,
                                                        const float* __restrict__ mask, const float* __restrict__ u_bg,
                                                        float* __restrict__ x_fg, float* __restrict__ t_fg,
                                                        float* __restrict__ x_bg, float* __restrict__ t_bg) {
  __shared__ float s_v[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * 4 + w;
  if (ray >= R) return;
  const float* c = camloc + (ray / rays_per_batch) * 3;
  const float* d = raydir + ray * 3;
  const float cx = c[0], cy = c[1], cz = c[2], dx = d[0], dy = d[1], dz = d[2];
  const float tf = t_far[ray];
  for (int i = lane; i <= N; i += 64) {
    const float tv = i < N ? t[ray * N + i] : tf;
    t_fg[ray * (N + 1) + i] = tv;
    if (i < N) {
      float* x = x_fg + (ray * N + i) * 3;
      x[0] = cx + tv * dx; x[1] = cy + tv * dy; x[2] = cz + tv * dz;
    }
  }
  if (!x_bg) return;
  const float m = mask[ray];
  const float cn = sqrtf(cx * cx + cy * cy + cz * cz);
  const float t_base = tf * m + (cn - radius) * (1.f - m);
  float v = 0.f;
  if (lane <= Nb) { v = t_base / u_bg[ray * (Nb + 1) + lane]; s_v[w][lane] = v; }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane <= Nb) {
    int rank = 0;
    for (int j = 0; j <= Nb; ++j) { const float o = s_v[w][j]; rank += (o < v || (o == v && j < lane)) ? 1 : 0; }
    t_bg[ray * (Nb + 1) + rank] = v;
    if (rank < Nb) {
      const float px = cx + v * dx, py = cy + v * dy, pz = cz + v * dz;
      const float dist = sqrtf(px * px + py * py + pz * pz) + 1e-6f;
      float* x = x_bg + (ray * Nb + rank) * 4;
      x[0] = px / dist; x[1] = py / dist; x[2] = pz / dist; x[3] = 1.0f / dist;
    }
  }
}

// diagnostics: the shared exp / sigmoid definitions of include/ndjir_math.h (what the sampler's bin decisions are built on)
__global__ void __launch_bounds__(256) k_math_expf(int n, float* __restrict__ y, const float* __restrict__ x, int sigmoid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = sigmoid ? ndjir_sigmoidf(x[i]) : ndjir_expf(x[i]);
}


int launch_math_expf(int n, float* y, const float* x, int sigmoid, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  hipLaunchKernelGGL(k_math_expf, dim3((n + 255) / 256), dim3(256), 0, stream, n, y, x, sigmoid);
  return ndjir_check_launch();
}

}  // namespace ndjir

extern "C" int ndjir_sampler_importance_round(int R, int N, int M, float gain, const float* t, const float* sdf,
                                              const float* t_near, const float* t_far, float* t_out, int* idx_out,
                                              int* src_out, float* tnew_out, hipStream_t stream) {
  if (R > 0 && (!t || !sdf || !t_near || !t_far || !t_out || !idx_out)) return NDJIR_ERR_ARG;
  return ndjir::launch_importance_round(R, N, M, gain, t, sdf, t_near, t_far, t_out, idx_out, src_out, tnew_out, ndjir::RoundGlue{},
                                        stream);
}

// The same round with the surrounding glue fused (see RoundGlue): sdf = gather([sdf_prev (R, n_prev) | sdf_new (R, N - n_prev)],
// src_prev) when sdf_new is given (sdf_out receives it), else sdf_prev is the (R, N) array itself; xnew_out (R, M, 3) = camloc +
// t_new raydir when camloc is given (camloc (B, 3), rays_per_batch = R / B).
extern "C" int ndjir_sampler_round_fused(int R, int N, int M, float gain, const float* t, const float* sdf_prev, int n_prev,
                                         const float* sdf_new, const int* src_prev, float* sdf_out, const float* t_near,
                                         const float* t_far, const float* camloc, const float* raydir, int rays_per_batch,
                                         float* t_out, int* idx_out, int* src_out, float* tnew_out, float* xnew_out,
                                         hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!t || !sdf_prev || !t_near || !t_far || !t_out || !idx_out) return NDJIR_ERR_ARG;
  if (sdf_new && (!src_prev || n_prev < 1 || n_prev >= N)) return NDJIR_ERR_ARG;
  if (xnew_out && (!camloc || !raydir || rays_per_batch < 1)) return NDJIR_ERR_ARG;
  ndjir::RoundGlue g{};
  g.sdf_new = sdf_new; g.src_prev = src_prev; g.n_prev = sdf_new ? n_prev : N; g.sdf_out = sdf_out;
  g.camloc = xnew_out ? camloc : nullptr; g.raydir = raydir; g.rays_per_batch = rays_per_batch; g.xnew_out = xnew_out;
  return ndjir::launch_importance_round(R, N, M, gain, t, sdf_prev, t_near, t_far, t_out, idx_out, src_out, tnew_out, g, stream);
}

extern "C" int ndjir_sampler_begin(long long R, int N0, int rays_per_batch, const float* camloc, const float* raydir,
                                   const float* t_near, const float* t_far, const float* n_hits, const float* stratified_sample,
                                   float* mask, float* t0, float* x0, hipStream_t stream) {
  if (R <= 0 || N0 <= 0) return NDJIR_OK;
  if (!camloc || !raydir || !t_near || !t_far || !stratified_sample || !t0 || !x0 || rays_per_batch < 1) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(ndjir::k_sampler_begin, dim3((unsigned)((R * N0 + 255) / 256)), dim3(256), 0, stream, R, N0, rays_per_batch,
                     camloc, raydir, t_near, t_far, n_hits, stratified_sample, mask, t0, x0);
  return ndjir::ndjir_check_launch();
}

extern "C" int ndjir_sampler_finish(long long R, int N, int Nb, int rays_per_batch, float radius, const float* camloc,
                                    const float* raydir, const float* t, const float* t_far, const float* mask,
                                    const float* background_sample, float* x_fg, float* t_fg, float* x_bg, float* t_bg,
                                    hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!camloc || !raydir || !t || !t_far || !x_fg || !t_fg || rays_per_batch < 1 || N < 1) return NDJIR_ERR_ARG;
  if (x_bg && (!mask || !background_sample || !t_bg || Nb < 1)) return NDJIR_ERR_ARG;
  if (x_bg && Nb + 1 > 64) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ndjir::k_sampler_finish, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, stream, R, N, Nb, rays_per_batch, radius,
                     camloc, raydir, t, t_far, mask, background_sample, x_fg, t_fg, x_bg, t_bg);
  return ndjir::ndjir_check_launch();
}
