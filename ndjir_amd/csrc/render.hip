// render.hip -- volume-rendering stage of pb_render as fused kernels (forward + hand-derived backward).
//
// Reference graph: python/renderer.py:55-67 (foreground alpha from the SDF, its gradient and the ray
// direction), :79-87 (transmittance = exclusive cumprod, weights) and the `VR` integrals (:84-87, used
// nine times).  The reference builds these from ~50 nnabla elementwise / reduction functions (and as
// many again in backward); the exclusive cumprod's stock backward additionally synchronises the host.
// Here: one launch each for {alpha, transmittance, weights}, its backward, an integral and its
// backward.  One wave per ray; the 160-sample scan is a sequential pass of lane 0 through LDS (a few
// hundred cycles -- not worth a parallel scan, and it keeps the product order of the reference).
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include "common.h"

namespace ndjir {

constexpr int RD_SLOTS = 256;   // samples per ray (foreground + background) handled by one wave

__device__ __forceinline__ float rd_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

struct AlphaTerms {
  float a;        // clamp(q, 0, 1)
  float q, c0, c1, s0, s1, ic, tc, delta;
};

// renderer.py:57-67 for one foreground sample
__device__ __forceinline__ AlphaTerms alpha_terms(float sdf, float nx, float ny, float nz, float dx, float dy, float dz,
                                                  float t0, float t1, float gain, float car) {
  AlphaTerms o;
  o.tc = dx * nx + dy * ny + dz * nz;
  o.ic = -(fmaxf(-o.tc * 0.5f + 0.5f, 0.f) * (1.f - car) + fmaxf(-o.tc, 0.f) * car);
  o.delta = t1 - t0;
  o.s1 = sdf + o.ic * o.delta * 0.5f;
  o.s0 = sdf - o.ic * o.delta * 0.5f;
  o.c0 = rd_sigmoid(gain * o.s0);
  o.c1 = rd_sigmoid(gain * o.s1);
  o.q = (o.c0 - o.c1 + 1e-5f) / (o.c0 + 1e-5f);
  o.a = fminf(fmaxf(o.q, 0.f), 1.f);
  return o;
}

// grid: one 64-thread workgroup per ray.
__global__ void __launch_bounds__(64) k_alpha_weights(int N, int Nb, const float* __restrict__ sdf, const float* __restrict__ n,
                                                      const float* __restrict__ raydir, const float* __restrict__ t,
                                                      const float* __restrict__ gain_p, const float* __restrict__ car_p,
                                                      const float* __restrict__ mask, const float* __restrict__ alpha_bg,
                                                      float* __restrict__ alpha_fg, float* __restrict__ trans,
                                                      float* __restrict__ weights) {
  __shared__ float X[RD_SLOTS], A[RD_SLOTS], T[RD_SLOTS];
  const long long r = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = N + Nb;
  const float gain = gain_p[0], car = car_p[0], m = mask[r];
  const float dx = raydir[r * 3], dy = raydir[r * 3 + 1], dz = raydir[r * 3 + 2];
  for (int i = lane; i < S; i += 64) {
    float a_all;
    if (i < N) {
      const long long p = r * N + i;
      AlphaTerms o = alpha_terms(sdf[p], n[p * 3], n[p * 3 + 1], n[p * 3 + 2], dx, dy, dz, t[r * (N + 1) + i],
                                 t[r * (N + 1) + i + 1], gain, car);
      alpha_fg[p] = o.a;
      a_all = o.a * m;
    } else {
      a_all = alpha_bg[r * Nb + (i - N)];
    }
    A[i] = a_all;
    X[i] = 1.f - a_all;
  }
  __syncthreads();
  if (lane == 0) {
    float prod = 1.f;
    for (int i = 0; i < S; ++i) { T[i] = prod; prod *= X[i]; }
  }
  __syncthreads();
  for (int i = lane; i < S; i += 64) {
    trans[r * S + i] = T[i];
    weights[r * S + i] = A[i] * T[i];
  }
}

// backward: g_alpha_fg / g_trans / g_weights may be null (treated as zero)
__global__ void __launch_bounds__(64) k_alpha_weights_bwd(int N, int Nb, const float* __restrict__ sdf, const float* __restrict__ n,
                                                          const float* __restrict__ raydir, const float* __restrict__ t,
                                                          const float* __restrict__ gain_p, const float* __restrict__ car_p,
                                                          const float* __restrict__ mask, const float* __restrict__ alpha_bg,
                                                          const float* __restrict__ trans, const float* __restrict__ g_alpha_fg,
                                                          const float* __restrict__ g_trans, const float* __restrict__ g_weights,
                                                          float* __restrict__ g_sdf, float* __restrict__ g_n,
                                                          float* __restrict__ g_gain_ray, float* __restrict__ g_alpha_bg) {
  __shared__ float X[RD_SLOTS], H[RD_SLOTS], Sx[RD_SLOTS];
  const long long r = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = N + Nb;
  const float gain = gain_p[0], car = car_p[0], m = mask[r];
  const float dx = raydir[r * 3], dy = raydir[r * 3 + 1], dz = raydir[r * 3 + 2];
  // pass 1: hT_i = gT_i + gw_i A_i and x_i = 1 - A_i
  for (int i = lane; i < S; i += 64) {
    float a_all;
    if (i < N) {
      const long long p = r * N + i;
      AlphaTerms o = alpha_terms(sdf[p], n[p * 3], n[p * 3 + 1], n[p * 3 + 2], dx, dy, dz, t[r * (N + 1) + i],
                                 t[r * (N + 1) + i + 1], gain, car);
      a_all = o.a * m;
    } else {
      a_all = alpha_bg[r * Nb + (i - N)];
    }
    const float gT = g_trans ? g_trans[r * S + i] : 0.f;
    const float gw = g_weights ? g_weights[r * S + i] : 0.f;
    H[i] = gT + gw * a_all;
    X[i] = 1.f - a_all;
  }
  __syncthreads();
  // S_j = sum_{i>j} hT_i prod_{j<k<i} x_k  (no division: exact also where x_k = 0)
  if (lane == 0) {
    float s = 0.f;
    for (int j = S - 1; j >= 0; --j) { Sx[j] = s; s = H[j] + X[j] * s; }
  }
  __syncthreads();
  float gg = 0.f;
  for (int i = lane; i < S; i += 64) {
    const float Ti = trans[r * S + i];
    const float gw = g_weights ? g_weights[r * S + i] : 0.f;
    const float gA = gw * Ti - Ti * Sx[i];          // dL/dA_i (A_i enters w_i and x_i = 1 - A_i)
    if (i >= N) {
      if (g_alpha_bg) g_alpha_bg[r * Nb + (i - N)] = gA;
      continue;
    }
    const long long p = r * N + i;
    const float nx = n[p * 3], ny = n[p * 3 + 1], nz = n[p * 3 + 2];
    AlphaTerms o = alpha_terms(sdf[p], nx, ny, nz, dx, dy, dz, t[r * (N + 1) + i], t[r * (N + 1) + i + 1], gain, car);
    float ga = gA * m + (g_alpha_fg ? g_alpha_fg[p] : 0.f);
    const float gq = (o.q >= 0.f && o.q <= 1.f) ? ga : 0.f;        // clamp backward (bounds inclusive)
    const float v = o.c0 + 1e-5f;
    const float gc0 = gq * (o.c1 / (v * v));
    const float gc1 = -gq / v;
    const float d0 = o.c0 * (1.f - o.c0), d1 = o.c1 * (1.f - o.c1);   // sigmoid'
    const float gs0 = gc0 * d0 * gain, gs1 = gc1 * d1 * gain;
    gg += gc0 * d0 * o.s0 + gc1 * d1 * o.s1;
    g_sdf[p] = gs0 + gs1;
    const float gic = (gs1 - gs0) * o.delta * 0.5f;
    // d ic / d tc: relu'(x) = [x > 0]
    const float dic = ((-o.tc * 0.5f + 0.5f > 0.f) ? 0.5f * (1.f - car) : 0.f) + ((-o.tc > 0.f) ? car : 0.f);
    const float gtc = gic * dic;
    g_n[p * 3] = gtc * dx;
    g_n[p * 3 + 1] = gtc * dy;
    g_n[p * 3 + 2] = gtc * dz;
  }
  // per-ray partial of dL/d gain
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gg += __shfl_xor(gg, off);
  if (lane == 0) g_gain_ray[r] = gg;
}

// ---- VR integral: out[r][c] = sum_i w[r][i] x[r][i][c] -----------------------------------------------
// threads = (channel phase tx < TX, sample phase ty); TX = pow2 >= min(C, 256)
__device__ __forceinline__ int rd_pow2(int v) {
  int t = 1;
  while (t < v && t < 256) t <<= 1;
  return t;
}

__global__ void __launch_bounds__(256) k_integrate(int S, int C, const float* __restrict__ w, int ldw,
                                                   const float* __restrict__ x, float* __restrict__ out) {
  __shared__ float red[256];
  const long long r = blockIdx.x;
  const int TX = rd_pow2(C), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const float* wr = w + r * ldw;
  const float* xr = x + r * (long long)S * C;
  for (int c0 = 0; c0 < C; c0 += TX) {
    const int c = c0 + tx;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int i = ty; i < S; i += TY) acc += wr[i] * xr[(long long)i * C + c];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = TY / 2; s > 0; s >>= 1) {
      if (ty < s) red[threadIdx.x] += red[threadIdx.x + s * TX];
      __syncthreads();
    }
    if (ty == 0 && c < C) out[r * C + c] = red[tx];
    __syncthreads();
  }
}

// gx[r][i][c] = w[r][i] g[r][c];  gw[r][i] = sum_c x[r][i][c] g[r][c].  One workgroup per ray, a wave
// per sample (lanes over channels, shuffle reduction).  gx / gw may be null.
__global__ void __launch_bounds__(256) k_integrate_bwd(int S, int C, const float* __restrict__ w, int ldw,
                                                       const float* __restrict__ x, const float* __restrict__ g,
                                                       float* __restrict__ gx, float* __restrict__ gw, int ldgw) {
  const long long r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* wr = w + r * ldw;
  const float* gr = g + r * C;
  if (C <= 4) {
    // few channels: one thread per sample
    for (int i = threadIdx.x; i < S; i += 256) {
      const float wi = wr[i];
      float acc = 0.f;
      for (int c = 0; c < C; ++c) {
        const long long e = (r * S + i) * C + c;
        if (gw) acc += x[e] * gr[c];
        if (gx) gx[e] = wi * gr[c];
      }
      if (gw) gw[r * ldgw + i] = acc;
    }
    return;
  }
  for (int i = wave; i < S; i += 4) {
    const float wi = wr[i];
    const long long base = (r * S + i) * C;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float gc = gr[c];
      if (gw) acc += x[base + c] * gc;
      if (gx) gx[base + c] = wi * gc;
    }
    if (gw) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0) gw[r * ldgw + i] = acc;
    }
  }
}

}  // namespace ndjir

using namespace ndjir;

extern "C" int ndjir_render_alpha_weights(int R, int N, int Nb, const float* sdf, const float* n, const float* raydir,
                                          const float* t, const float* gain, const float* cos_anneal_ratio, const float* mask,
                                          const float* alpha_bg, float* alpha_fg, float* trans, float* weights,
                                          hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (N < 1 || Nb < 0 || N + Nb > RD_SLOTS) return NDJIR_ERR_UNSUPPORTED;
  if (!sdf || !n || !raydir || !t || !gain || !cos_anneal_ratio || !mask || (Nb > 0 && !alpha_bg) || !alpha_fg || !trans ||
      !weights)
    return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_alpha_weights, dim3(R), dim3(64), 0, stream, N, Nb, sdf, n, raydir, t, gain, cos_anneal_ratio, mask,
                     alpha_bg, alpha_fg, trans, weights);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_alpha_weights_backward(int R, int N, int Nb, const float* sdf, const float* n, const float* raydir,
                                                   const float* t, const float* gain, const float* cos_anneal_ratio,
                                                   const float* mask, const float* alpha_bg, const float* trans,
                                                   const float* g_alpha_fg, const float* g_trans, const float* g_weights,
                                                   float* g_sdf, float* g_n, float* g_gain_ray, float* g_alpha_bg,
                                                   hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (N < 1 || Nb < 0 || N + Nb > RD_SLOTS) return NDJIR_ERR_UNSUPPORTED;
  if (!sdf || !n || !raydir || !t || !gain || !cos_anneal_ratio || !mask || (Nb > 0 && !alpha_bg) || !trans || !g_sdf || !g_n ||
      !g_gain_ray)
    return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_alpha_weights_bwd, dim3(R), dim3(64), 0, stream, N, Nb, sdf, n, raydir, t, gain, cos_anneal_ratio, mask,
                     alpha_bg, trans, g_alpha_fg, g_trans, g_weights, g_sdf, g_n, g_gain_ray, g_alpha_bg);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_integrate(int R, int S, int C, const float* w, int ldw, const float* x, float* out,
                                      hipStream_t stream) {
  if (R <= 0 || C <= 0) return NDJIR_OK;
  if (S < 0 || ldw < S || !w || !x || !out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_integrate, dim3(R), dim3(256), 0, stream, S, C, w, ldw, x, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_integrate_backward(int R, int S, int C, const float* w, int ldw, const float* x, const float* g,
                                               float* gx, float* gw, int ldgw, hipStream_t stream) {
  if (R <= 0 || C <= 0 || S <= 0) return NDJIR_OK;
  if (ldw < S || !w || !x || !g || (gw && ldgw < S)) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_integrate_bwd, dim3(R), dim3(256), 0, stream, S, C, w, ldw, x, g, gx, gw, ldgw);
  return ndjir_check_launch();
}
