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

constexpr int RD_SLOTS = 512;   // samples per ray (foreground + background) handled by one wave (cfg5: 256 + 32)

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
                                                   const float* __restrict__ x, int ldx, float* __restrict__ out) {
  __shared__ float red[256];
  const long long r = blockIdx.x;
  const int TX = rd_pow2(C), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const float* wr = w + r * ldw;
  const float* xr = x + r * (long long)S * ldx;
  for (int c0 = 0; c0 < C; c0 += TX) {
    const int c = c0 + tx;
    float acc = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int i = ty; i < S; i += TY) acc += wr[i] * xr[(long long)i * ldx + c];
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
                                                       const float* __restrict__ x, int ldx, const float* __restrict__ g,
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
        if (gw) acc += x[(r * S + i) * ldx + c] * gr[c];
        if (gx) gx[e] = wi * gr[c];
      }
      if (gw) gw[r * ldgw + i] = acc;
    }
    return;
  }
  for (int i = wave; i < S; i += 4) {
    const float wi = wr[i];
    const long long base = (r * S + i) * C, xbase = (r * S + i) * ldx;
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float gc = gr[c];
      if (gw) acc += x[xbase + c] * gc;
      if (gx) gx[base + c] = wi * gc;
    }
    if (gw) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0) gw[r * ldgw + i] = acc;
    }
  }
}

// ---- several VR integrals of one ray's weights in one launch ----
// python/renderer.py:84-87 is called for the normal, the position, the feature, the material products and the background
// colour of every ray (:90-176); each call reads the same weights and -- backward -- adds to the same weight gradient.
// Segment k: x_k (R, S_k, C_k) with row stride ld_k against weights[:, off_k : off_k + S_k].
constexpr int IM_SEGS = 6;
struct IntSegs {
  const float* x[IM_SEGS];
  const float* g[IM_SEGS];      // backward: (R, C_k) or null (segment without a gradient)
  float* out[IM_SEGS];          // forward: (R, C_k); backward: gx (R, S_k, C_k) or null
  int ld[IM_SEGS], C[IM_SEGS], S[IM_SEGS], off[IM_SEGS];
  int n;
};

__global__ void __launch_bounds__(256) k_integrate_many(int S_all, const float* __restrict__ w, IntSegs sg) {
  __shared__ float red[256];
  const long long r = blockIdx.x;
  const float* wrow = w + r * S_all;
  for (int k = 0; k < sg.n; ++k) {
    const int C = sg.C[k], S = sg.S[k], ldx = sg.ld[k];
    int TX = rd_pow2(C);
    if (TX > 64) TX = 64;                          // wide segments: 64-channel chunks dealt to the workgroups (r, y)
    const int TY = 256 / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const float* wr = wrow + sg.off[k];
    const float* xr = sg.x[k] + r * (long long)S * ldx;
    for (int c0 = blockIdx.y * TX; c0 < C; c0 += gridDim.y * TX) {     // uniform per workgroup
      const int c = c0 + tx;
      float acc = 0.f;
      if (c < C) {
#pragma unroll 4
        for (int i = ty; i < S; i += TY) acc += wr[i] * xr[(long long)i * ldx + c];
      }
      red[threadIdx.x] = acc;
      __syncthreads();
      for (int s = TY / 2; s > 0; s >>= 1) {
        if (ty < s) red[threadIdx.x] += red[threadIdx.x + s * TX];
        __syncthreads();
      }
      if (ty == 0 && c < C) sg.out[k][r * C + c] = red[tx];
      __syncthreads();
    }
  }
}

// gx_k[r][i][c] = w[r][off_k + i] g_k[r][c];  gw[r][j] = sum over the segments covering j of sum_c x_k[r][j - off_k][c] g_k[r][c]
// Workgroup (r, y) owns the weight positions j = y, y + gridDim.y, ... of ray r (a wave per position, lanes over channels):
// the segments are visited in order, so each gw[r][j] is summed in a fixed order by one wave.
__global__ void __launch_bounds__(256) k_integrate_many_bwd(int S_all, const float* __restrict__ w, IntSegs sg,
                                                            float* __restrict__ gw) {
  const long long r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* wrow = w + r * S_all;
  for (int j = blockIdx.y * 4 + wave; j < S_all; j += gridDim.y * 4) {
    const float wj = wrow[j];
    float tot = 0.f;
    for (int k = 0; k < sg.n; ++k) {
      const float* g = sg.g[k];
      const int i = j - sg.off[k];
      if (!g || i < 0 || i >= sg.S[k]) continue;
      const int C = sg.C[k], S = sg.S[k], ldx = sg.ld[k];
      const float* gr = g + r * C;
      const float* x = sg.x[k] + (r * (long long)S + i) * ldx;
      float* gx = sg.out[k] ? sg.out[k] + (r * (long long)S + i) * C : nullptr;
      float acc = 0.f;
      if ((C & 3) == 0) {
        // wide segments (the 256 feature channels): 16 bytes per lane, x at whatever alignment its packed row has
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef f4 f4u __attribute__((aligned(4)));
        for (int c = 4 * lane; c < C; c += 256) {
          const f4 gc = *reinterpret_cast<const f4u*>(gr + c);
          if (gw) {
            const f4 xv = *reinterpret_cast<const f4u*>(x + c);
            acc += xv[0] * gc[0] + xv[1] * gc[1] + xv[2] * gc[2] + xv[3] * gc[3];
          }
          if (gx) *reinterpret_cast<f4u*>(gx + c) = gc * wj;
        }
      } else {
        for (int c = lane; c < C; c += 64) {
          const float gc = gr[c];
          if (gw) acc += x[c] * gc;
          if (gx) gx[c] = wj * gc;
        }
      }
      tot += acc;
    }
    if (gw) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
      if (lane == 0) gw[r * S_all + j] = tot;
    }
  }
}

static int int_segs(IntSegs& sg, int S_all, int nseg, const float* const* x, const int* ld, const int* C, const int* S, const int* off) {
  if (nseg < 1 || nseg > IM_SEGS || !x || !ld || !C || !S || !off) return NDJIR_ERR_ARG;
  sg.n = nseg;
  for (int k = 0; k < IM_SEGS; ++k) {
    sg.x[k] = k < nseg ? x[k] : nullptr;
    sg.g[k] = nullptr;
    sg.out[k] = nullptr;
    sg.ld[k] = k < nseg ? ld[k] : 0;
    sg.C[k] = k < nseg ? C[k] : 0;
    sg.S[k] = k < nseg ? S[k] : 0;
    sg.off[k] = k < nseg ? off[k] : 0;
    if (k < nseg && (!x[k] || C[k] < 1 || ld[k] < C[k] || S[k] < 0 || off[k] < 0 || off[k] + S[k] > S_all)) return NDJIR_ERR_ARG;
  }
  return NDJIR_OK;
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

extern "C" int ndjir_render_integrate(int R, int S, int C, const float* w, int ldw, const float* x, int ldx, float* out,
                                      hipStream_t stream) {
  if (R <= 0 || C <= 0) return NDJIR_OK;
  if (S < 0 || ldw < S || ldx < C || !w || !x || !out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_integrate, dim3(R), dim3(256), 0, stream, S, C, w, ldw, x, ldx, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_integrate_backward(int R, int S, int C, const float* w, int ldw, const float* x, int ldx,
                                               const float* g, float* gx, float* gw, int ldgw, hipStream_t stream) {
  if (R <= 0 || C <= 0 || S <= 0) return NDJIR_OK;
  if (ldw < S || ldx < C || !w || !x || !g || (gw && ldgw < S)) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_integrate_bwd, dim3(R), dim3(256), 0, stream, S, C, w, ldw, x, ldx, g, gx, gw, ldgw);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_integrate_many(int R, int S_all, const float* w, int nseg, const float* const* x, const int* ld,
                                           const int* C, const int* S, const int* off, float* const* out, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  IntSegs sg;
  if (!w || !out || int_segs(sg, S_all, nseg, x, ld, C, S, off) != NDJIR_OK) return NDJIR_ERR_ARG;
  for (int k = 0; k < nseg; ++k) {
    if (!out[k]) return NDJIR_ERR_ARG;
    sg.out[k] = out[k];
  }
  int cmax = 1;
  for (int k = 0; k < nseg; ++k) cmax = C[k] > cmax ? C[k] : cmax;
  int ny = (cmax + 63) / 64;
  if (ny > 8) ny = 8;
  hipLaunchKernelGGL(k_integrate_many, dim3(R, ny), dim3(256), 0, stream, S_all, w, sg);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_integrate_many_backward(int R, int S_all, const float* w, int nseg, const float* const* x, const int* ld,
                                                    const int* C, const int* S, const int* off, const float* const* g,
                                                    float* const* gx, float* gw, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  IntSegs sg;
  if (!w || !g || !gx || int_segs(sg, S_all, nseg, x, ld, C, S, off) != NDJIR_OK) return NDJIR_ERR_ARG;
  for (int k = 0; k < nseg; ++k) {
    sg.g[k] = g[k];
    sg.out[k] = gx[k];
  }
  // enough workgroups for a bandwidth-bound pass (the feature segment alone writes R x S x 256 floats): ~4096 in all
  int ny = (4096 + R - 1) / R;
  const int ny_max = (S_all + 3) / 4;
  if (ny > ny_max) ny = ny_max;
  if (ny < 1) ny = 1;
  hipLaunchKernelGGL(k_integrate_many_bwd, dim3(R, ny), dim3(256), 0, stream, S_all, w, sg, gw);
  return ndjir_check_launch();
}

// ---- direct-light integrals over the M sampled light directions of a ray ----------------------------
// python/renderer.py:117-118  env_pixel = mean_m( soft_vis * env * clamp(n.l, eps) )                     (diffuse)
// python/renderer.py:136-161 + python/specular_brdf.py:40-118 (filament model, importance sampling, no
// split sum):  spec_pixel_c = weight * mean_m( sBRDF_c * soft_vis * env * clamp(n.l, eps) )               (specular)
// with the BRDF algebra (half vector, four clamped dots, Smith-GGX visibility, Schlick Fresnel) done
// per (ray, light) in registers.  The reference spells each of these as ~50 nnabla functions over
// (B,R,M,*) tensors; here it is one launch each way.  Light and view directions carry no gradient
// (SampleDirections has none, python/sampler.py:391-392).  One workgroup of 128 threads per ray.
namespace ndjir {

constexpr int SH_THREADS = 128;

__device__ __forceinline__ float sh_block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1];
}

// out (R,C) ; soft_vis (R,M) ; env (R,M,C) ; light (R,M,3) ; normal (R,3)
__global__ void __launch_bounds__(SH_THREADS) k_diffuse_light(int M, int C, const float* __restrict__ normal,
                                                              const float* __restrict__ light, const float* __restrict__ soft_vis,
                                                              const float* __restrict__ env, float eps_dot,
                                                              float* __restrict__ out) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    const float c = fmaxf(nx * light[e * 3] + ny * light[e * 3 + 1] + nz * light[e * 3 + 2], eps_dot);
    const float t = soft_vis[e] * c;
    for (int k = 0; k < C; ++k) acc[k] += t * env[e * C + k];
  }
  for (int k = 0; k < C; ++k) {
    const float s = sh_block_sum(acc[k], red);
    if (threadIdx.x == 0) out[r * C + k] = s / (float)M;
  }
}

__global__ void __launch_bounds__(SH_THREADS) k_diffuse_light_bwd(int M, int C, const float* __restrict__ normal,
                                                                  const float* __restrict__ light, const float* __restrict__ soft_vis,
                                                                  const float* __restrict__ env, float eps_dot,
                                                                  const float* __restrict__ g, float* __restrict__ g_normal,
                                                                  float* __restrict__ g_soft_vis, float* __restrict__ g_env) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float inv = 1.f / (float)M;
  float gn[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    const float lx = light[e * 3], ly = light[e * 3 + 1], lz = light[e * 3 + 2];
    const float raw = nx * lx + ny * ly + nz * lz;
    const float c = fmaxf(raw, eps_dot);
    const float sv = soft_vis[e];
    float ge = 0.f;                       // sum_k g_k env_k
    for (int k = 0; k < C; ++k) {
      const float gk = g[r * C + k] * inv;
      ge += gk * env[e * C + k];
      g_env[e * C + k] = gk * sv * c;
    }
    g_soft_vis[e] = ge * c;
    const float gc = (raw >= eps_dot) ? ge * sv : 0.f;      // clamp(min) backward
    gn[0] += gc * lx; gn[1] += gc * ly; gn[2] += gc * lz;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gn[k], red);
    if (threadIdx.x == 0) g_normal[r * 3 + k] = s;
  }
}

struct SpecTerms {
  float nol, nov, noh, voh;            // clamped dots
  float rnol, rnov, rnoh;              // raw dots
  float hx, hy, hz;
  float mask, V1l, V1v, sl, sv_, a2;
};

__device__ __forceinline__ SpecTerms spec_terms(float nx, float ny, float nz, float vx, float vy, float vz, float lx, float ly,
                                                float lz, float rough, float eps_dot) {
  SpecTerms t;
  float ux = lx + vx, uy = ly + vy, uz = lz + vz;
  const float un = sqrtf(ux * ux + uy * uy + uz * uz);
  t.hx = ux / un; t.hy = uy / un; t.hz = uz / un;
  t.rnol = nx * lx + ny * ly + nz * lz;
  t.rnov = nx * vx + ny * vy + nz * vz;
  t.rnoh = nx * t.hx + ny * t.hy + nz * t.hz;
  const float rvoh = vx * t.hx + vy * t.hy + vz * t.hz;
  t.nol = fmaxf(t.rnol, eps_dot); t.nov = fmaxf(t.rnov, eps_dot); t.noh = fmaxf(t.rnoh, eps_dot); t.voh = fmaxf(rvoh, eps_dot);
  t.mask = (t.rnol > eps_dot && t.rnov > eps_dot && t.rnoh > eps_dot) ? 1.f : 0.f;
  t.a2 = rough * rough;
  t.sl = sqrtf(t.a2 + (1.f - t.a2) * t.nol * t.nol);
  t.sv_ = sqrtf(t.a2 + (1.f - t.a2) * t.nov * t.nov);
  t.V1l = 1.f / (t.nol + t.sl + 1e-6f);
  t.V1v = 1.f / (t.nov + t.sv_ + 1e-6f);
  return t;
}

// out (R,3); spec (R,3); rough (R); env (R,M,C) with C = 1 or 3
__global__ void __launch_bounds__(SH_THREADS) k_specular_light(int M, int C, const float* __restrict__ normal,
                                                               const float* __restrict__ view, const float* __restrict__ light,
                                                               const float* __restrict__ rough, const float* __restrict__ spec,
                                                               const float* __restrict__ soft_vis, const float* __restrict__ env,
                                                               float eps_dot, float weight, float* __restrict__ out) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = rough[r];
  float acc[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, light[e * 3], light[e * 3 + 1], light[e * 3 + 2], ro, eps_dot);
    const float V = t.V1l * t.V1v;
    const float f5 = powf(1.f - t.voh, 5.f);
    const float Kf = 4.f * t.voh / t.noh * t.mask;
    const float sv = soft_vis[e];
    for (int k = 0; k < 3; ++k) {
      const float sc = spec[r * 3 + k];
      const float Fs = sc + (1.f - sc) * f5;
      acc[k] += V * Fs * Kf * sv * env[e * C + (C == 1 ? 0 : k)] * t.nol;
    }
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(acc[k], red);
    if (threadIdx.x == 0) out[r * 3 + k] = weight * s / (float)M;
  }
}

__global__ void __launch_bounds__(SH_THREADS) k_specular_light_bwd(int M, int C, const float* __restrict__ normal,
                                                                   const float* __restrict__ view, const float* __restrict__ light,
                                                                   const float* __restrict__ rough, const float* __restrict__ spec,
                                                                   const float* __restrict__ soft_vis, const float* __restrict__ env,
                                                                   float eps_dot, float weight, const float* __restrict__ g,
                                                                   float* __restrict__ g_normal, float* __restrict__ g_rough,
                                                                   float* __restrict__ g_spec, float* __restrict__ g_soft_vis,
                                                                   float* __restrict__ g_env) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = rough[r];
  const float wM = weight / (float)M;
  float gk[3], sc[3];
  for (int k = 0; k < 3; ++k) { gk[k] = g[r * 3 + k] * wM; sc[k] = spec[r * 3 + k]; }
  float gn[3] = {0.f, 0.f, 0.f}, ga2 = 0.f, gsc[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    const float lx = light[e * 3], ly = light[e * 3 + 1], lz = light[e * 3 + 2];
    SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, lx, ly, lz, ro, eps_dot);
    const float V = t.V1l * t.V1v;
    const float omv = 1.f - t.voh;
    const float f4 = omv * omv * omv * omv, f5 = f4 * omv;
    const float Kf = 4.f * t.voh / t.noh * t.mask;
    const float sv = soft_vis[e];
    // out_k += sB_k * T_k,  sB_k = V Fs_k Kf,  T_k = sv env_k nol
    float dV = 0.f, dK = 0.f, dFsum = 0.f /* sum_k dFs_k (1 - sc_k) */, dnol = 0.f, dsv = 0.f;
    float denv[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
      const float ek = env[e * C + (C == 1 ? 0 : k)];
      const float Fs = sc[k] + (1.f - sc[k]) * f5;
      const float sB = V * Fs * Kf;
      const float dsB = gk[k] * sv * ek * t.nol;
      dsv += gk[k] * sB * ek * t.nol;
      denv[C == 1 ? 0 : k] += gk[k] * sB * sv * t.nol;
      dnol += gk[k] * sB * sv * ek;
      dV += dsB * Fs * Kf;
      dK += dsB * V * Fs;
      const float dFs = dsB * V * Kf;
      gsc[k] += dFs * (1.f - f5);
      dFsum += dFs * (1.f - sc[k]);
    }
    g_soft_vis[e] = dsv;
    for (int k = 0; k < C; ++k) g_env[e * C + k] = denv[k];
    // Kf = 4 voh / noh * mask ; Fs depends on voh -- voh only reaches h (no gradient) -> dropped
    const float dnoh = -dK * 4.f * t.voh / (t.noh * t.noh) * t.mask;
    (void)dFsum;
    // V = V1(nol) V1(nov);  V1(u) = 1 / (u + sqrt(a2 + (1 - a2) u^2) + eps)
    const float dV1l = dV * t.V1v, dV1v = dV * t.V1l;
    const float om = 1.f - t.a2;
    dnol += dV1l * (-t.V1l * t.V1l) * (1.f + om * t.nol / t.sl);
    const float dnov = dV1v * (-t.V1v * t.V1v) * (1.f + om * t.nov / t.sv_);
    ga2 += dV1l * (-t.V1l * t.V1l) * (1.f - t.nol * t.nol) / (2.f * t.sl) + dV1v * (-t.V1v * t.V1v) * (1.f - t.nov * t.nov) / (2.f * t.sv_);
    // clamped dots -> normal (clamp(min) passes the gradient where raw >= eps)
    const float cl = (t.rnol >= eps_dot) ? dnol : 0.f, cv = (t.rnov >= eps_dot) ? dnov : 0.f, ch = (t.rnoh >= eps_dot) ? dnoh : 0.f;
    gn[0] += cl * lx + cv * vx + ch * t.hx;
    gn[1] += cl * ly + cv * vy + ch * t.hy;
    gn[2] += cl * lz + cv * vz + ch * t.hz;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gn[k], red);
    if (threadIdx.x == 0) g_normal[r * 3 + k] = s;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gsc[k], red);
    if (threadIdx.x == 0) g_spec[r * 3 + k] = s;
  }
  {
    const float s = sh_block_sum(ga2, red);
    if (threadIdx.x == 0) g_rough[r] = s * 2.f * ro;      // a2 = roughness^2
  }
}

// ---- every other branch of the specular BRDF (python/specular_brdf.py:40-199, python/renderer.py:141-161) -------------------
// MODEL 0 filament (:40-118) | 1 ue4 (:121-191);  SAMPLING 0 importance | 1 uniform;  SPLIT: use_split_sum
// (renderer.py:153-156:  mean_m(soft_vis env) * mean_m(sBRDF cos)  instead of  mean_m(sBRDF soft_vis env cos)).
//   sBRDF_k = W(nol, nov; rough) K(noh, nol, nov, voh; rough) Fs_k(voh) mask
//   filament: W = V1(nol) V1(nov), V1(u) = 1 / (u + sqrt(a2 + (1 - a2) u^2) + eps), a2 = rough^2, Fs = sc + (1 - sc)(1 - voh)^5,
//             importance K = 4 voh / noh,  uniform K = pi D,  D = a2 / (pi (noh^2 (a2 - 1) + 1)^2 + eps)
//   ue4:      W = G1(nol) G1(nov), G1(u) = u / (u (1 - k) + k + eps), k = (rough + 1)^2 / 8, Fs = sc + (1 - sc) 2^((-5.55473 voh - 6.98316) voh),
//             importance K = voh / (noh nov),  uniform K = pi D / (4 nov nol) with a2 = rough^4
// One workgroup per ray, lanes over the M light directions; the half vector carries no gradient (view and light directions are
// inputs without one), so voh is a constant of the backward pass.  The default branch (filament, importance, no split sum) keeps
// its own kernels above and inside k_direct_light.
struct SpecG {
  float S;                 // W K mask
  float W, K;
  float dW_dnol, dW_dnov, dW_dr;          // partials of W (r = roughness)
  float dK_dnoh, dK_dnol, dK_dnov, dK_dr; // partials of K
  float f;                 // Fresnel weight: Fs_k = sc_k + (1 - sc_k) f
};

template <int MODEL, int SAMPLING>
__device__ __forceinline__ SpecG spec_general(const SpecTerms& t, float r) {
  constexpr float PI = 3.14159265358979323846f, eps = 1e-6f;
  SpecG o;
  float a2, da2_dr;
  if (MODEL == 0) { a2 = r * r; da2_dr = 2.f * r; } else { a2 = r * r * r * r; da2_dr = 4.f * r * r * r; }
  if (MODEL == 0) {
    // (t.V1l, t.V1v, t.sl, t.sv_ are the filament terms with a2 = r^2)
    const float om = 1.f - a2;
    o.W = t.V1l * t.V1v;
    const float dV1l_du = -t.V1l * t.V1l * (1.f + om * t.nol / t.sl), dV1v_du = -t.V1v * t.V1v * (1.f + om * t.nov / t.sv_);
    const float dV1l_da = -t.V1l * t.V1l * (1.f - t.nol * t.nol) / (2.f * t.sl), dV1v_da = -t.V1v * t.V1v * (1.f - t.nov * t.nov) / (2.f * t.sv_);
    o.dW_dnol = dV1l_du * t.V1v;
    o.dW_dnov = t.V1l * dV1v_du;
    o.dW_dr = (dV1l_da * t.V1v + t.V1l * dV1v_da) * da2_dr;
    const float omv = 1.f - t.voh;
    o.f = omv * omv * omv * omv * omv;
  } else {
    const float k = (r + 1.f) * (r + 1.f) * 0.125f, dk_dr = (r + 1.f) * 0.25f;
    const float dl = t.nol * (1.f - k) + k + eps, dv = t.nov * (1.f - k) + k + eps;
    const float G1l = t.nol / dl, G1v = t.nov / dv;
    o.W = G1l * G1v;
    o.dW_dnol = (k + eps) / (dl * dl) * G1v;
    o.dW_dnov = G1l * (k + eps) / (dv * dv);
    const float dG1l_dk = -t.nol * (1.f - t.nol) / (dl * dl), dG1v_dk = -t.nov * (1.f - t.nov) / (dv * dv);
    o.dW_dr = (dG1l_dk * G1v + G1l * dG1v_dk) * dk_dr;
    o.f = exp2f((-5.55473f * t.voh - 6.98316f) * t.voh);
  }
  o.dK_dnoh = o.dK_dnol = o.dK_dnov = o.dK_dr = 0.f;
  if (SAMPLING == 0) {
    if (MODEL == 0) { o.K = 4.f * t.voh / t.noh; o.dK_dnoh = -o.K / t.noh; }
    else { o.K = t.voh / (t.noh * t.nov); o.dK_dnoh = -o.K / t.noh; o.dK_dnov = -o.K / t.nov; }
  } else {
    const float q = t.noh * t.noh * (a2 - 1.f) + 1.f;
    const float den = PI * q * q + eps;
    const float D = a2 / den;
    const float dD_dnoh = -a2 * PI * 2.f * q * (2.f * t.noh * (a2 - 1.f)) / (den * den);
    const float dD_da2 = 1.f / den - a2 * PI * 2.f * q * t.noh * t.noh / (den * den);
    if (MODEL == 0) { o.K = PI * D; o.dK_dnoh = PI * dD_dnoh; o.dK_dr = PI * dD_da2 * da2_dr; }
    else {
      const float c = PI / (4.f * t.nov * t.nol);
      o.K = c * D; o.dK_dnoh = c * dD_dnoh; o.dK_dr = c * dD_da2 * da2_dr;
      o.dK_dnov = -o.K / t.nov; o.dK_dnol = -o.K / t.nol;
    }
  }
  o.S = o.W * o.K * t.mask;
  return o;
}

template <int MODEL, int SAMPLING, bool SPLIT>
__global__ void __launch_bounds__(SH_THREADS) k_specular_light_g(int M, int C, const float* __restrict__ normal,
                                                                 const float* __restrict__ view, const float* __restrict__ light,
                                                                 const float* __restrict__ rough, const float* __restrict__ spec,
                                                                 const float* __restrict__ soft_vis, const float* __restrict__ env,
                                                                 float eps_dot, float weight, float* __restrict__ out) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = rough[r];
  float acc[3] = {0.f, 0.f, 0.f}, accA[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    const SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, light[e * 3], light[e * 3 + 1], light[e * 3 + 2], ro, eps_dot);
    const SpecG gq = spec_general<MODEL, SAMPLING>(t, ro);
    const float sv = soft_vis[e];
    for (int k = 0; k < 3; ++k) {
      const float sc = spec[r * 3 + k];
      const float Fs = sc + (1.f - sc) * gq.f;
      const float ek = env[e * C + (C == 1 ? 0 : k)];
      if (SPLIT) { acc[k] += gq.S * Fs * t.nol; accA[k] += sv * ek; }
      else acc[k] += gq.S * Fs * sv * ek * t.nol;
    }
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(acc[k], red);
    float a = 1.f;
    if (SPLIT) a = sh_block_sum(accA[k], red) / (float)M;
    if (threadIdx.x == 0) out[r * 3 + k] = weight * a * s / (float)M;
  }
}

template <int MODEL, int SAMPLING, bool SPLIT>
__global__ void __launch_bounds__(SH_THREADS) k_specular_light_g_bwd(int M, int C, const float* __restrict__ normal,
                                                                     const float* __restrict__ view, const float* __restrict__ light,
                                                                     const float* __restrict__ rough, const float* __restrict__ spec,
                                                                     const float* __restrict__ soft_vis, const float* __restrict__ env,
                                                                     float eps_dot, float weight, const float* __restrict__ g,
                                                                     float* __restrict__ g_normal, float* __restrict__ g_rough,
                                                                     float* __restrict__ g_spec, float* __restrict__ g_soft_vis,
                                                                     float* __restrict__ g_env) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = rough[r];
  const float wM = weight / (float)M;
  float gk[3], sc[3];
  for (int k = 0; k < 3; ++k) { gk[k] = g[r * 3 + k] * wM; sc[k] = spec[r * 3 + k]; }
  // split sum: out_k = w A_k B_k, A_k = mean(sv env_k), B_k = mean(S Fs_k nol) -> each factor's gradient carries the other mean
  float gA[3] = {0.f, 0.f, 0.f}, gB[3] = {gk[0], gk[1], gk[2]};
  if (SPLIT) {
    float sA[3] = {0.f, 0.f, 0.f}, sB[3] = {0.f, 0.f, 0.f};
    for (int m = threadIdx.x; m < M; m += SH_THREADS) {
      const long long e = r * M + m;
      const SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, light[e * 3], light[e * 3 + 1], light[e * 3 + 2], ro, eps_dot);
      const SpecG gq = spec_general<MODEL, SAMPLING>(t, ro);
      const float sv = soft_vis[e];
      for (int k = 0; k < 3; ++k) {
        sA[k] += sv * env[e * C + (C == 1 ? 0 : k)];
        sB[k] += gq.S * (sc[k] + (1.f - sc[k]) * gq.f) * t.nol;
      }
    }
    for (int k = 0; k < 3; ++k) {
      const float A = sh_block_sum(sA[k], red) / (float)M, Bm = sh_block_sum(sB[k], red) / (float)M;
      gA[k] = gk[k] * Bm;       // d out_k / d (sv env_k) per light
      gB[k] = gk[k] * A;        // d out_k / d (S Fs_k nol) per light
    }
  }
  float gn[3] = {0.f, 0.f, 0.f}, gr = 0.f, gsc[3] = {0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * M + m;
    const float lx = light[e * 3], ly = light[e * 3 + 1], lz = light[e * 3 + 2];
    const SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, lx, ly, lz, ro, eps_dot);
    const SpecG gq = spec_general<MODEL, SAMPLING>(t, ro);
    const float sv = soft_vis[e];
    float dS = 0.f, dnol = 0.f, dsv = 0.f;
    float denv[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
      const float ek = env[e * C + (C == 1 ? 0 : k)];
      const float Fs = sc[k] + (1.f - sc[k]) * gq.f;
      if (SPLIT) {
        dsv += gA[k] * ek;
        denv[C == 1 ? 0 : k] += gA[k] * sv;
        dS += gB[k] * Fs * t.nol;
        gsc[k] += gB[k] * gq.S * t.nol * (1.f - gq.f);
        dnol += gB[k] * gq.S * Fs;
      } else {
        const float T = sv * ek * t.nol;
        dsv += gk[k] * gq.S * Fs * ek * t.nol;
        denv[C == 1 ? 0 : k] += gk[k] * gq.S * Fs * sv * t.nol;
        dnol += gk[k] * gq.S * Fs * sv * ek;
        dS += gk[k] * Fs * T;
        gsc[k] += gk[k] * gq.S * T * (1.f - gq.f);
      }
    }
    g_soft_vis[e] = dsv;
    for (int k = 0; k < C; ++k) g_env[e * C + k] = denv[k];
    const float dW = dS * gq.K * t.mask, dK = dS * gq.W * t.mask;
    dnol += dW * gq.dW_dnol + dK * gq.dK_dnol;
    const float dnov = dW * gq.dW_dnov + dK * gq.dK_dnov;
    const float dnoh = dK * gq.dK_dnoh;
    gr += dW * gq.dW_dr + dK * gq.dK_dr;
    // clamped dots -> normal (clamp(min) passes the gradient where raw >= eps)
    const float cl = (t.rnol >= eps_dot) ? dnol : 0.f, cv = (t.rnov >= eps_dot) ? dnov : 0.f, ch = (t.rnoh >= eps_dot) ? dnoh : 0.f;
    gn[0] += cl * lx + cv * vx + ch * t.hx;
    gn[1] += cl * ly + cv * vy + ch * t.hy;
    gn[2] += cl * lz + cv * vz + ch * t.hz;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gn[k], red);
    if (threadIdx.x == 0) g_normal[r * 3 + k] = s;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gsc[k], red);
    if (threadIdx.x == 0) g_spec[r * 3 + k] = s;
  }
  {
    const float s = sh_block_sum(gr, red);
    if (threadIdx.x == 0) g_rough[r] = s;
  }
}

// ---- background head (python/network.py:543-556) ---------------------------------------------------------------------------
// h (P, 1 + F) = output of the background geometric net: density = softplus_100(h_0), alpha = 1 - exp(-density delta), and the
// lighting net's per-sample input  [x (nx) | feature (F)]  (its per-ray inputs, the view direction and its encoding, enter
// the fused chain as a row term).  One launch each way instead of slice / softplus / mul / neg / exp / rsub / cat and their
// backward launches; one thread per element of the (P, nx + F) input, the thread of column 0 also does the row's alpha.
__global__ void __launch_bounds__(256) k_background_head(long long P, int nx, int F, const float* __restrict__ h,
                                                         const float* __restrict__ x, const float* __restrict__ delta,
                                                         float* __restrict__ alpha, float* __restrict__ inp) {
  const int W = nx + F;
  const long long n = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    inp[i] = c < nx ? x[p * nx + c] : h[p * (F + 1) + 1 + (c - nx)];
    if (c == 0) {
      const float v = h[p * (F + 1)];
      const float density = 100.f * v > 20.f ? v : log1pf(expf(100.f * v)) / 100.f;
      alpha[p] = 1.f - expf(-density * delta[p]);
    }
  }
}

__global__ void __launch_bounds__(256) k_background_head_bwd(long long P, int nx, int F, const float* __restrict__ h,
                                                             const float* __restrict__ delta, const float* __restrict__ g_alpha,
                                                             const float* __restrict__ g_inp, float* __restrict__ g_h) {
  const int W = F + 1;
  const long long n = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    if (c > 0) {
      g_h[i] = g_inp ? g_inp[p * (nx + F) + nx + (c - 1)] : 0.f;
    } else {
      float gv = 0.f;
      if (g_alpha) {
        const float v = h[i], d = delta[p];
        const bool lin = 100.f * v > 20.f;
        const float e = lin ? 0.f : expf(100.f * v);
        const float density = lin ? v : log1pf(e) / 100.f;
        const float dsoft = lin ? 1.f : e / (e + 1.f);
        gv = g_alpha[p] * d * expf(-density * d) * dsoft;
      }
      g_h[i] = gv;
    }
  }
}

// ---- the SDF-to-density gain: clamp(exp(scale p), lo, hi) of a scalar parameter (python/network.py:229-231) -----------------
// Three elementwise functions forward and eight backward in the reference's spelling; one single-thread launch each way.
__global__ void k_gain(int n, const float* __restrict__ p, float scale, float lo, float hi, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fminf(fmaxf(expf(scale * p[i]), lo), hi);
}

__global__ void k_gain_bwd(int n, const float* __restrict__ p, float scale, float lo, float hi, const float* __restrict__ g,
                           float* __restrict__ gp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = expf(scale * p[i]);
  gp[i] = (e >= lo && e <= hi) ? g[i] * e * scale : 0.f;       // clamp passes the gradient inside [lo, hi] (bounds included)
}

// ---- both light integrals + the pixel composition of a ray in one launch each way ------------------------------------
// python/renderer.py:105-178 for the default branch (filament BRDF, importance sampling, no split sum, fused material head):
// the environment-light and soft-visibility nets are evaluated once over the 2 M directions [diffuse | specular] of a ray
// and hand over their RAW outputs; their output activations (python/network.py:288-296, 372-376), the two light integrals
// above and  diffuse = env + implicit;  color = base * diffuse + photo * spec  (entangle) | photo * (base * diffuse + spec);
// color += background  (csrc/loss.hip k_pixel_compose) run here per ray.  pix (R,9) = VR of the material head's V:
// [implicit, roughness, specular x3, photo, base x3].  The backward writes the gradient of the raw net outputs for all 2 M
// directions and the FULL g_pix row (roughness / specular columns from the specular integral) -- no slices of the 2 M
// tensors or of pix exist on either pass.
struct LightCfg {
  int M, C;                       // lights per integral, env channels (1 | 3)
  int act_sv, act_env;            // 0 identity, 1 softplus(beta), 2 sigmoid, 3 relu
  float beta_sv, beta_env;
  float ub_env;                   // > 0: clamp(act(env), 0, ub)
  float eps_dot, weight;
  int entangle;
};

__device__ __forceinline__ float light_act(int kind, float beta, float v, float& d) {
  if (kind == 1) {                // TF.softplus(v, beta), threshold 20
    const float bv = beta * v;
    if (bv > 20.f) { d = 1.f; return v; }
    const float e = expf(bv);
    d = e / (e + 1.f);
    return log1pf(e) / beta;
  }
  if (kind == 2) { const float y = 1.f / (1.f + expf(-v)); d = y * (1.f - y); return y; }
  if (kind == 3) { d = v > 0.f ? 1.f : 0.f; return fmaxf(v, 0.f); }
  d = 1.f;
  return v;
}

__device__ __forceinline__ float env_act(const LightCfg& c, float v, float& d) {
  float y = light_act(c.act_env, c.beta_env, v, d);
  if (c.ub_env > 0.f) {
    if (!(y >= 0.f && y <= c.ub_env)) d = 0.f;
    y = fminf(fmaxf(y, 0.f), c.ub_env);
  }
  return y;
}

__global__ void __launch_bounds__(SH_THREADS) k_direct_light(LightCfg c, const float* __restrict__ normal, const float* __restrict__ view,
                                                             const float* __restrict__ dirs, const float* __restrict__ raw_sv,
                                                             const float* __restrict__ raw_env, const float* __restrict__ pix,
                                                             const float* __restrict__ bg, float* __restrict__ color,
                                                             float* __restrict__ env_pix, float* __restrict__ spec_pix) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const int M = c.M, C = c.C;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = pix[r * 9 + 1];
  float dsum[3] = {0.f, 0.f, 0.f}, ssum[3] = {0.f, 0.f, 0.f}, d_;
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * 2 * M + m;
    const float cs = fmaxf(nx * dirs[e * 3] + ny * dirs[e * 3 + 1] + nz * dirs[e * 3 + 2], c.eps_dot);
    const float t = light_act(c.act_sv, c.beta_sv, raw_sv[e], d_) * cs;
    for (int k = 0; k < C; ++k) dsum[k] += t * env_act(c, raw_env[e * C + k], d_);
  }
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * 2 * M + M + m;
    SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, dirs[e * 3], dirs[e * 3 + 1], dirs[e * 3 + 2], ro, c.eps_dot);
    const float V = t.V1l * t.V1v;
    const float f5 = powf(1.f - t.voh, 5.f);
    const float Kf = 4.f * t.voh / t.noh * t.mask;
    const float sv = light_act(c.act_sv, c.beta_sv, raw_sv[e], d_);
    float ev[3];
    for (int k = 0; k < C; ++k) ev[k] = env_act(c, raw_env[e * C + k], d_);
    for (int k = 0; k < 3; ++k) {
      const float sc = pix[r * 9 + 2 + k];
      const float Fs = sc + (1.f - sc) * f5;
      ssum[k] += V * Fs * Kf * sv * ev[C == 1 ? 0 : k] * t.nol;
    }
  }
  float envp[3] = {0.f, 0.f, 0.f}, specp[3];
  for (int k = 0; k < C; ++k) envp[k] = sh_block_sum(dsum[k], red) / (float)M;
  for (int k = 0; k < 3; ++k) specp[k] = c.weight * sh_block_sum(ssum[k], red) / (float)M;
  if (threadIdx.x == 0) {
    const float imp = pix[r * 9], photo = pix[r * 9 + 5];
    for (int k = 0; k < C; ++k) env_pix[r * C + k] = envp[k];
    for (int k = 0; k < 3; ++k) {
      spec_pix[r * 3 + k] = specp[k];
      const float diff = envp[C == 3 ? k : 0] + imp;
      const float base = pix[r * 9 + 6 + k];
      const float fg = c.entangle ? base * diff + photo * specp[k] : photo * (base * diff + specp[k]);
      color[r * 3 + k] = fg + (bg ? bg[r * 3 + k] : 0.f);
    }
  }
}

__global__ void __launch_bounds__(SH_THREADS) k_direct_light_bwd(LightCfg c, const float* __restrict__ normal,
                                                                 const float* __restrict__ view, const float* __restrict__ dirs,
                                                                 const float* __restrict__ raw_sv, const float* __restrict__ raw_env,
                                                                 const float* __restrict__ pix, const float* __restrict__ env_pix,
                                                                 const float* __restrict__ spec_pix, const float* __restrict__ g,
                                                                 float* __restrict__ g_normal, float* __restrict__ g_raw_sv,
                                                                 float* __restrict__ g_raw_env, float* __restrict__ g_pix,
                                                                 float* __restrict__ g_bg) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const int M = c.M, C = c.C;
  const float nx = normal[r * 3], ny = normal[r * 3 + 1], nz = normal[r * 3 + 2];
  const float vx = view[r * 3], vy = view[r * 3 + 1], vz = view[r * 3 + 2];
  const float ro = pix[r * 9 + 1];
  // composition backward (every thread: it needs g_env_pix / g_spec_pix for its lights)
  const float imp = pix[r * 9], photo = pix[r * 9 + 5];
  float g_imp = 0.f, g_photo = 0.f, gd[3], gk[3], sc[3], g_base[3];
  for (int k = 0; k < 3; ++k) {
    const float gc = g[r * 3 + k];
    const float diff = env_pix[r * C + (C == 3 ? k : 0)] + imp;
    const float base = pix[r * 9 + 6 + k], sp = spec_pix[r * 3 + k];
    float g_sp;
    if (c.entangle) { g_base[k] = gc * diff; gd[k] = gc * base; g_photo += gc * sp; g_sp = gc * photo; }
    else { g_photo += gc * (base * diff + sp); g_base[k] = gc * photo * diff; gd[k] = gc * photo * base; g_sp = gc * photo; }
    g_imp += gd[k];
    gk[k] = g_sp * c.weight / (float)M;
    sc[k] = pix[r * 9 + 2 + k];
  }
  float ge_pix[3];                       // d / d env_pixel
  if (C == 3) { ge_pix[0] = gd[0]; ge_pix[1] = gd[1]; ge_pix[2] = gd[2]; }
  else { ge_pix[0] = gd[0] + gd[1] + gd[2]; ge_pix[1] = ge_pix[2] = 0.f; }
  const float inv = 1.f / (float)M;
  float gn[3] = {0.f, 0.f, 0.f}, ga2 = 0.f, gsc[3] = {0.f, 0.f, 0.f};
  // diffuse integral
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * 2 * M + m;
    const float lx = dirs[e * 3], ly = dirs[e * 3 + 1], lz = dirs[e * 3 + 2];
    const float raw = nx * lx + ny * ly + nz * lz;
    const float cs = fmaxf(raw, c.eps_dot);
    float dsv;
    const float sv = light_act(c.act_sv, c.beta_sv, raw_sv[e], dsv);
    float ge = 0.f;
    for (int k = 0; k < C; ++k) {
      float de;
      const float ev = env_act(c, raw_env[e * C + k], de);
      const float gkk = ge_pix[k] * inv;
      ge += gkk * ev;
      g_raw_env[e * C + k] = gkk * sv * cs * de;
    }
    g_raw_sv[e] = ge * cs * dsv;
    const float gc = (raw >= c.eps_dot) ? ge * sv : 0.f;
    gn[0] += gc * lx; gn[1] += gc * ly; gn[2] += gc * lz;
  }
  // specular integral
  for (int m = threadIdx.x; m < M; m += SH_THREADS) {
    const long long e = r * 2 * M + M + m;
    const float lx = dirs[e * 3], ly = dirs[e * 3 + 1], lz = dirs[e * 3 + 2];
    SpecTerms t = spec_terms(nx, ny, nz, vx, vy, vz, lx, ly, lz, ro, c.eps_dot);
    const float V = t.V1l * t.V1v;
    const float omv = 1.f - t.voh;
    const float f4 = omv * omv * omv * omv, f5 = f4 * omv;
    const float Kf = 4.f * t.voh / t.noh * t.mask;
    float dsv_;
    const float sv = light_act(c.act_sv, c.beta_sv, raw_sv[e], dsv_);
    float ev[3], de[3];
    for (int k = 0; k < C; ++k) ev[k] = env_act(c, raw_env[e * C + k], de[k]);
    float dV = 0.f, dK = 0.f, dnol = 0.f, dsv = 0.f;
    float denv[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
      const float ek = ev[C == 1 ? 0 : k];
      const float Fs = sc[k] + (1.f - sc[k]) * f5;
      const float sB = V * Fs * Kf;
      const float dsB = gk[k] * sv * ek * t.nol;
      dsv += gk[k] * sB * ek * t.nol;
      denv[C == 1 ? 0 : k] += gk[k] * sB * sv * t.nol;
      dnol += gk[k] * sB * sv * ek;
      dV += dsB * Fs * Kf;
      dK += dsB * V * Fs;
      gsc[k] += dsB * V * Kf * (1.f - f5);
    }
    g_raw_sv[e] = dsv * dsv_;
    for (int k = 0; k < C; ++k) g_raw_env[e * C + k] = denv[k] * de[k];
    const float dnoh = -dK * 4.f * t.voh / (t.noh * t.noh) * t.mask;     // (voh only reaches h: no gradient)
    const float dV1l = dV * t.V1v, dV1v = dV * t.V1l;
    const float om = 1.f - t.a2;
    dnol += dV1l * (-t.V1l * t.V1l) * (1.f + om * t.nol / t.sl);
    const float dnov = dV1v * (-t.V1v * t.V1v) * (1.f + om * t.nov / t.sv_);
    ga2 += dV1l * (-t.V1l * t.V1l) * (1.f - t.nol * t.nol) / (2.f * t.sl) + dV1v * (-t.V1v * t.V1v) * (1.f - t.nov * t.nov) / (2.f * t.sv_);
    const float cl = (t.rnol >= c.eps_dot) ? dnol : 0.f, cv = (t.rnov >= c.eps_dot) ? dnov : 0.f, ch = (t.rnoh >= c.eps_dot) ? dnoh : 0.f;
    gn[0] += cl * lx + cv * vx + ch * t.hx;
    gn[1] += cl * ly + cv * vy + ch * t.hy;
    gn[2] += cl * lz + cv * vz + ch * t.hz;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gn[k], red);
    if (threadIdx.x == 0) g_normal[r * 3 + k] = s;
  }
  for (int k = 0; k < 3; ++k) {
    const float s = sh_block_sum(gsc[k], red);
    if (threadIdx.x == 0) g_pix[r * 9 + 2 + k] = s;
  }
  {
    const float s = sh_block_sum(ga2, red);
    if (threadIdx.x == 0) g_pix[r * 9 + 1] = s * 2.f * ro;
  }
  if (threadIdx.x == 0) {
    g_pix[r * 9] = g_imp;
    g_pix[r * 9 + 5] = g_photo;
    for (int k = 0; k < 3; ++k) {
      g_pix[r * 9 + 6 + k] = g_base[k];
      if (g_bg) g_bg[r * 3 + k] = g[r * 3 + k];
    }
  }
}

}  // namespace ndjir

extern "C" int ndjir_render_background_head(long long P, int nx, int F, const float* h, const float* x, const float* delta,
                                            float* alpha, float* inp, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (nx < 1 || F < 1) return NDJIR_ERR_UNSUPPORTED;
  if (!h || !x || !delta || !alpha || !inp) return NDJIR_ERR_ARG;
  const long long blocks = (P * (nx + F) + 255) / 256;
  hipLaunchKernelGGL(ndjir::k_background_head, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, P, nx, F, h, x,
                     delta, alpha, inp);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_background_head_backward(long long P, int nx, int F, const float* h, const float* delta,
                                                     const float* g_alpha, const float* g_inp, float* g_h, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (nx < 1 || F < 1) return NDJIR_ERR_UNSUPPORTED;
  if (!h || !delta || !g_h) return NDJIR_ERR_ARG;
  const long long blocks = (P * (F + 1) + 255) / 256;
  hipLaunchKernelGGL(ndjir::k_background_head_bwd, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, P, nx, F, h,
                     delta, g_alpha, g_inp, g_h);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_gain(int n, const float* p, float scale, float lo, float hi, float* out, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!p || !out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(ndjir::k_gain, dim3((n + 63) / 64), dim3(64), 0, stream, n, p, scale, lo, hi, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_gain_backward(int n, const float* p, float scale, float lo, float hi, const float* g, float* gp,
                                          hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!p || !g || !gp) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(ndjir::k_gain_bwd, dim3((n + 63) / 64), dim3(64), 0, stream, n, p, scale, lo, hi, g, gp);
  return ndjir_check_launch();
}

static int light_cfg(ndjir::LightCfg& c, int M, int C, const int* acts, const float* params, int entangle) {
  if (M < 1 || (C != 1 && C != 3) || !acts || !params) return NDJIR_ERR_ARG;
  if (acts[0] < 0 || acts[0] > 3 || acts[1] < 0 || acts[1] > 3) return NDJIR_ERR_UNSUPPORTED;
  c.M = M; c.C = C; c.act_sv = acts[0]; c.act_env = acts[1];
  c.beta_sv = params[0]; c.beta_env = params[1]; c.ub_env = params[2]; c.eps_dot = params[3]; c.weight = params[4];
  c.entangle = entangle;
  return NDJIR_OK;
}

extern "C" int ndjir_render_direct_light(int R, int M, int C, const int* acts, const float* params, int entangle, const float* normal,
                                         const float* view_dir, const float* light_dirs, const float* raw_soft_vis,
                                         const float* raw_env, const float* pix, const float* bg, float* color, float* env_pixel,
                                         float* spec_pixel, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  ndjir::LightCfg c;
  if (int e = light_cfg(c, M, C, acts, params, entangle)) return e;
  if (!normal || !view_dir || !light_dirs || !raw_soft_vis || !raw_env || !pix || !color || !env_pixel || !spec_pixel) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(ndjir::k_direct_light, dim3(R), dim3(ndjir::SH_THREADS), 0, stream, c, normal, view_dir, light_dirs, raw_soft_vis,
                     raw_env, pix, bg, color, env_pixel, spec_pixel);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_direct_light_backward(int R, int M, int C, const int* acts, const float* params, int entangle,
                                                  const float* normal, const float* view_dir, const float* light_dirs,
                                                  const float* raw_soft_vis, const float* raw_env, const float* pix,
                                                  const float* env_pixel, const float* spec_pixel, const float* g_color,
                                                  float* g_normal, float* g_raw_soft_vis, float* g_raw_env, float* g_pix, float* g_bg,
                                                  hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  ndjir::LightCfg c;
  if (int e = light_cfg(c, M, C, acts, params, entangle)) return e;
  if (!normal || !view_dir || !light_dirs || !raw_soft_vis || !raw_env || !pix || !env_pixel || !spec_pixel || !g_color || !g_normal ||
      !g_raw_soft_vis || !g_raw_env || !g_pix)
    return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(ndjir::k_direct_light_bwd, dim3(R), dim3(ndjir::SH_THREADS), 0, stream, c, normal, view_dir, light_dirs,
                     raw_soft_vis, raw_env, pix, env_pixel, spec_pixel, g_color, g_normal, g_raw_soft_vis, g_raw_env, g_pix, g_bg);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_diffuse_light(int R, int M, int C, const float* normal, const float* light_dir, const float* soft_vis,
                                          const float* env, float eps_dot, float* out, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || C < 1 || C > 3) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !light_dir || !soft_vis || !env || !out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_diffuse_light, dim3(R), dim3(SH_THREADS), 0, stream, M, C, normal, light_dir, soft_vis, env, eps_dot, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_diffuse_light_backward(int R, int M, int C, const float* normal, const float* light_dir,
                                                   const float* soft_vis, const float* env, float eps_dot, const float* g,
                                                   float* g_normal, float* g_soft_vis, float* g_env, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || C < 1 || C > 3) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !light_dir || !soft_vis || !env || !g || !g_normal || !g_soft_vis || !g_env) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_diffuse_light_bwd, dim3(R), dim3(SH_THREADS), 0, stream, M, C, normal, light_dir, soft_vis, env, eps_dot, g,
                     g_normal, g_soft_vis, g_env);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_specular_light_filament(int R, int M, int C, const float* normal, const float* view_dir,
                                                    const float* light_dir, const float* roughness, const float* specular_color,
                                                    const float* soft_vis, const float* env, float eps_dot, float weight,
                                                    float* out, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || (C != 1 && C != 3)) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !view_dir || !light_dir || !roughness || !specular_color || !soft_vis || !env || !out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_specular_light, dim3(R), dim3(SH_THREADS), 0, stream, M, C, normal, view_dir, light_dir, roughness,
                     specular_color, soft_vis, env, eps_dot, weight, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_specular_light_filament_backward(int R, int M, int C, const float* normal, const float* view_dir,
                                                             const float* light_dir, const float* roughness,
                                                             const float* specular_color, const float* soft_vis, const float* env,
                                                             float eps_dot, float weight, const float* g, float* g_normal,
                                                             float* g_roughness, float* g_specular_color, float* g_soft_vis,
                                                             float* g_env, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || (C != 1 && C != 3)) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !view_dir || !light_dir || !roughness || !specular_color || !soft_vis || !env || !g || !g_normal ||
      !g_roughness || !g_specular_color || !g_soft_vis || !g_env)
    return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_specular_light_bwd, dim3(R), dim3(SH_THREADS), 0, stream, M, C, normal, view_dir, light_dir, roughness,
                     specular_color, soft_vis, env, eps_dot, weight, g, g_normal, g_roughness, g_specular_color, g_soft_vis, g_env);
  return ndjir_check_launch();
}

// every model / sampling / split-sum combination (see k_specular_light_g)
#define NDJIR_SPEC_DISPATCH(KERNEL, ...)                                                                                   \
  do {                                                                                                                     \
    const int key = (model << 2) | (sampling << 1) | (split ? 1 : 0);                                                      \
    switch (key) {                                                                                                         \
      case 0: hipLaunchKernelGGL((ndjir::KERNEL<0, 0, false>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break; \
      case 1: hipLaunchKernelGGL((ndjir::KERNEL<0, 0, true>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break;  \
      case 2: hipLaunchKernelGGL((ndjir::KERNEL<0, 1, false>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break; \
      case 3: hipLaunchKernelGGL((ndjir::KERNEL<0, 1, true>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break;  \
      case 4: hipLaunchKernelGGL((ndjir::KERNEL<1, 0, false>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break; \
      case 5: hipLaunchKernelGGL((ndjir::KERNEL<1, 0, true>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break;  \
      case 6: hipLaunchKernelGGL((ndjir::KERNEL<1, 1, false>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((ndjir::KERNEL<1, 1, true>), dim3(R), dim3(ndjir::SH_THREADS), 0, stream, __VA_ARGS__); break; \
    }                                                                                                                      \
  } while (0)

extern "C" int ndjir_render_specular_light(int R, int M, int C, int model, int sampling, int split, const float* normal,
                                           const float* view_dir, const float* light_dir, const float* roughness,
                                           const float* specular_color, const float* soft_vis, const float* env, float eps_dot,
                                           float weight, float* out, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || (C != 1 && C != 3) || model < 0 || model > 1 || sampling < 0 || sampling > 1) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !view_dir || !light_dir || !roughness || !specular_color || !soft_vis || !env || !out) return NDJIR_ERR_ARG;
  NDJIR_SPEC_DISPATCH(k_specular_light_g, M, C, normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_specular_light_backward(int R, int M, int C, int model, int sampling, int split, const float* normal,
                                                    const float* view_dir, const float* light_dir, const float* roughness,
                                                    const float* specular_color, const float* soft_vis, const float* env,
                                                    float eps_dot, float weight, const float* g, float* g_normal, float* g_roughness,
                                                    float* g_specular_color, float* g_soft_vis, float* g_env, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (M < 1 || (C != 1 && C != 3) || model < 0 || model > 1 || sampling < 0 || sampling > 1) return NDJIR_ERR_UNSUPPORTED;
  if (!normal || !view_dir || !light_dir || !roughness || !specular_color || !soft_vis || !env || !g || !g_normal ||
      !g_roughness || !g_specular_color || !g_soft_vis || !g_env)
    return NDJIR_ERR_ARG;
  NDJIR_SPEC_DISPATCH(k_specular_light_g_bwd, M, C, normal, view_dir, light_dir, roughness, specular_color, soft_vis, env, eps_dot, weight,
                      g, g_normal, g_roughness, g_specular_color, g_soft_vis, g_env);
  return ndjir_check_launch();
}
#undef NDJIR_SPEC_DISPATCH

// ---- positional encoding (python/network.py:96-117) ----------------------------------------------------
// out[p] = [x (C), cos(x_i 2^k) (C*M, band fastest), sin(x_i 2^k) (C*M)]; the reference builds it from
// a broadcast multiply, cos, sin and a concatenate.  One thread per output element.
namespace ndjir {

__global__ void __launch_bounds__(256) k_posenc(long long P, int C, int M, int include_input, const float* __restrict__ x,
                                                float* __restrict__ out) {
  const int CM = C * M, W = (include_input ? C : 0) + 2 * CM;
  const long long total = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    int c = rc.c;
    float v;
    if (include_input && c < C) v = x[p * C + c];
    else {
      c -= include_input ? C : 0;
      const bool is_sin = c >= CM;
      if (is_sin) c -= CM;
      const float b = x[p * C + c / M] * (float)(1 << (c % M));
      v = is_sin ? sinf(b) : cosf(b);
    }
    out[t] = v;
  }
}

// gx[p][i] = g_x[p][i] + sum_k 2^k ( -sin(b) g_cos + cos(b) g_sin )
__global__ void __launch_bounds__(256) k_posenc_bwd(long long P, int C, int M, int include_input, const float* __restrict__ x,
                                                    const float* __restrict__ g, float* __restrict__ gx) {
  const int CM = C * M, W = (include_input ? C : 0) + 2 * CM, off = include_input ? C : 0;
  const long long total = P * C;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, C);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int i = rc.c;
    const float xv = x[t];
    const float* gp = g + p * W;
    float acc = include_input ? gp[i] : 0.f;
    for (int k = 0; k < M; ++k) {
      const float s = (float)(1 << k), b = xv * s;
      acc += s * (-sinf(b) * gp[off + i * M + k] + cosf(b) * gp[off + CM + i * M + k]);
    }
    gx[t] = acc;
  }
}

}  // namespace ndjir

extern "C" int ndjir_positional_encoding(long long P, int C, int M, int include_input, const float* x, float* out,
                                         hipStream_t stream) {
  if (P <= 0 || C <= 0) return NDJIR_OK;
  if (M < 0 || M > 30 || !x || !out) return NDJIR_ERR_ARG;
  const long long total = P * ((include_input ? C : 0) + 2LL * C * M);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_posenc, dim3((unsigned)blocks), dim3(256), 0, stream, P, C, M, include_input, x, out);
  return ndjir_check_launch();
}

extern "C" int ndjir_positional_encoding_backward(long long P, int C, int M, int include_input, const float* x, const float* g,
                                                  float* gx, hipStream_t stream) {
  if (P <= 0 || C <= 0) return NDJIR_OK;
  if (M < 0 || M > 30 || !x || !g || !gx) return NDJIR_ERR_ARG;
  long long blocks = (P * C + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_posenc_bwd, dim3((unsigned)blocks), dim3(256), 0, stream, P, C, M, include_input, x, g, gx);
  return ndjir_check_launch();
}

// ---- material head: output activations of the per-sample nets + the prior terms ------------------------
// python/network.py:262 (base colour: sigmoid), :335 (implicit illumination: sigmoid), :423 (photogrammetric
// light: sigmoid(gain h)), :456-463 (roughness: sigmoid [^2] clamp, std = softplus), :498-508 (specular
// reflectance: sigmoid -> 0.16 s^2 | scale s, std = softplus) and the prior / regulariser integrands of
// python/loss.py:117-166, which the reference evaluates as ~60 nnabla functions (+ backward) on (B,R,N,*)
// tensors.  One thread per sample, one workgroup per ray (N <= 1024):
//   V      (P,9)  = [implicit, roughness, spec x3, photo, base (* photo if entangle) x3]  -> ONE VR integral
//   aux    (P,10) = [base x3, base_ptb x3, std_rough, std_spec x3]
//   prior  (R,5)  = per-ray sums of |base - base_ptb| (3 ch), |r - p_r| / std_r, clamp(log std_r),
//                   sum_c |s_c - p_s| / std_s_c, sum_c clamp(log std_s_c)      (clamp to [1e-5, 1e5])
namespace ndjir {

struct HeadCfg {
  int remap, entangle, sym;
  float rough_lb, spec_scale, prior_r, prior_s;
};

__device__ __forceinline__ float mh_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float mh_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch: beta 1, threshold 20
__device__ __forceinline__ float mh_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(128) k_material_head(int N, const float* __restrict__ raw_bc, const float* __restrict__ raw_ptb,
                                                       const float* __restrict__ raw_imp, const float* __restrict__ raw_photo,
                                                       const float* __restrict__ photo_gain, const float* __restrict__ raw_rough,
                                                       const float* __restrict__ raw_spec, HeadCfg c, float* __restrict__ V,
                                                       float* __restrict__ aux, float* __restrict__ prior) {
  __shared__ float red[2];
  const long long r = blockIdx.x;
  const float pg = photo_gain[0];
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < N; i += 128) {
    const long long p = r * N + i;
    float bc[3], pt[3];
    for (int k = 0; k < 3; ++k) { bc[k] = mh_sigmoid(raw_bc[p * 3 + k]); pt[k] = mh_sigmoid(raw_ptb[p * 3 + k]); }
    const float imp = mh_sigmoid(raw_imp[p]);
    const float photo = mh_sigmoid(pg * raw_photo[p]);
    const float r0 = mh_sigmoid(raw_rough[p * 2]);
    const float r1 = c.remap ? r0 * r0 : r0;
    const float ro = fminf(fmaxf(r1, c.rough_lb), 1.f);
    const float sr = mh_softplus(raw_rough[p * 2 + 1]);
    float sp[3], ss[3];
    for (int k = 0; k < 3; ++k) {
      const float s0 = mh_sigmoid(raw_spec[p * 6 + k]);
      sp[k] = c.remap ? 0.16f * (s0 * s0) : c.spec_scale * s0;
      ss[k] = mh_softplus(raw_spec[p * 6 + 3 + k]);
    }
    float* v = V + p * 9;
    v[0] = imp; v[1] = ro; v[2] = sp[0]; v[3] = sp[1]; v[4] = sp[2]; v[5] = photo;
    for (int k = 0; k < 3; ++k) v[6 + k] = c.entangle ? bc[k] * photo : bc[k];
    float* a = aux + p * 10;
    for (int k = 0; k < 3; ++k) { a[k] = bc[k]; a[3 + k] = pt[k]; a[7 + k] = ss[k]; }
    a[6] = sr;
    for (int k = 0; k < 3; ++k) acc[0] += fabsf(bc[k] - pt[k]);
    acc[1] += fabsf(ro - c.prior_r) / sr;
    acc[2] += fminf(fmaxf(logf(sr), 1e-5f), 1e5f);
    for (int k = 0; k < 3; ++k) {
      acc[3] += fabsf(sp[k] - c.prior_s) / ss[k];
      acc[4] += fminf(fmaxf(logf(ss[k]), 1e-5f), 1e5f);
    }
  }
  for (int k = 0; k < 5; ++k) {
    float v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) prior[r * 5 + k] = red[0] + red[1];
  }
}

// gV (P,9), g_prior (R,5) -> gradients of the raw net outputs
__global__ void __launch_bounds__(128) k_material_head_bwd(int N, const float* __restrict__ raw_bc, const float* __restrict__ raw_ptb,
                                                           const float* __restrict__ raw_imp, const float* __restrict__ raw_photo,
                                                           const float* __restrict__ photo_gain, const float* __restrict__ raw_rough,
                                                           const float* __restrict__ raw_spec, HeadCfg c,
                                                           const float* __restrict__ gV, const float* __restrict__ g_prior,
                                                           float* __restrict__ g_bc, float* __restrict__ g_ptb,
                                                           float* __restrict__ g_imp, float* __restrict__ g_photo,
                                                           float* __restrict__ g_rough, float* __restrict__ g_spec) {
  const long long r = blockIdx.x;
  const float pg = photo_gain[0];
  float gp[5];
  for (int k = 0; k < 5; ++k) gp[k] = g_prior ? g_prior[r * 5 + k] : 0.f;
  for (int i = threadIdx.x; i < N; i += 128) {
    const long long p = r * N + i;
    const float* gv = gV + p * 9;
    const float photo = mh_sigmoid(pg * raw_photo[p]);
    float dphoto = gv[5];
    for (int k = 0; k < 3; ++k) {
      const float bc = mh_sigmoid(raw_bc[p * 3 + k]), pt = mh_sigmoid(raw_ptb[p * 3 + k]);
      const float sg = mh_sign(bc - pt);
      float dbc = gv[6 + k] * (c.entangle ? photo : 1.f) + (c.sym ? gp[0] * sg : 0.f);
      if (c.entangle) dphoto += gv[6 + k] * bc;
      g_bc[p * 3 + k] = dbc * bc * (1.f - bc);
      g_ptb[p * 3 + k] = -gp[0] * sg * pt * (1.f - pt);
    }
    const float imp = mh_sigmoid(raw_imp[p]);
    g_imp[p] = gv[0] * imp * (1.f - imp);
    g_photo[p] = dphoto * photo * (1.f - photo) * pg;
    {
      const float h1 = raw_rough[p * 2 + 1];
      const float r0 = mh_sigmoid(raw_rough[p * 2]);
      const float r1 = c.remap ? r0 * r0 : r0;
      const float ro = fminf(fmaxf(r1, c.rough_lb), 1.f);
      const float sr = mh_softplus(h1);
      const float dr = gv[1] + gp[1] * mh_sign(ro - c.prior_r) / sr;
      const float dr1 = (r1 >= c.rough_lb && r1 <= 1.f) ? dr : 0.f;
      const float dr0 = c.remap ? dr1 * 2.f * r0 : dr1;
      g_rough[p * 2] = dr0 * r0 * (1.f - r0);
      const float ls = logf(sr);
      const float dsr = gp[1] * (-fabsf(ro - c.prior_r) / (sr * sr)) + ((ls >= 1e-5f && ls <= 1e5f) ? gp[2] / sr : 0.f);
      g_rough[p * 2 + 1] = dsr * (h1 > 20.f ? 1.f : mh_sigmoid(h1));
    }
    for (int k = 0; k < 3; ++k) {
      const float h1 = raw_spec[p * 6 + 3 + k];
      const float s0 = mh_sigmoid(raw_spec[p * 6 + k]);
      const float sp = c.remap ? 0.16f * (s0 * s0) : c.spec_scale * s0;
      const float ss = mh_softplus(h1);
      const float ds = gv[2 + k] + gp[3] * mh_sign(sp - c.prior_s) / ss;
      const float ds0 = c.remap ? ds * 0.32f * s0 : ds * c.spec_scale;
      g_spec[p * 6 + k] = ds0 * s0 * (1.f - s0);
      const float ls = logf(ss);
      const float dss = gp[3] * (-fabsf(sp - c.prior_s) / (ss * ss)) + ((ls >= 1e-5f && ls <= 1e5f) ? gp[4] / ss : 0.f);
      g_spec[p * 6 + 3 + k] = dss * (h1 > 20.f ? 1.f : mh_sigmoid(h1));
    }
  }
}

}  // namespace ndjir

extern "C" int ndjir_render_material_head(int R, int N, const float* raw_base_color, const float* raw_base_color_ptb,
                                          const float* raw_implicit, const float* raw_photo, const float* photo_gain,
                                          const float* raw_roughness, const float* raw_specular, int remap, int entangle,
                                          int sym_backward, float roughness_lower_bound, float specular_scale,
                                          float roughness_prior, float specular_prior, float* V, float* aux, float* prior,
                                          hipStream_t stream) {
  if (R <= 0 || N <= 0) return NDJIR_OK;
  if (!raw_base_color || !raw_base_color_ptb || !raw_implicit || !raw_photo || !photo_gain || !raw_roughness || !raw_specular ||
      !V || !aux || !prior)
    return NDJIR_ERR_ARG;
  HeadCfg c{remap, entangle, sym_backward, roughness_lower_bound, specular_scale, roughness_prior, specular_prior};
  hipLaunchKernelGGL(k_material_head, dim3(R), dim3(128), 0, stream, N, raw_base_color, raw_base_color_ptb, raw_implicit,
                     raw_photo, photo_gain, raw_roughness, raw_specular, c, V, aux, prior);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_material_head_backward(int R, int N, const float* raw_base_color, const float* raw_base_color_ptb,
                                                   const float* raw_implicit, const float* raw_photo, const float* photo_gain,
                                                   const float* raw_roughness, const float* raw_specular, int remap,
                                                   int entangle, int sym_backward, float roughness_lower_bound,
                                                   float specular_scale, float roughness_prior, float specular_prior,
                                                   const float* gV, const float* g_prior, float* g_base_color,
                                                   float* g_base_color_ptb, float* g_implicit, float* g_photo,
                                                   float* g_roughness, float* g_specular, hipStream_t stream) {
  if (R <= 0 || N <= 0) return NDJIR_OK;
  if (!raw_base_color || !raw_base_color_ptb || !raw_implicit || !raw_photo || !photo_gain || !raw_roughness || !raw_specular ||
      !gV || !g_base_color || !g_base_color_ptb || !g_implicit || !g_photo || !g_roughness || !g_specular)
    return NDJIR_ERR_ARG;
  HeadCfg c{remap, entangle, sym_backward, roughness_lower_bound, specular_scale, roughness_prior, specular_prior};
  hipLaunchKernelGGL(k_material_head_bwd, dim3(R), dim3(128), 0, stream, N, raw_base_color, raw_base_color_ptb, raw_implicit,
                     raw_photo, photo_gain, raw_roughness, raw_specular, c, gV, g_prior, g_base_color, g_base_color_ptb, g_implicit,
                     g_photo, g_roughness, g_specular);
  return ndjir_check_launch();
}
