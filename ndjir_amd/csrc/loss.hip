// loss.hip -- the per-ray tail of the step as fused kernels (forward + hand-derived backward):
//   * pixel composition, python/renderer.py:163-176 + :178: diffuse + specular + background -> color_pixel;
//   * the loss terms of python/loss.py:59-166 that are reductions over rays / samples: RGB error, eikonal term, sampled
//     TV term(s), the five prior / regulariser sums -- one pass over the rays, per-ray partial sums, a fixed-order final
//     reduction (no float atomics: the loss is bit-reproducible), then the weighted total.
// The reference builds these from ~40 nnabla functions (each an own launch, and as many again in backward).
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include "common.h"

namespace ndjir {

// ---- pixel normal, python/renderer.py:90-91: n = (VR(grad) + eps) / |VR(grad) + eps| ------------------------------------
__global__ void __launch_bounds__(256) k_pixel_normal(int R, float eps, const float* __restrict__ g, float* __restrict__ n) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float x = g[r * 3] + eps, y = g[r * 3 + 1] + eps, z = g[r * 3 + 2] + eps;
  const float len = sqrtf(x * x + y * y + z * z);
  n[r * 3] = x / len;
  n[r * 3 + 1] = y / len;
  n[r * 3 + 2] = z / len;
}

// d/dg of the above: (gn - n (n . gn)) / len
__global__ void __launch_bounds__(256) k_pixel_normal_bwd(int R, float eps, const float* __restrict__ g, const float* __restrict__ gn,
                                                          float* __restrict__ gg) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float x = g[r * 3] + eps, y = g[r * 3 + 1] + eps, z = g[r * 3 + 2] + eps;
  const float len = sqrtf(x * x + y * y + z * z);
  const float nx = x / len, ny = y / len, nz = z / len;
  const float a = gn[r * 3], b = gn[r * 3 + 1], c = gn[r * 3 + 2];
  const float dot = nx * a + ny * b + nz * c;
  gg[r * 3] = (a - nx * dot) / len;
  gg[r * 3 + 1] = (b - ny * dot) / len;
  gg[r * 3 + 2] = (c - nz * dot) / len;
}

// ---- pixel composition ----------------------------------------------------------------------------------------------
// pix (R,9) = VR of [implicit, roughness, specular x3, photo, base term x3]; env (R,Ce) diffuse light integral, Ce = 1 or 3;
// spec (R,3); bg (R,3).  diffuse = env + implicit;
// entangle: color = base * diffuse + photo * spec   else: color = photo * (base * diffuse + spec);   color += bg
__global__ void __launch_bounds__(256) k_pixel_compose(int R, int Ce, int entangle, const float* __restrict__ pix,
                                                       const float* __restrict__ env, const float* __restrict__ spec,
                                                       const float* __restrict__ bg, float* __restrict__ color) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float imp = pix[r * 9], photo = pix[r * 9 + 5];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float diff = env[r * Ce + (Ce == 3 ? c : 0)] + imp;
    const float base = pix[r * 9 + 6 + c], sp = spec[r * 3 + c];
    const float fg = entangle ? base * diff + photo * sp : photo * (base * diff + sp);
    color[r * 3 + c] = fg + (bg ? bg[r * 3 + c] : 0.f);
  }
}

__global__ void __launch_bounds__(256) k_pixel_compose_bwd(int R, int Ce, int entangle, const float* __restrict__ pix,
                                                           const float* __restrict__ env, const float* __restrict__ spec,
                                                           const float* __restrict__ g, float* __restrict__ g_pix,
                                                           float* __restrict__ g_env, float* __restrict__ g_spec,
                                                           float* __restrict__ g_bg) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float imp = pix[r * 9], photo = pix[r * 9 + 5];
  float g_imp = 0.f, g_photo = 0.f, ge[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float gc = g[r * 3 + c];
    const float diff = env[r * Ce + (Ce == 3 ? c : 0)] + imp;
    const float base = pix[r * 9 + 6 + c], sp = spec[r * 3 + c];
    float g_base, g_diff, g_sp;
    if (entangle) {
      g_base = gc * diff; g_diff = gc * base; g_photo += gc * sp; g_sp = gc * photo;
    } else {
      g_photo += gc * (base * diff + sp);
      g_base = gc * photo * diff; g_diff = gc * photo * base; g_sp = gc * photo;
    }
    g_pix[r * 9 + 6 + c] = g_base;
    g_spec[r * 3 + c] = g_sp;
    g_imp += g_diff;
    ge[c] = g_diff;
    if (g_bg) g_bg[r * 3 + c] = gc;
  }
  g_pix[r * 9] = g_imp;
  g_pix[r * 9 + 1] = 0.f; g_pix[r * 9 + 2] = 0.f; g_pix[r * 9 + 3] = 0.f; g_pix[r * 9 + 4] = 0.f;   // roughness / specular: not in the colour
  g_pix[r * 9 + 5] = g_photo;
  if (Ce == 3) { g_env[r * 3] = ge[0]; g_env[r * 3 + 1] = ge[1]; g_env[r * 3 + 2] = ge[2]; }
  else g_env[r] = ge[0] + ge[1] + ge[2];
}

// ---- loss terms -------------------------------------------------------------------------------------------------------
// Per-ray partial sums (one wave per ray), row layout of `partial` (R, LT_COLS):
//   0 rgb error sum   1 eikonal   2 TV (all TV tensors)   3..7 the five prior sums x mask   8 mask
constexpr int LT_COLS = 9;
// terms (device, LT_TERMS floats):
//   0 loss   1 loss_rgb   2 loss_eikonal   3 loss_tv   4 prior_base_color   5 prior_roughness   6 reg_std_roughness
//   7 prior_specular_reflectance   8 reg_std_specular_reflectance   9 1 / denorm   10 sum(mask)   11 1 / denorm of the priors
constexpr int LT_TERMS = 12;

struct LossWeights {
  float inv_rays;        // 1 / (B R ray_shards)
  float eikonal, tv, base_color, roughness, specular;   // train.*_weight (roughness / specular also weigh their reg terms)
  int l2;                // rgb_loss == "l2"
};

__global__ void __launch_bounds__(256) k_loss_partial(int R, int N, const float* __restrict__ color, const float* __restrict__ gt,
                                                      const float* __restrict__ mask, const float* __restrict__ grad_x,
                                                      const float* __restrict__ tv0, int D0, const float* __restrict__ tv1,
                                                      int D1, const float* __restrict__ prior, int l2,
                                                      float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float m = mask[r];
  float eik = 0.f, tv = 0.f;
  if (grad_x) {
    for (int i = lane; i < N; i += 64) {
      const float* gp = grad_x + ((long long)r * N + i) * 3;
      const float gn = sqrtf(gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2]);
      const float e = (gn - 1.f) * m;
      eik += e * e;
    }
  }
  if (tv0) {
    const float* p = tv0 + (long long)r * N * D0;
    for (int i = lane; i < N * D0; i += 64) tv += p[i];
  }
  if (tv1) {
    const float* p = tv1 + (long long)r * N * D1;
    for (int i = lane; i < N * D1; i += 64) tv += p[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { eik += __shfl_xor(eik, off); tv += __shfl_xor(tv, off); }
  if (lane == 0) {
    float rgb = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = color[r * 3 + c] - gt[r * 3 + c];
      rgb += l2 ? d * d : fabsf(d);
    }
    float* o = partial + (long long)r * LT_COLS;
    o[0] = rgb; o[1] = eik; o[2] = tv * m;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[3 + k] = prior ? prior[r * 5 + k] * m : 0.f;
    o[8] = m;
  }
}

// one workgroup: column sums of `partial` in a fixed order, normalisation (loss.py:74, 95, 118: every term but the RGB one is
// divided by sum(mask) N + 1e-5 -- sum(mask) over ALL ray shards when mask_sum_global is given), weighted total.
// The priors' N is its own argument: python/loss.py:36 binds N = n_samples0, :72 rebinds it to the sample count only inside
// `if eikonal_weight > 0`, and :118 divides the priors by whatever N is then -- with the eikonal term off that is n_samples0.
// A term whose weight is zero is not evaluated by the reference (python/loss.py:70, 85, 124, 135, 152): it is reported as 0
// and stays out of the total, so that a non-finite value of a switched-off term (0 * Inf) cannot poison the loss.
__global__ void __launch_bounds__(256) k_loss_finish(int R, int N, int Np, const float* __restrict__ partial,
                                                     const float* __restrict__ mask_sum_global, LossWeights w,
                                                     float* __restrict__ terms) {
  __shared__ float red[LT_COLS][256];
  float acc[LT_COLS];
#pragma unroll
  for (int c = 0; c < LT_COLS; ++c) acc[c] = 0.f;
  for (int r = threadIdx.x; r < R; r += 256)
#pragma unroll
    for (int c = 0; c < LT_COLS; ++c) acc[c] += partial[(long long)r * LT_COLS + c];
#pragma unroll
  for (int c = 0; c < LT_COLS; ++c) red[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
#pragma unroll
      for (int c = 0; c < LT_COLS; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float msum = mask_sum_global ? mask_sum_global[0] : red[8][0];
    const float denorm = msum * (float)N + 1e-5f;
    const float dprior = msum * (float)Np + 1e-5f;
    const float inv = 1.f / denorm;
    const float l_rgb = red[0][0] * w.inv_rays;
    const float l_eik = w.eikonal > 0.f ? red[1][0] / denorm : 0.f, l_tv = w.tv > 0.f ? red[2][0] / denorm : 0.f;
    const float p_bc = w.base_color > 0.f ? red[3][0] / dprior : 0.f;
    const float p_r = w.roughness > 0.f ? red[4][0] / dprior : 0.f, g_r = w.roughness > 0.f ? red[5][0] / dprior : 0.f;
    const float p_s = w.specular > 0.f ? red[6][0] / dprior : 0.f, g_s = w.specular > 0.f ? red[7][0] / dprior : 0.f;
    terms[1] = l_rgb; terms[2] = l_eik; terms[3] = l_tv; terms[4] = p_bc; terms[5] = p_r; terms[6] = g_r; terms[7] = p_s;
    terms[8] = g_s; terms[9] = inv; terms[10] = red[8][0]; terms[11] = 1.f / dprior;
    // python/loss.py:168-178 in its order of additions
    float total = l_rgb;
    if (w.eikonal > 0.f) total += w.eikonal * l_eik;
    if (w.tv > 0.f) total += w.tv * l_tv;
    if (w.base_color > 0.f) total += w.base_color * p_bc;
    if (w.roughness > 0.f) total += w.roughness * p_r;
    if (w.specular > 0.f) total += w.specular * p_s;
    if (w.roughness > 0.f) total += w.roughness * g_r;
    if (w.specular > 0.f) total += w.specular * g_s;
    terms[0] = total;
  }
}

// d loss / d (color, grad_x, tv tensors, prior sums) for upstream gradient g_loss (device scalar) of terms[0]
__global__ void __launch_bounds__(256) k_loss_bwd(int R, int N, const float* __restrict__ color, const float* __restrict__ gt,
                                                  const float* __restrict__ mask, const float* __restrict__ grad_x, int D0, int D1,
                                                  const float* __restrict__ terms, const float* __restrict__ g_loss, LossWeights w,
                                                  float* __restrict__ g_color, float* __restrict__ g_grad_x,
                                                  float* __restrict__ g_tv0, float* __restrict__ g_tv1,
                                                  float* __restrict__ g_prior) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const float g = g_loss[0], inv = terms[9], inv_prior = terms[11], m = mask[r];
  if (g_grad_x) {
    const float k = g * w.eikonal * inv * m * m;
    for (int i = lane; i < N; i += 64) {
      const float* gp = grad_x + ((long long)r * N + i) * 3;
      const float gn = sqrtf(gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2]);
      const float s = k * 2.f * (gn - 1.f) / gn;           // (0 / 0 at a vanishing gradient, like the chain rule through sqrt)
      float* o = g_grad_x + ((long long)r * N + i) * 3;
      o[0] = s * gp[0]; o[1] = s * gp[1]; o[2] = s * gp[2];
    }
  }
  const float gtv = g * w.tv * inv * m;
  if (g_tv0) { float* p = g_tv0 + (long long)r * N * D0; for (int i = lane; i < N * D0; i += 64) p[i] = gtv; }
  if (g_tv1) { float* p = g_tv1 + (long long)r * N * D1; for (int i = lane; i < N * D1; i += 64) p[i] = gtv; }
  if (lane < 3) {
    const float d = color[r * 3 + lane] - gt[r * 3 + lane];
    const float s = w.l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    g_color[r * 3 + lane] = g * w.inv_rays * s;
  }
  if (g_prior && lane < 5) {
    const float wk = lane == 0 ? w.base_color : (lane <= 2 ? w.roughness : w.specular);
    g_prior[r * 5 + lane] = g * wk * inv_prior * m;
  }
}

}  // namespace ndjir

using namespace ndjir;

extern "C" int ndjir_render_pixel_normal(int R, float eps, const float* grad_pixel, float* normal, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!grad_pixel || !normal) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_pixel_normal, dim3((R + 255) / 256), dim3(256), 0, stream, R, eps, grad_pixel, normal);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_pixel_normal_backward(int R, float eps, const float* grad_pixel, const float* g_normal, float* g_grad_pixel,
                                                  hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!grad_pixel || !g_normal || !g_grad_pixel) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_pixel_normal_bwd, dim3((R + 255) / 256), dim3(256), 0, stream, R, eps, grad_pixel, g_normal, g_grad_pixel);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_pixel_compose(int R, int Ce, int entangle, const float* pix, const float* env, const float* spec,
                                          const float* bg, float* color, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!pix || !env || !spec || !color) return NDJIR_ERR_ARG;
  if (Ce != 1 && Ce != 3) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pixel_compose, dim3((R + 255) / 256), dim3(256), 0, stream, R, Ce, entangle, pix, env, spec, bg, color);
  return ndjir_check_launch();
}

extern "C" int ndjir_render_pixel_compose_backward(int R, int Ce, int entangle, const float* pix, const float* env,
                                                   const float* spec, const float* g, float* g_pix, float* g_env, float* g_spec,
                                                   float* g_bg, hipStream_t stream) {
  if (R <= 0) return NDJIR_OK;
  if (!pix || !env || !spec || !g || !g_pix || !g_env || !g_spec) return NDJIR_ERR_ARG;
  if (Ce != 1 && Ce != 3) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_pixel_compose_bwd, dim3((R + 255) / 256), dim3(256), 0, stream, R, Ce, entangle, pix, env, spec, g, g_pix,
                     g_env, g_spec, g_bg);
  return ndjir_check_launch();
}

static LossWeights make_weights(float inv_rays, const float* weights5, int l2) {
  LossWeights w;
  w.inv_rays = inv_rays; w.eikonal = weights5[0]; w.tv = weights5[1]; w.base_color = weights5[2]; w.roughness = weights5[3];
  w.specular = weights5[4]; w.l2 = l2;
  return w;
}

extern "C" int ndjir_loss_terms_workspace(int R) { return R * LT_COLS; }

extern "C" int ndjir_loss_terms(int R, int N, int N_prior, const float* color, const float* color_gt, const float* mask, const float* grad_x,
                                const float* tv0, int D0, const float* tv1, int D1, const float* prior,
                                const float* mask_sum_global, float inv_rays, const float* weights5, int l2, float* workspace,
                                float* terms, hipStream_t stream) {
  if (R <= 0 || N <= 0 || N_prior <= 0) return NDJIR_ERR_ARG;
  if (!color || !color_gt || !mask || !weights5 || !workspace || !terms) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_loss_partial, dim3((R + 3) / 4), dim3(256), 0, stream, R, N, color, color_gt, mask, grad_x, tv0, D0, tv1, D1,
                     prior, l2, workspace);
  hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(256), 0, stream, R, N, N_prior, workspace, mask_sum_global,
                     make_weights(inv_rays, weights5, l2), terms);
  return ndjir_check_launch();
}

extern "C" int ndjir_loss_terms_backward(int R, int N, const float* color, const float* color_gt, const float* mask,
                                         const float* grad_x, int D0, int D1, const float* terms, const float* g_loss,
                                         float inv_rays, const float* weights5, int l2, float* g_color, float* g_grad_x,
                                         float* g_tv0, float* g_tv1, float* g_prior, hipStream_t stream) {
  if (R <= 0 || N <= 0) return NDJIR_ERR_ARG;
  if (!color || !color_gt || !mask || !terms || !g_loss || !weights5 || !g_color) return NDJIR_ERR_ARG;
  if (g_grad_x && !grad_x) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_loss_bwd, dim3((R + 3) / 4), dim3(256), 0, stream, R, N, color, color_gt, mask, grad_x, D0, D1, terms, g_loss,
                     make_weights(inv_rays, weights5, l2), g_color, g_grad_x, g_tv0, g_tv1, g_prior);
  return ndjir_check_launch();
}
