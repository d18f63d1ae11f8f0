// misc.hip -- ray/box and ray/sphere intersection, hemisphere light-direction sampling,
// squareplus.  One lane per ray (intersection) / per (ray, light) sample.  All trivially
// parallel and HBM-streaming; arithmetic follows the cited reference kernels.
#include <hip/hip_runtime.h>

#include "common.h"
#include "grid.h"

// scalar geometry kernels: keep the reference's unfused expression order (bit-comparable with the
// CPU oracle apart from libm calls)
#pragma clang fp contract(off)

namespace ndjir {

static inline int blocks_for(long long n) {
  long long b = (n + 255) / 256;
  if (b > 256LL * 16) b = 256LL * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// csrc/intersection/ray_aabb_intersection_cuda.cu:27-142
__global__ void __launch_bounds__(256) k_ray_aabb(int N, float* __restrict__ t_near, float* __restrict__ t_far,
                                                  float* __restrict__ n_hits, const float* __restrict__ camloc,
                                                  const float* __restrict__ raydir, int R, float3 mn, float3 mx) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    int b = n / R;
    float cx = camloc[b * 3], cy = camloc[b * 3 + 1], cz = camloc[b * 3 + 2];
    float dx = raydir[n * 3], dy = raydir[n * 3 + 1], dz = raydir[n * 3 + 2];
    float ix = 1.f / dx, iy = 1.f / dy, iz = 1.f / dz;
    float t[6] = {(mx.x - cx) * ix, (mx.y - cy) * iy, (mx.z - cz) * iz,
                  (mn.x - cx) * ix, (mn.y - cy) * iy, (mn.z - cz) * iz};
    int hits = 0, i0 = 0, i1 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float ti = t[i];
      float x = cx + ti * dx, y = cy + ti * dy, z = cz + ti * dz;
      // snap the coordinate of the hit plane (:60-66)
      if (i == 0) x = mx.x; else if (i == 1) y = mx.y; else if (i == 2) z = mx.z;
      else if (i == 3) x = mn.x; else if (i == 4) y = mn.y; else z = mn.z;
      bool ok = !isinf(ti) && (ti >= 0.f) && (x >= mn.x) && (x <= mx.x) && (y >= mn.y) && (y <= mx.y) &&
                (z >= mn.z) && (z <= mx.z);
      if (ok) {
        if (hits == 0) i0 = i; else i1 = i;
        hits++;
      }
    }
    float tn = 0.f, tf = 0.f;
    if (hits >= 2) {
      float a = t[0], c = t[0];
#pragma unroll
      for (int i = 0; i < 6; ++i) { if (i == i0) a = t[i]; if (i == i1) c = t[i]; }
      if (a <= c) { tn = a; tf = c; } else { tn = c; tf = a; }
    } else if (hits == 1) {
      float a = t[0];
#pragma unroll
      for (int i = 0; i < 6; ++i) if (i == i0) a = t[i];
      tf = a;
    }
    n_hits[n] = (float)hits;
    t_near[n] = tn;
    t_far[n] = tf;
  }
}

// csrc/intersection/ray_sphere_intersection_cuda.cu:26-78
__global__ void __launch_bounds__(256) k_ray_sphere(int N, float* __restrict__ t_near, float* __restrict__ t_far,
                                                    float* __restrict__ n_hits, const float* __restrict__ camloc,
                                                    const float* __restrict__ raydir, int R, float radius) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    int b = n / R;
    float cx = camloc[b * 3], cy = camloc[b * 3 + 1], cz = camloc[b * 3 + 2];
    float vx = raydir[n * 3], vy = raydir[n * 3 + 1], vz = raydir[n * 3 + 2];
    float r2 = radius * radius;
    float cv = cx * vx + cy * vy + cz * vz;
    float vv = vx * vx + vy * vy + vz * vz;
    float cc = cx * cx + cy * cy + cz * cz;
    float X = -cv, Y = cv * cv - vv * (cc - r2), Zi = 1.f / vv;
    int hits = 0;
    float tn = 0.f, tf = 0.f;
    if (Y > 0) {
      float Ys = sqrtf(Y);
      tn = (X - Ys) * Zi;
      tf = (X + Ys) * Zi;
      int pos = (tn >= 0);
      tn = pos * tn;
      hits = 2 - (1 - pos);
    } else if (Y == 0) {
      hits = 1;
      tn = X * Zi;
      tf = X * Zi;
    }
    n_hits[n] = (float)hits;
    t_near[n] = tn;
    t_far[n] = tf;
  }
}

// csrc/sampling/inverse_transform_cuda.cu:30-69 (uniform), :93-136 (GGX importance)
template <bool IMPORTANCE>
__global__ void __launch_bounds__(256) k_sample_dirs(int size, float* __restrict__ light_dirs,
                                                     const float* __restrict__ normal, const float* __restrict__ cdf_the,
                                                     const float* __restrict__ cdf_phi, const float* __restrict__ alpha,
                                                     int n_lights, int n_thes, int n_phis, float eps) {
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x) {
    int b = s / n_lights, m = s - b * n_lights;
    int m_the = m / n_phis, m_phi = m - m_the * n_phis;
    float ct = cdf_the[b * n_thes + m_the];
    float cp = cdf_phi[b * n_phis + m_phi];
    double phi = 2.f * M_PI * cp;  // evaluated in double in the reference (:43)
    float cos_the = ct;
    if constexpr (IMPORTANCE) {
      float a = alpha[b], a2 = a * a;
      cos_the = sqrtf((1.f - ct) / ((a2 - 1.f) * ct + 1.f));
    }
    float sin_the = sqrtf(1.f - cos_the * cos_the);
    float x = sin_the * cosf((float)phi);
    float y = sin_the * sinf((float)phi);
    float z = cos_the;
    float nx = normal[b * 3] + eps, ny = normal[b * 3 + 1] + eps, nz = normal[b * 3 + 2] + eps;
    float il = 1.0f / sqrtf(nx * nx + ny * ny + nz * nz);   // helper_math.h:1325-1329 (rsqrtf)
    float zx = nx * il, zy = ny * il, zz = nz * il;
    float ix = 1.0f / sqrtf(ny * ny + nx * nx + 0.f);
    float xx = -ny * ix, xy = nx * ix, xz = 0.f * ix;
    float yx = zy * xz - zz * xy, yy = zz * xx - zx * xz, yz = zx * xy - zy * xx;
    light_dirs[s * 3] = x * xx + y * yx + z * zx;
    light_dirs[s * 3 + 1] = x * xy + y * yy + z * zy;
    light_dirs[s * 3 + 2] = x * xz + y * yz + z * zz;
  }
}

// csrc/activation/squareplus_cuda.cu:29-59
__global__ void __launch_bounds__(256) k_squareplus_fwd(int n, float* __restrict__ y, const float* __restrict__ x, float b) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    y[i] = 0.5f * (x[i] + sqrtf(x[i] * x[i] + b));
}
template <bool ACCUM>
__global__ void __launch_bounds__(256) k_squareplus_bwd(int n, float* __restrict__ dx, const float* __restrict__ dy,
                                                        const float* __restrict__ x, float b) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v = dy[i] * 0.5f * (1.f + x[i] * (1.0f / sqrtf(x[i] * x[i] + b)));
    dx[i] = ACCUM ? dx[i] + v : v;
  }
}

int launch_ray_aabb(int N, float* tn, float* tf, float* nh, const float* camloc, const float* raydir, int R,
                    const float* mn, const float* mx, hipStream_t stream) {
  if (N <= 0) return NDJIR_OK;
  hipLaunchKernelGGL(k_ray_aabb, dim3(blocks_for(N)), dim3(256), 0, stream, N, tn, tf, nh, camloc, raydir, R,
                     make_float3(mn[0], mn[1], mn[2]), make_float3(mx[0], mx[1], mx[2]));
  return ndjir_check_launch();
}

int launch_ray_sphere(int N, float* tn, float* tf, float* nh, const float* camloc, const float* raydir, int R,
                      float radius, hipStream_t stream) {
  if (N <= 0) return NDJIR_OK;
  hipLaunchKernelGGL(k_ray_sphere, dim3(blocks_for(N)), dim3(256), 0, stream, N, tn, tf, nh, camloc, raydir, R, radius);
  return ndjir_check_launch();
}

int launch_sample_dirs(int size, float* light_dirs, const float* normal, const float* cdf_the, const float* cdf_phi,
                       const float* alpha, int n_lights, int n_thes, int n_phis, float eps, hipStream_t stream) {
  if (size <= 0) return NDJIR_OK;
  if (alpha)
    hipLaunchKernelGGL((k_sample_dirs<true>), dim3(blocks_for(size)), dim3(256), 0, stream, size, light_dirs, normal,
                       cdf_the, cdf_phi, alpha, n_lights, n_thes, n_phis, eps);
  else
    hipLaunchKernelGGL((k_sample_dirs<false>), dim3(blocks_for(size)), dim3(256), 0, stream, size, light_dirs, normal,
                       cdf_the, cdf_phi, alpha, n_lights, n_thes, n_phis, eps);
  return ndjir_check_launch();
}

int launch_squareplus(int n, bool bwd, float* out, const float* dy, const float* x, float b, bool accum,
                      hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!bwd) hipLaunchKernelGGL(k_squareplus_fwd, dim3(blocks_for(n)), dim3(256), 0, stream, n, out, x, b);
  else if (accum) hipLaunchKernelGGL((k_squareplus_bwd<true>), dim3(blocks_for(n)), dim3(256), 0, stream, n, out, dy, x, b);
  else hipLaunchKernelGGL((k_squareplus_bwd<false>), dim3(blocks_for(n)), dim3(256), 0, stream, n, out, dy, x, b);
  return ndjir_check_launch();
}

}  // namespace ndjir
