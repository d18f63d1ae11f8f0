// rays.hip -- camera rays on the device.
//
// Reference: python/helper.py:44-73 `generate_raydir_camloc` (numpy, float64, assigned to fp32 variables every
// iteration, python/train.py:131-133 -- a host computation plus two host->device copies per step) and the pixel
// bookkeeping of python/dataset.py:96-101 (`x = idx - y W`, `y = idx // W`).  One lane per ray:
//     x_w = R_c2w K^-1 (x, y, 1)^T ;  raydir = x_w / |x_w| ;  camloc = pose[:3, 3]
// Arithmetic is done in double like the reference's and rounded once to fp32; K^-1 by the adjugate (numpy's LU
// inverse agrees to a few ulp of double, far below the fp32 rounding of the result).
#include <hip/hip_runtime.h>

#include "common.h"

#pragma clang fp contract(off)

namespace ndjir {

__global__ void __launch_bounds__(256) k_generate_rays(int B, int R, const double* __restrict__ pose,
                                                       const double* __restrict__ intrinsic, const int* __restrict__ pixel_index,
                                                       const float* __restrict__ xy, int W, float* __restrict__ raydir,
                                                       float* __restrict__ camloc) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= B * R) return;
  int b = n / R;
  const double* K = intrinsic + b * 9;
  const double* P = pose + b * 16;
  double a = K[0], bb = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
  double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + bb * Bc + c * C;
  double inv[9] = {A / det, -(bb * i - c * h) / det, (bb * f - c * e) / det,
                   Bc / det, (a * i - c * g) / det, -(a * f - c * d) / det,
                   C / det, -(a * h - bb * g) / det, (a * e - bb * d) / det};
  double px, py;
  if (pixel_index) {
    int idx = pixel_index[n];
    int y = idx / W;
    px = (double)(idx - y * W);
    py = (double)y;
  } else {
    px = (double)xy[n * 2];
    py = (double)xy[n * 2 + 1];
  }
  double cx = inv[0] * px + inv[1] * py + inv[2];
  double cy = inv[3] * px + inv[4] * py + inv[5];
  double cz = inv[6] * px + inv[7] * py + inv[8];
  double wx = P[0] * cx + P[1] * cy + P[2] * cz;
  double wy = P[4] * cx + P[5] * cy + P[6] * cz;
  double wz = P[8] * cx + P[9] * cy + P[10] * cz;
  double nrm = sqrt(wx * wx + wy * wy + wz * wz);
  raydir[n * 3] = (float)(wx / nrm);
  raydir[n * 3 + 1] = (float)(wy / nrm);
  raydir[n * 3 + 2] = (float)(wz / nrm);
  if (n - b * R == 0) {
    camloc[b * 3] = (float)P[3];
    camloc[b * 3 + 1] = (float)P[7];
    camloc[b * 3 + 2] = (float)P[11];
  }
}

}  // namespace ndjir

// pose (B, 4, 4) camera-to-world and intrinsic (B, 3, 3), both double (the reference keeps them in float64 numpy);
// pixels either as flat indices pixel_index (B, R) int32 into a W-wide image (python/dataset.py:96-101) or as
// coordinates xy (B, R, 2) float32 -- exactly one of the two non-null.  Outputs raydir (B, R, 3), camloc (B, 3) fp32.
extern "C" int ndjir_generate_raydir_camloc(int B, int R, const double* pose, const double* intrinsic, const int* pixel_index,
                                            const float* xy, int W, float* raydir, float* camloc, hipStream_t stream) {
  if (B <= 0 || R <= 0) return NDJIR_OK;
  if (!pose || !intrinsic || !raydir || !camloc) return NDJIR_ERR_ARG;
  if ((pixel_index == nullptr) == (xy == nullptr)) return NDJIR_ERR_ARG;
  if (pixel_index && W <= 0) return NDJIR_ERR_ARG;
  int n = B * R;
  hipLaunchKernelGGL(ndjir::k_generate_rays, dim3((n + 255) / 256), dim3(256), 0, stream, B, R, pose, intrinsic, pixel_index, xy, W,
                     raydir, camloc);
  return ndjir::ndjir_check_launch();
}
