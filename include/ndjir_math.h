/*
 * ndjir_math.h -- arithmetic DEFINITIONS shared by the HIP kernels and the CPU oracle wherever the
 * result feeds an integer decision (the sampler's bin indices).  Everything here is built from
 * IEEE-754 single-precision +, *, /, fmaf, rintf, ldexpf only, in a fixed order, so the same
 * inputs give the same bits on gfx950 and on the host (translation units that include this header
 * are compiled with floating-point contraction off).
 *
 * Also the scan / reduction ORDERS of the hierarchical sampler (python/sampler.py:199-222): the
 * reference leaves them to nnabla's CUDA scan kernels (unspecified association); here they are fixed
 * as Kogge-Stone inclusive scans over the slots and a sum that first adds the slots of one residue
 * class mod 64, s_l = (w[l] + w[l+64]) + (w[l+128] + w[l+192]), then combines the 64 classes with an
 * xor butterfly (masks 32, 16, ..., 1) -- which a 64-lane wave and a plain C loop evaluate identically.
 * Slots beyond the ray's samples hold the identity (1 for the product scan, 0 for the sums), so the
 * results do not depend on how many slots an implementation carries (128 or 256).
 */
#ifndef NDJIR_MATH_H
#define NDJIR_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define NDJIR_HD __host__ __device__ static inline
#else
#define NDJIR_HD static inline
#endif

/* exp(x), |error| <= ~1 ulp (Cephes expf polynomial with Cody-Waite reduction) */
NDJIR_HD float ndjir_expf(float x) {
  if (x > 88.0f) x = 88.0f;
  if (x < -87.0f) x = -87.0f;
  float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(k, -0.693359375f, x);
  r = fmaf(k, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float e = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(e, (int)k);
}

NDJIR_HD float ndjir_sigmoidf(float x) { return 1.0f / (1.0f + ndjir_expf(-x)); }

#define NDJIR_SAMPLER_SLOTS 256   /* max samples per ray the sampler kernels handle */

#endif /* NDJIR_MATH_H */
