// geo.hip -- the element-wise work around the geometric network's fused chains (ndjir_amd/geometric.py): building the
// chain input e = [x, cos(x 2^k), sin(x 2^k), grid features] (python/network.py:96-117, 154-170), turning the sdf chain's
// input gradient g_0 into the normal n = J_e(x)^T g_0 (`nn.grad([sdf], [x])`, python/renderer.py:52), and the matching
// pieces of the hand-derived double backward.  The reference spells each as a handful of nnabla functions; here each is
// one launch, and the outputs are laid out so that the per-sample material nets read them without a concatenation:
//   Z (P, ldz) = [ x (3) | feature (D) | n (3) | spare columns ]      (the input cat(x, feature, normal) of
//                                                                       python/network.py:235-263, 300-336, 380-509)
// The forward chain stores its output y = [sdf | feature] at Z + 2 (row stride ldz), so the feature columns are already in
// place; k_geo_normal moves the sdf out of column 2 before x goes there.
#include <hip/hip_runtime.h>

#include "common.h"

namespace ndjir {

constexpr int GEO_SEGS = 4;
struct GeoSegs {
  const float* p[GEO_SEGS];     // (P, C[i]) row-major, contiguous
  int C[GEO_SEGS];
  int n;
};

// e[p] = [x, cos(x_d 2^k) (d major, k fastest), sin(...), seg_0[p], seg_1[p], ...]
__global__ void __launch_bounds__(256) k_geo_encode(long long P, int M, const float* __restrict__ x, GeoSegs s,
                                                    float* __restrict__ e, int lde, int W) {
  const int npe = 3 + 6 * M;
  const long long total = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    int c = rc.c;
    float v;
    if (c < 3) v = x[p * 3 + c];
    else if (c < npe) {
      c -= 3;
      const bool is_sin = c >= 3 * M;
      if (is_sin) c -= 3 * M;
      const float b = x[p * 3 + c / M] * (float)(1 << (c % M));
      v = is_sin ? sinf(b) : cosf(b);
      c = (int)(t - p * W);
    } else {
      int cc = c - npe;
      v = 0.f;
#pragma unroll
      for (int i = 0; i < GEO_SEGS; ++i) {
        if (i < s.n) {
          if (cc >= 0 && cc < s.C[i]) v = s.p[i][p * s.C[i] + cc];
          cc -= s.C[i];
        }
      }
    }
    e[p * lde + c] = v;
  }
}

// n[p][d] = g0[p][d] + sum_k 2^k (g_sin[d][k] cos(b) - g_cos[d][k] sin(b)) + sum_i gq_i[p][d]   (cos / sin read from e)
// With Z: sdf_out[p] = Z[p][2] (where the forward chain left it), then Z[p] = [x | (feature, untouched) | n | 0 ...].
__global__ void __launch_bounds__(256) k_geo_normal(long long P, int M, const float* __restrict__ e, int lde,
                                                    const float* __restrict__ g0, int ldg, GeoSegs gq, float* __restrict__ n_out,
                                                    float* __restrict__ Z, int ldz, int D, float* __restrict__ sdf_out) {
  const long long total = P * 3;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long p = t / 3;
    const int d = (int)(t - p * 3);
    const float* er = e + p * lde;
    const float* gr = g0 + p * ldg;
    float acc = 0.f;
    for (int k = 0; k < M; ++k) {
      const float cosv = er[3 + d * M + k], sinv = er[3 + 3 * M + d * M + k];
      acc += (gr[3 + 3 * M + d * M + k] * cosv - gr[3 + d * M + k] * sinv) * (float)(1 << k);
    }
    acc += gr[d];
#pragma unroll
    for (int i = 0; i < GEO_SEGS; ++i)
      if (i < gq.n) acc += gq.p[i][p * 3 + d];
    if (n_out) n_out[t] = acc;
    if (Z) {
      float* zr = Z + p * ldz;
      if (d == 2) sdf_out[p] = zr[2];
      zr[d] = er[d];
      zr[3 + D + d] = acc;
      if (d == 0)
        for (int c = 6 + D; c < ldz; ++c) zr[c] = 0.f;
    }
  }
}

// gy[p] = [g_sdf | g_feat + gZ[:, 3:3+D]],  nbar[p] = g_n + gZ[:, 3+D:6+D]   (absent terms are zero)
__global__ void __launch_bounds__(256) k_geo_bwd_begin(long long P, int D, const float* __restrict__ g_sdf,
                                                       const float* __restrict__ g_feat, int ldf, const float* __restrict__ g_n,
                                                       const float* __restrict__ gZ, int ldz, float* __restrict__ gy,
                                                       float* __restrict__ nbar) {
  const int W = 1 + D + 3;
  const long long total = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    if (c == 0) gy[p * (1 + D)] = g_sdf ? g_sdf[p] : 0.f;
    else if (c <= D) {
      float v = g_feat ? g_feat[p * ldf + c - 1] : 0.f;
      if (gZ) v += gZ[p * ldz + 2 + c];
      gy[p * (1 + D) + c] = v;
    } else if (nbar) {
      const int d = c - 1 - D;
      float v = g_n ? g_n[p * 3 + d] : 0.f;
      if (gZ) v += gZ[p * ldz + 3 + D + d];
      nbar[p * 3 + d] = v;
    }
  }
}

// The same with D % 4 == 0 (the 256 feature channels): one lane per 16 bytes of the feature block -- the rows of gy (1 + D floats)
// and of gZ start at odd float offsets, so the vectors are 4-byte aligned (correct on gfx950, within 10 % of the aligned rate:
// tools/ubench/unaligned.hip) -- plus one lane for the sdf column and three for n-bar: 136 MB moved in ~35 instead of 64 us.
typedef float geo_f4 __attribute__((ext_vector_type(4)));
typedef geo_f4 geo_f4u __attribute__((aligned(4)));
__global__ void __launch_bounds__(256) k_geo_bwd_begin_v4(long long P, int D, const float* __restrict__ g_sdf,
                                                          const float* __restrict__ g_feat, int ldf, const float* __restrict__ g_n,
                                                          const float* __restrict__ gZ, int ldz, float* __restrict__ gy,
                                                          float* __restrict__ nbar) {
  const int V = D >> 2, W = V + 4;                 // work items per row: V vectors, the sdf column, three n-bar components
  const long long total = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    if (c < V) {
      geo_f4 v = {0.f, 0.f, 0.f, 0.f};
      if (g_feat) v = *reinterpret_cast<const geo_f4u*>(g_feat + p * ldf + 4 * c);
      if (gZ) {
        const geo_f4 z = *reinterpret_cast<const geo_f4u*>(gZ + p * ldz + 3 + 4 * c);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += z[q];
      }
      *reinterpret_cast<geo_f4u*>(gy + p * (1 + D) + 1 + 4 * c) = v;
    } else if (c == V) gy[p * (1 + D)] = g_sdf ? g_sdf[p] : 0.f;
    else if (nbar) {
      const int d = c - V - 1;
      float v = g_n ? g_n[p * 3 + d] : 0.f;
      if (gZ) v += gZ[p * ldz + 3 + D + d];
      nbar[p * 3 + d] = v;
    }
  }
}

// g-bar_0 = J_e(x) n-bar: [nbar | -sin(b) nbar_d 2^k | cos(b) nbar_d 2^k | ggo_0 | ggo_1 ...]
__global__ void __launch_bounds__(256) k_geo_gbar0(long long P, int M, const float* __restrict__ e, int lde,
                                                   const float* __restrict__ nbar, GeoSegs s, float* __restrict__ gb0, int W) {
  const int npe = 3 + 6 * M;
  const long long total = P * W;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, W);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    float v;
    if (c < 3) v = nbar[p * 3 + c];
    else if (c < npe) {
      int cc = c - 3;
      const bool is_sin = cc >= 3 * M;                    // the sin columns of e carry +cos, the cos columns -sin
      if (is_sin) cc -= 3 * M;
      const int d = cc / M, k = cc % M;
      const float nb = nbar[p * 3 + d] * (float)(1 << k);
      v = is_sin ? e[p * lde + 3 + d * M + k] * nb : -e[p * lde + 3 + 3 * M + d * M + k] * nb;
    } else {
      int cc = c - npe;
      v = 0.f;
#pragma unroll
      for (int i = 0; i < GEO_SEGS; ++i) {
        if (i < s.n) {
          if (cc >= 0 && cc < s.C[i]) v = s.p[i][p * s.C[i] + cc];
          cc -= s.C[i];
        }
      }
    }
    gb0[p * W + c] = v;
  }
}

// dst[p][0:C] = src[p][0:C] for row-strided matrices (a column slice made contiguous, or placed into a wider row)
__global__ void __launch_bounds__(256) k_copy_cols(long long P, int C, const float* __restrict__ src, int lds,
                                                   float* __restrict__ dst, int ldd) {
  const long long total = P * C;
  RowCol rc((long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256, C);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256, rc.next()) {
    const long long p = rc.p;
    const int c = rc.c;
    dst[p * ldd + c] = src[p * lds + c];
  }
}

// out[p] = 1 / (|x_p - camloc_b|^2 + 1e-5), the inverse squared distance input of the photogrammetric light net
// (python/network.py:405-409: `F.norm(...)**2`, i.e. the square of the rounded root)
__global__ void __launch_bounds__(256) k_inv_distance(long long P, long long rows_per_batch, const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ camloc, float* __restrict__ out, int ldo) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long long)gridDim.x * 256) {
    const float* c = camloc + (p / rows_per_batch) * 3;
    const float dx = x[p * ldx] - c[0], dy = x[p * ldx + 1] - c[1], dz = x[p * ldx + 2] - c[2];
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
    out[p * ldo] = 1.f / (nrm * nrm + 1e-5f);
  }
}

static inline unsigned geo_blocks(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));      // (8 workgroups per CU x 4 rounds; a thread then visits several elements)
}

static inline int geo_segs(GeoSegs& s, int n, const float* const* p, const int* C) {
  if (n < 0 || n > GEO_SEGS || (n > 0 && (!p || !C))) return -1;
  s.n = n;
  int w = 0;
  for (int i = 0; i < GEO_SEGS; ++i) {
    s.p[i] = i < n ? p[i] : nullptr;
    s.C[i] = i < n ? C[i] : 0;
    if (i < n && (!p[i] || C[i] <= 0)) return -1;
    w += s.C[i];
  }
  return w;
}

}  // namespace ndjir

using namespace ndjir;

extern "C" int ndjir_geo_encode(long long P, int M, const float* x, int nseg, const float* const* seg, const int* segC, float* e,
                                int lde, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  GeoSegs s;
  const int w = geo_segs(s, nseg, seg, segC);
  if (w < 0 || M < 0 || M > 30 || !x || !e) return NDJIR_ERR_ARG;
  const int W = 3 + 6 * M + w;
  if (lde < W) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_geo_encode, dim3(geo_blocks(P * W)), dim3(256), 0, stream, P, M, x, s, e, lde, W);
  return ndjir_check_launch();
}

extern "C" int ndjir_geo_normal(long long P, int M, const float* e, int lde, const float* g0, int ldg, int nseg,
                                const float* const* gq, float* n_out, float* Z, int ldz, int D, float* sdf_out,
                                hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  GeoSegs s;
  const int three[GEO_SEGS] = {3, 3, 3, 3};
  if (geo_segs(s, nseg, gq, three) < 0 || M < 0 || M > 30 || !e || !g0 || lde < 3 + 6 * M || ldg < 3 + 6 * M) return NDJIR_ERR_ARG;
  if (Z && (!sdf_out || D < 0 || ldz < 6 + D)) return NDJIR_ERR_ARG;
  if (!Z && !n_out) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_geo_normal, dim3(geo_blocks(P * 3)), dim3(256), 0, stream, P, M, e, lde, g0, ldg, s, n_out, Z, ldz, D, sdf_out);
  return ndjir_check_launch();
}

extern "C" int ndjir_geo_backward_begin(long long P, int D, const float* g_sdf, const float* g_feat, int ldf, const float* g_n,
                                        const float* gZ, int ldz, float* gy, float* nbar, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (D < 0 || !gy || (g_feat && ldf < D) || (gZ && ldz < 6 + D)) return NDJIR_ERR_ARG;
  if (D > 0 && (D & 3) == 0)
    hipLaunchKernelGGL(k_geo_bwd_begin_v4, dim3(geo_blocks(P * (4 + D / 4))), dim3(256), 0, stream, P, D, g_sdf, g_feat, ldf, g_n, gZ,
                       ldz, gy, nbar);
  else
    hipLaunchKernelGGL(k_geo_bwd_begin, dim3(geo_blocks(P * (4 + D))), dim3(256), 0, stream, P, D, g_sdf, g_feat, ldf, g_n, gZ, ldz,
                       gy, nbar);
  return ndjir_check_launch();
}

extern "C" int ndjir_geo_gbar0(long long P, int M, const float* e, int lde, const float* nbar, int nseg, const float* const* seg,
                               const int* segC, float* gb0, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  GeoSegs s;
  const int w = geo_segs(s, nseg, seg, segC);
  if (w < 0 || M < 0 || M > 30 || !e || !nbar || !gb0 || lde < 3 + 6 * M) return NDJIR_ERR_ARG;
  const int W = 3 + 6 * M + w;
  hipLaunchKernelGGL(k_geo_gbar0, dim3(geo_blocks(P * W)), dim3(256), 0, stream, P, M, e, lde, nbar, s, gb0, W);
  return ndjir_check_launch();
}

extern "C" int ndjir_copy_columns(long long P, int C, const float* src, int lds, float* dst, int ldd, hipStream_t stream) {
  if (P <= 0 || C <= 0) return NDJIR_OK;
  if (!src || !dst || lds < C || ldd < C) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_copy_cols, dim3(geo_blocks(P * C)), dim3(256), 0, stream, P, C, src, lds, dst, ldd);
  return ndjir_check_launch();
}

extern "C" int ndjir_inverse_squared_distance(long long P, long long rows_per_batch, const float* x, int ldx, const float* camloc,
                                              float* out, int ldo, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (rows_per_batch <= 0 || !x || !camloc || !out || ldx < 3 || ldo < 1) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_inv_distance, dim3(geo_blocks(P)), dim3(256), 0, stream, P, rows_per_batch, x, ldx, camloc, out, ldo);
  return ndjir_check_launch();
}
