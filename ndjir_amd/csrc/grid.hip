// grid.hip -- grid-feature interpolation kernels for gfx950 (MI355X), written from scratch.
//
// What the reference does (csrc/grid_feature/*.cu): one CUDA thread per (point, channel), eight
// scalar gathers, one scalar atomicAdd per (point, channel, corner).  Here: one lane per
// (sub-grid, point) handles ALL channels with 16-byte (float4) / 8-byte corner loads, so a
// D=4 voxel corner is one dwordx4 load and the 8 corners of a cell share 4 cache lines along z;
// sub-grid-major lane order keeps adjacent lanes on adjacent ray samples (coherent cells).
// Scatter kernels use hardware fp32 atomics (global_atomic_add_f32, -munsafe-fp-atomics).
//
// One templated stencil covers every family:
//   topology  VOXEL (G,G,G,D) | TRIPLANE (3,G,G,D) | TRILINE (3,G,D) | HASH (levels of (T,D))
//   interp    LINEAR | COSINE (2 taps/axis) | LANCZOS a=2 (4 taps/axis)
// Arithmetic definitions follow the reference kernels (file:line cited at each piece).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "grid.h"

namespace ndjir {

template <int I> struct NTaps { static constexpr int v = (I == LANCZOS) ? 4 : 2; };

// --- per-axis taps ---------------------------------------------------------------------------
__device__ __forceinline__ float sincf_(float x) { return x == 0.f ? 1.0f : sinf(x) / x; }

// common.cuh:54-69 (z computed in double, rounded to float before sinf)
__device__ __forceinline__ float lanczos2(float x) {
  double z = M_PI * (double)x;
  return sincf_((float)z) * sincf_((float)(z / 2));
}
// common.cuh:82-97
__device__ __forceinline__ float lanczos2_grad(float x) {
  if (x == 0.f) return 0.0f;
  double z0 = M_PI * (double)x, z1 = M_PI * (double)x / 2;
  float s0 = sincf_((float)z0), s1 = sincf_((float)z1);
  float t0 = (cosf((float)z0) - s0) * s1;
  float t1 = (cosf((float)z1) - s1) * s0;
  return (t0 + t1) / x;
}

template <int I>
struct AxisTaps {
  static constexpr int NT = NTaps<I>::v;
  unsigned idx[NT];
  float w[NT];    // interpolation coefficient
  float dw[NT];   // d(coefficient)/d(continuous coordinate), up to the factor gm
  float gm;       // 1 (linear, lanczos) or 0.5*pi*sin(pi*frac) (cosine)
};

// x: continuous grid coordinate, G1 = G - 1
template <int I>
__device__ __forceinline__ void axis_taps(AxisTaps<I>& t, float x, float G1) {
  if constexpr (I == LANCZOS) {
    // lanczos_voxel_feature_cuda.cu:57-70: floor is NOT clamped, each tap index is
    float x0 = floorf(x);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float xi = fminf(fmaxf(x0 + (float)(i - 1), 0.f), G1);
      t.idx[i] = (unsigned)xi;
      t.w[i] = lanczos2(x - xi);
      t.dw[i] = lanczos2_grad(x - xi);
    }
    t.gm = 1.f;
  } else {
    // voxel_feature_cuda.cu:56-64
    float x0 = fminf(fmaxf(floorf(x), 0.f), G1);
    float x1 = fminf(x0 + 1.f, G1);
    t.idx[0] = (unsigned)x0;
    t.idx[1] = (unsigned)x1;
    if constexpr (I == LINEAR) {
      t.w[0] = x1 - x;
      t.gm = 1.f;
    } else {
      // cosine_voxel_feature_cuda.cu:65-66, 157
      const float pi = (float)M_PI;
      float fr = x - x0;
      t.w[0] = 0.5f * cosf(pi * fr) + 0.5f;
      t.gm = (float)(0.5 * M_PI) * sinf(pi * fr);
    }
    t.w[1] = 1.f - t.w[0];
    t.dw[0] = -1.f;
    t.dw[1] = 1.f;
  }
}

// voxel_hash_feature_cuda.cu:37-48 (tiny-cuda-nn primes)
__device__ __forceinline__ unsigned hash3(unsigned x, unsigned y, unsigned z, int T) {
  unsigned r = 0;
  r ^= x * 1u;
  r ^= y * 2654435761u;
  r ^= z * 805459861u;
  return r % (unsigned)T;
}

// --- stencil of one (point, sub-grid) -----------------------------------------------------------
template <int TOPO> struct NDims { static constexpr int v = (TOPO == TRIPLANE) ? 2 : (TOPO == TRILINE ? 1 : 3); };

template <int TOPO, int I>
struct Stencil {
  static constexpr int ND = NDims<TOPO>::v;
  static constexpr int NT = NTaps<I>::v;
  AxisTaps<I> ax[ND];
  float scale[ND];       // (G-1)/(max-min) of the axis
  int axis[ND];          // which query component (0,1,2) each stencil axis follows
  long long base;        // float offset of the sub-grid inside the feature tensor
  unsigned stride[ND];   // dense strides in floats (incl. D)
  int T;                 // hash table size of the level
  int D;
};

template <int TOPO, int I>
__device__ __forceinline__ void make_stencil(Stencil<TOPO, I>& st, const GridDesc& g, int s, const float* __restrict__ q) {
  constexpr int ND = NDims<TOPO>::v;
  st.D = g.D;
  st.T = 0;
  if constexpr (TOPO == VOXEL) {
    st.base = 0;
    st.stride[0] = (unsigned)(g.G[1] * g.G[2] * g.D);
    st.stride[1] = (unsigned)(g.G[2] * g.D);
    st.stride[2] = (unsigned)g.D;
#pragma unroll
    for (int a = 0; a < 3; ++a) st.axis[a] = a;
  } else if constexpr (TOPO == TRIPLANE) {
    // common_triplane.cuh:23-33: plane 0 = (x,y), 1 = (y,z), 2 = (z,x)
    st.base = (long long)s * g.G[0] * g.G[0] * g.D;
    st.stride[0] = (unsigned)(g.G[0] * g.D);
    st.stride[1] = (unsigned)g.D;
    st.axis[0] = s;
    st.axis[1] = (s + 1) % 3;
  } else if constexpr (TOPO == TRILINE) {
    st.base = (long long)s * g.G[0] * g.D;
    st.stride[0] = (unsigned)g.D;
    st.axis[0] = s;
  } else {
    st.base = g.lvlOff[s];
    st.T = g.lvlT[s];
#pragma unroll
    for (int a = 0; a < 3; ++a) { st.axis[a] = a; st.stride[a] = 0; }
  }
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    int c = st.axis[a];
    float G1 = (TOPO == HASH) ? (float)g.lvlG[s] - 1.f : (float)g.G[(TOPO == VOXEL) ? a : 0] - 1.f;
    float sc = G1 / (g.mx[c] - g.mn[c]);
    st.scale[a] = sc;
    axis_taps<I>(st.ax[a], (q[c] - g.mn[c]) * sc, G1);
  }
}

template <int TOPO, int I>
__device__ __forceinline__ long long cell_offset(const Stencil<TOPO, I>& st, int i, int j, int k) {
  constexpr int ND = NDims<TOPO>::v;
  if constexpr (TOPO == HASH) {
    return st.base + (long long)hash3(st.ax[0].idx[i], st.ax[1].idx[j], st.ax[2].idx[k], st.T) * st.D;
  } else {
    unsigned o = st.ax[0].idx[i] * st.stride[0];
    if constexpr (ND > 1) o += st.ax[1].idx[j] * st.stride[1];
    if constexpr (ND > 2) o += st.ax[2].idx[k] * st.stride[2];
    return st.base + (long long)o;
  }
}

// loop helper over all tap combinations of the stencil
#define NDJIR_FOR_TAPS(ND, NT)                          \
  _Pragma("unroll") for (int i = 0; i < NT; ++i)        \
  _Pragma("unroll") for (int j = 0; j < ((ND) > 1 ? NT : 1); ++j) \
  _Pragma("unroll") for (int k = 0; k < ((ND) > 2 ? NT : 1); ++k)

template <int TOPO, int I>
__device__ __forceinline__ float tap_w(const Stencil<TOPO, I>& st, int i, int j, int k) {
  constexpr int ND = NDims<TOPO>::v;
  float w = st.ax[0].w[i];
  if constexpr (ND > 1) w = w * st.ax[1].w[j];
  if constexpr (ND > 2) w = w * st.ax[2].w[k];
  return w;
}

// derivative weight of the tap along stencil axis a (before scale*gm)
template <int TOPO, int I>
__device__ __forceinline__ float tap_dw(const Stencil<TOPO, I>& st, int a, int i, int j, int k) {
  constexpr int ND = NDims<TOPO>::v;
  float w = (a == 0) ? st.ax[0].dw[i] : st.ax[0].w[i];
  if constexpr (ND > 1) w = w * ((a == 1) ? st.ax[1].dw[j] : st.ax[1].w[j]);
  if constexpr (ND > 2) w = w * ((a == 2) ? st.ax[2].dw[k] : st.ax[2].w[k]);
  return w;
}

// vector access helpers
template <int VW> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<1> { using T = float; };

template <int VW>
__device__ __forceinline__ void vload(float* dst, const float* __restrict__ src) {
  if constexpr (VW == 4) { float4 v = *reinterpret_cast<const float4*>(src); dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
  else if constexpr (VW == 2) { float2 v = *reinterpret_cast<const float2*>(src); dst[0] = v.x; dst[1] = v.y; }
  else dst[0] = *src;
}

// output channel index (reference layouts)
template <int TOPO>
__device__ __forceinline__ long long out_index(const GridDesc& g, long long P, long long b, int s, int d) {
  if constexpr (TOPO == VOXEL) return b * g.D + d;
  else if constexpr (TOPO == HASH) return (long long)d * g.S * P + (long long)s * P + b;   // (D, L, P)
  else return b * (g.D * 3) + d * 3 + s;                                                  // plane fastest
}

#define NDJIR_GRID_THREAD_PROLOGUE                                                   \
  long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;                  \
  long long total = P * g.S;                                                         \
  for (; tid < total; tid += (long long)gridDim.x * blockDim.x) {                    \
    int s = (int)(tid / P);                                                          \
    long long b = tid - (long long)s * P;                                            \
    float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};                 \
    Stencil<TOPO, I> st;                                                             \
    make_stencil<TOPO, I>(st, g, s, q);

#define NDJIR_GRID_THREAD_EPILOGUE }

// ------------------------------------------------------------------------------------------------
// forward: out = sum_taps w * F           (voxel_feature_cuda.cu:33-98 and siblings)
// ------------------------------------------------------------------------------------------------
template <int TOPO, int I, int VW, bool ACCUM>
__global__ void __launch_bounds__(256) k_query(long long P, float* __restrict__ out, const float* __restrict__ query,
                                               const float* __restrict__ feature, GridDesc g) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  NDJIR_GRID_THREAD_PROLOGUE
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    float acc[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) acc[v] = 0.f;
    if constexpr (I == LANCZOS && ND == 3) {
      // a slab of 16 gathers in flight before its products (same order of the sum): at 512^3 x 4 floats every run of 4 taps is
      // a DRAM line, and the loop below left the compiler a few loads per wait
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        float f[NT * NT][VW];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < NT; ++k) vload<VW>(f[j * NT + k], feature + cell_offset(st, i, j, k) + d0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            const float w = tap_w(st, i, j, k);
#pragma unroll
            for (int v = 0; v < VW; ++v) acc[v] += w * f[j * NT + k][v];
          }
      }
    } else {
      NDJIR_FOR_TAPS(ND, NT) {
        float f[VW];
        vload<VW>(f, feature + cell_offset(st, i, j, k) + d0);
        float w = tap_w(st, i, j, k);
#pragma unroll
        for (int v = 0; v < VW; ++v) acc[v] += w * f[v];
      }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      long long o = out_index<TOPO>(g, P, b, s, d0 + v);
      out[o] = ACCUM ? out[o] + acc[v] : acc[v];
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

// Dense voxel forward FUSED with the geometric net's input encoding (round 6; python/network.py:96-117 + :120-151 in one launch):
// e[p] = [x, cos(x_d 2^k) (d major, k fastest), sin(...), voxel feature of p], row stride lde -- what k_query + k_geo_encode
// (csrc/geo.hip) produce in two launches and an intermediate (P, D) tensor; the sampler's rounds of 8 192 points are bound by
// launch latency, not by bytes.  One lane per (point, column); a feature column accumulates its taps in k_query's order with
// k_query's weights: bit-identical values.
template <int I>
__global__ void __launch_bounds__(256) k_voxel_query_encode(long long P, int M, const float* __restrict__ query,
                                                            const float* __restrict__ feature, GridDesc g, float* __restrict__ e,
                                                            int lde) {
  constexpr int TOPO = VOXEL, ND = 3, NT = NTaps<I>::v;
  const int npe = 3 + 6 * M, W = npe + g.D;
  const long long total = P * W;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long p = t / W;
    int c = (int)(t - p * W);
    const int c_out = c;
    float v;
    if (c < 3) v = query[p * 3 + c];
    else if (c < npe) {
      c -= 3;
      const bool is_sin = c >= 3 * M;
      if (is_sin) c -= 3 * M;
      const float b = query[p * 3 + c / M] * (float)(1 << (c % M));
      v = is_sin ? sinf(b) : cosf(b);
    } else {
      const int d = c - npe;
      const float q[3] = {query[p * 3], query[p * 3 + 1], query[p * 3 + 2]};
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, 0, q);
      float acc = 0.f;
      NDJIR_FOR_TAPS(ND, NT) {
        const float f = feature[cell_offset(st, i, j, k) + d];
        const float w = tap_w(st, i, j, k);
        acc += w * f;
      }
      v = acc;
    }
    e[p * lde + c_out] = v;
  }
}

// The tri-plane + tri-line pair of config/triplaneline.yaml (python/network.py:137-147: the two features concatenated behind the
// encoding): e[p] = [x, cos, sin, tri-plane feature (Dp, 3), tri-line feature (Dl, 3)] in one launch -- what k_query_rows x 2 +
// k_geo_encode produced in three (and the sampler's rounds a torch.cat besides).  One lane per (point, column); a feature column
// (d, s) walks the taps of sub-grid s in k_query_rows' order with its products: bit-identical values.
template <int I>
__global__ void __launch_bounds__(256) k_tri_query_encode(long long P, int M, const float* __restrict__ query,
                                                          const float* __restrict__ plane, GridDesc gp,
                                                          const float* __restrict__ line, GridDesc gl, float* __restrict__ e, int lde) {
  constexpr int NT = NTaps<I>::v;
  const int npe = 3 + 6 * M, Cp = 3 * gp.D, W = npe + Cp + 3 * gl.D;
  const long long total = P * W;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long p = t / W;
    int c = (int)(t - p * W);
    const int c_out = c;
    float v;
    if (c < 3) v = query[p * 3 + c];
    else if (c < npe) {
      c -= 3;
      const bool is_sin = c >= 3 * M;
      if (is_sin) c -= 3 * M;
      const float b = query[p * 3 + c / M] * (float)(1 << (c % M));
      v = is_sin ? sinf(b) : cosf(b);
    } else {
      const float q[3] = {query[p * 3], query[p * 3 + 1], query[p * 3 + 2]};
      c -= npe;
      float acc = 0.f;
      if (c < Cp) {
        const int d = c / 3, sub = c - 3 * d;                    // (D, 3): sub-grid fastest (reference layout)
        Stencil<TRIPLANE, I> st;
        make_stencil<TRIPLANE, I>(st, gp, sub, q);
        NDJIR_FOR_TAPS(2, NT) {
          const float w = tap_w(st, i, j, k);
          acc += w * plane[cell_offset(st, i, j, k) + d];
        }
      } else {
        c -= Cp;
        const int d = c / 3, sub = c - 3 * d;
        Stencil<TRILINE, I> st;
        make_stencil<TRILINE, I>(st, gl, sub, q);
        NDJIR_FOR_TAPS(1, NT) {
          const float w = tap_w(st, i, j, k);
          acc += w * line[cell_offset(st, i, j, k) + d];
        }
      }
      v = acc;
    }
    e[p * lde + c_out] = v;
  }
}

// The same for the Lanczos stencil.  There a lane per (point, column) made every feature column repeat the 24 software sines of
// the 4 x 4 x 4 stencil, in waves whose other lanes (the encoding columns) waited for it, and gathered its 64 taps a few at a
// time: 55 us per launch, 0.37 ms per `custom` step.  Here a workgroup takes PTS points: one lane per (point, axis) evaluates
// the axis taps once and hands weights and offsets over in LDS, then a lane per (point, feature column) issues all 64 gathers
// and walks them in k_query's order with k_query's products ((w0 w1) w2, one fma per tap): bit-identical values again, 0.37 ->
// 0.17 ms per step (profiles/r06_scatter.txt).
template <int PTS>
__global__ void __launch_bounds__(256) k_voxel_query_encode_lanczos(long long P, int M, const float* __restrict__ query,
                                                                    const float* __restrict__ feature, GridDesc g,
                                                                    float* __restrict__ e, int lde) {
  __shared__ float s_w[PTS][3][4];
  __shared__ unsigned s_off[PTS][3][4];
  const int tid = (int)threadIdx.x;
  const int npe = 3 + 6 * M;
  // the encoding columns go to the waves that evaluate no taps when that leaves at least half of the workgroup
  constexpr int TAP_LANES = (3 * PTS + 63) & ~63;
  constexpr int ENC_FIRST = (256 - TAP_LANES >= 128) ? TAP_LANES : 0, ENC_LANES = 256 - ENC_FIRST;
  for (long long base = (long long)blockIdx.x * PTS; base < P; base += (long long)gridDim.x * PTS) {   // uniform per workgroup
    const int npts = (int)((P - base) < (long long)PTS ? (P - base) : (long long)PTS);
    if (tid < 3 * npts) {
      const int pt = tid / 3, a = tid - 3 * pt;
      const float G1 = (float)g.G[a] - 1.f, sc = G1 / (g.mx[a] - g.mn[a]);        // make_stencil<VOXEL>, axis a
      const unsigned stride = a == 0 ? (unsigned)(g.G[1] * g.G[2] * g.D) : (a == 1 ? (unsigned)(g.G[2] * g.D) : (unsigned)g.D);
      AxisTaps<LANCZOS> t;
      axis_taps<LANCZOS>(t, (query[(base + pt) * 3 + a] - g.mn[a]) * sc, G1);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_w[pt][a][r] = t.w[r]; s_off[pt][a][r] = t.idx[r] * stride; }
    }
    for (int u = tid - ENC_FIRST; u >= 0 && u < npts * npe; u += ENC_LANES) {     // the encoding columns
      const int pt = u / npe;
      int c = u - pt * npe;
      const long long p = base + pt;
      float v;
      if (c < 3) v = query[p * 3 + c];
      else {
        int cc = c - 3;
        const bool is_sin = cc >= 3 * M;
        if (is_sin) cc -= 3 * M;
        const float b = query[p * 3 + cc / M] * (float)(1 << (cc % M));
        v = is_sin ? sinf(b) : cosf(b);
      }
      e[p * lde + c] = v;
    }
    __syncthreads();
    for (int u = tid; u < npts * g.D; u += 256) {                 // the feature columns
      const int pt = u / g.D, d = u - pt * g.D;
      const float* fd = feature + d;
      // all 64 gathers in flight before the first product (the grid is 2 GiB at 512^3 x 4: every run of 4 taps is a DRAM line)
      float w0[4], w1[4], w2[4], f[64];
      unsigned o0[4], o1[4], o2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w0[r] = s_w[pt][0][r]; w1[r] = s_w[pt][1][r]; w2[r] = s_w[pt][2][r];
        o0[r] = s_off[pt][0][r]; o1[r] = s_off[pt][1][r]; o2[r] = s_off[pt][2][r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) f[(i * 4 + j) * 4 + k] = fd[(long long)(o0[i] + o1[j] + o2[k])];
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w01 = w0[i] * w1[j];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float w = w01 * w2[k];
            acc += w * f[(i * 4 + j) * 4 + k];
          }
        }
      e[(base + pt) * lde + npe + d] = acc;
    }
    __syncthreads();
  }
}

// Tri-plane / tri-line forward, one lane per POINT: the three sub-grids of a point are gathered by the same lane, which then
// owns the point's whole output row (D, 3) -- D * 3 contiguous floats, written as 16-byte stores.  (k_query's lane per
// (sub-grid, point) stores D scalars at a stride of 12 bytes into rows that three lanes of different workgroups share:
// measured 4.2x / 3.1x write amplification at the reference authors' micro-benchmark shape.)
template <int TOPO, int I, int D>
__global__ void __launch_bounds__(256) k_query_rows(long long P, float* __restrict__ out, const float* __restrict__ query,
                                                    const float* __restrict__ feature, GridDesc g) {
  static_assert(TOPO == TRIPLANE || TOPO == TRILINE, "three sub-grids per point");
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < P; b += (long long)gridDim.x * blockDim.x) {
    const float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
    float row[D * 3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, s, q);
      float acc[D];
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = 0.f;
      NDJIR_FOR_TAPS(ND, NT) {
        const float w = tap_w(st, i, j, k);
        const float* cell = feature + cell_offset(st, i, j, k);
#pragma unroll
        for (int d0 = 0; d0 < D; d0 += 4) {
          const float4 f = *reinterpret_cast<const float4*>(cell + d0);
          acc[d0] += w * f.x; acc[d0 + 1] += w * f.y; acc[d0 + 2] += w * f.z; acc[d0 + 3] += w * f.w;
        }
      }
#pragma unroll
      for (int d = 0; d < D; ++d) row[d * 3 + s] = acc[d];           // (D, 3): plane fastest (reference layout)
    }
    float4* o = reinterpret_cast<float4*>(out + b * (D * 3));
#pragma unroll
    for (int v = 0; v < D * 3 / 4; ++v) o[v] = make_float4(row[4 * v], row[4 * v + 1], row[4 * v + 2], row[4 * v + 3]);
  }
}

// Tri-plane / tri-line grad_query (MODE 0) and grad_query_grad_grad_output (MODE 1), one lane per POINT: the three sub-grids'
// contributions to the point's gradient are summed in registers (k_dquery: one fp32 atomic per sub-grid, point and axis), and
// the (D, 3) rows of grad_output / of the result are read / written as contiguous 16-byte pieces.
template <int TOPO, int I, int D, int MODE>
__global__ void __launch_bounds__(256) k_dquery_rows(long long P, float* __restrict__ dst, const float* __restrict__ src,
                                                     const float* __restrict__ query, const float* __restrict__ feature, GridDesc g,
                                                     int accum) {
  static_assert(TOPO == TRIPLANE || TOPO == TRILINE, "three sub-grids per point");
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < P; b += (long long)gridDim.x * blockDim.x) {
    const float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
    float row[D * 3];                    // MODE 0: grad_output row (read); MODE 1: result row (written)
    float gq[3] = {0.f, 0.f, 0.f};
    float gg[3] = {0.f, 0.f, 0.f};
    if (MODE == 0) {
      const float4* o = reinterpret_cast<const float4*>(src + b * (D * 3));
#pragma unroll
      for (int v = 0; v < D * 3 / 4; ++v) { const float4 t = o[v]; row[4 * v] = t.x; row[4 * v + 1] = t.y; row[4 * v + 2] = t.z; row[4 * v + 3] = t.w; }
    } else {
      gg[0] = src[b * 3]; gg[1] = src[b * 3 + 1]; gg[2] = src[b * 3 + 2];
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, s, q);
      float ga[ND][D];
#pragma unroll
      for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int d = 0; d < D; ++d) ga[a][d] = 0.f;
      NDJIR_FOR_TAPS(ND, NT) {
        const float* cell = feature + cell_offset(st, i, j, k);
        float f[D];
#pragma unroll
        for (int d0 = 0; d0 < D; d0 += 4) {
          const float4 t = *reinterpret_cast<const float4*>(cell + d0);
          f[d0] = t.x; f[d0 + 1] = t.y; f[d0 + 2] = t.z; f[d0 + 3] = t.w;
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const float dw = tap_dw(st, a, i, j, k);
#pragma unroll
          for (int d = 0; d < D; ++d) ga[a][d] += dw * f[d];
        }
      }
      if (MODE == 0) {
        float lv[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) lv[a] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int a = 0; a < ND; ++a) lv[a] += row[d * 3 + s] * st.scale[a] * st.ax[a].gm * ga[a][d];
#pragma unroll
        for (int a = 0; a < ND; ++a) gq[st.axis[a]] += lv[a];
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          float r = 0.f;
#pragma unroll
          for (int a = 0; a < ND; ++a) r += gg[st.axis[a]] * st.scale[a] * st.ax[a].gm * ga[a][d];
          row[d * 3 + s] = r;
        }
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) dst[b * 3 + a] = accum ? dst[b * 3 + a] + gq[a] : gq[a];
    } else {
      float4* o = reinterpret_cast<float4*>(dst + b * (D * 3));
#pragma unroll
      for (int v = 0; v < D * 3 / 4; ++v) {
        float4 t = make_float4(row[4 * v], row[4 * v + 1], row[4 * v + 2], row[4 * v + 3]);
        if (accum) { const float4 c = o[v]; t.x += c.x; t.y += c.y; t.z += c.z; t.w += c.w; }
        o[v] = t;
      }
    }
  }
}

// Hash grid grad_query, one lane per POINT: the L levels of a point are summed in registers and the point's (3) gradient is
// written once.  (k_dquery's lane per (level, point) issues one fp32 atomic per level, point and axis: 48 atomics per point
// at L = 16 -- measured 295 MB written for a 6 MB result.)
template <int I, int VW>
__global__ void __launch_bounds__(256) k_dquery_hash_point(long long P, float* __restrict__ gq_out, const float* __restrict__ grad_output,
                                                           const float* __restrict__ query, const float* __restrict__ feature,
                                                           GridDesc g, int accum) {
  constexpr int TOPO = HASH, ND = 3, NT = NTaps<I>::v;
  for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < P; b += (long long)gridDim.x * blockDim.x) {
    const float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
    float gq[3] = {0.f, 0.f, 0.f};
    for (int s = 0; s < g.S; ++s) {
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, s, q);
      float lv[3] = {0.f, 0.f, 0.f};
      for (int d0 = 0; d0 < g.D; d0 += VW) {
        float ga[ND][VW];
#pragma unroll
        for (int a = 0; a < ND; ++a)
#pragma unroll
          for (int v = 0; v < VW; ++v) ga[a][v] = 0.f;
        NDJIR_FOR_TAPS(ND, NT) {
          float f[VW];
          vload<VW>(f, feature + cell_offset(st, i, j, k) + d0);
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            const float dw = tap_dw(st, a, i, j, k);
#pragma unroll
            for (int v = 0; v < VW; ++v) ga[a][v] += dw * f[v];
          }
        }
#pragma unroll
        for (int v = 0; v < VW; ++v) {
          const float og = grad_output[out_index<TOPO>(g, P, b, s, d0 + v)];
#pragma unroll
          for (int a = 0; a < ND; ++a) lv[a] += og * st.scale[a] * st.ax[a].gm * ga[a][v];
        }
      }
#pragma unroll
      for (int a = 0; a < ND; ++a) gq[st.axis[a]] += lv[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) gq_out[b * 3 + a] = accum ? gq_out[b * 3 + a] + gq[a] : gq[a];
  }
}

// ------------------------------------------------------------------------------------------------
// grad_query: gq[b, axis] += sum_d og[d] * scale * gm * sum_taps dw * F   (voxel_feature_cuda.cu:124-204)
// grad_query_grad_grad_output: ggo[d] = sum_axis gg[axis] * scale * gm * sum_taps dw * F  (:328-412)
// MODE 0 = grad_query (atomic into gq), MODE 1 = ggo
// ------------------------------------------------------------------------------------------------
template <int TOPO, int I, int VW, int MODE, bool ACCUM>
__global__ void __launch_bounds__(256) k_dquery(long long P, float* __restrict__ dst, const float* __restrict__ src,
                                                const float* __restrict__ query, const float* __restrict__ feature,
                                                GridDesc g) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  NDJIR_GRID_THREAD_PROLOGUE
  float gq[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) gq[a] = 0.f;
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    float ga[ND][VW];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
      for (int v = 0; v < VW; ++v) ga[a][v] = 0.f;
    if constexpr (I == LANCZOS && ND == 3) {      // slabs of 16 gathers in flight, see k_query
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        float f[NT * NT][VW];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < NT; ++k) vload<VW>(f[j * NT + k], feature + cell_offset(st, i, j, k) + d0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < NT; ++k) {
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              const float dw = tap_dw(st, a, i, j, k);
#pragma unroll
              for (int v = 0; v < VW; ++v) ga[a][v] += dw * f[j * NT + k][v];
            }
          }
      }
    } else {
      NDJIR_FOR_TAPS(ND, NT) {
        float f[VW];
        vload<VW>(f, feature + cell_offset(st, i, j, k) + d0);
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          float dw = tap_dw(st, a, i, j, k);
#pragma unroll
          for (int v = 0; v < VW; ++v) ga[a][v] += dw * f[v];
        }
      }
    }
    if constexpr (MODE == 0) {
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        float og = src[out_index<TOPO>(g, P, b, s, d0 + v)];
#pragma unroll
        for (int a = 0; a < ND; ++a) gq[a] += og * st.scale[a] * st.ax[a].gm * ga[a][v];
      }
    } else {
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        float r = 0.f;
#pragma unroll
        for (int a = 0; a < ND; ++a) r += src[b * 3 + st.axis[a]] * st.scale[a] * st.ax[a].gm * ga[a][v];
        long long o = out_index<TOPO>(g, P, b, s, d0 + v);
        dst[o] = ACCUM ? dst[o] + r : r;
      }
    }
  }
  if constexpr (MODE == 0) {
    if constexpr (TOPO == VOXEL) {
      // one lane owns the point: plain read-modify-write, no atomics needed
#pragma unroll
      for (int a = 0; a < ND; ++a) dst[b * 3 + a] += gq[a];
    } else {
#pragma unroll
      for (int a = 0; a < ND; ++a) atomicAdd(dst + b * 3 + st.axis[a], gq[a]);
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

// The same two products for the Lanczos voxel stencil with D = 4, one lane per (point, channel).  k_dquery's lane per point
// evaluates the stencil's 36 software sines / cosines and then walks 64 float4 gathers a few at a time: 67 us per launch of the
// step's 65 536 points with one wave per SIMD.  Here (as in k_voxel_query_encode_lanczos) one lane per (point, axis) evaluates
// the taps once and hands them over in LDS, and a lane per (point, channel) has all 64 gathers in flight before the first
// product.  Sums in k_dquery's order (taps i, j, k; then channels 0 ... 3 on the point's first lane): bit-identical values.
template <int MODE, bool ACCUM>
__global__ void __launch_bounds__(256) k_dquery_lanczos_voxel(long long P, float* __restrict__ dst, const float* __restrict__ src,
                                                              const float* __restrict__ query, const float* __restrict__ feature,
                                                              GridDesc g) {
  constexpr int PTS = 64;
  __shared__ float s_w[PTS][3][4], s_dw[PTS][3][4], s_sc[PTS][3], s_gm[PTS][3];
  __shared__ unsigned s_off[PTS][3][4];
  const int tid = (int)threadIdx.x;
  for (long long base = (long long)blockIdx.x * PTS; base < P; base += (long long)gridDim.x * PTS) {   // uniform per workgroup
    const int npts = (int)((P - base) < (long long)PTS ? (P - base) : (long long)PTS);
    if (tid < 3 * npts) {
      const int pt = tid / 3, a = tid - 3 * pt;
      const float G1 = (float)g.G[a] - 1.f, sc = G1 / (g.mx[a] - g.mn[a]);        // make_stencil<VOXEL>, axis a
      const unsigned stride = a == 0 ? (unsigned)(g.G[1] * g.G[2] * g.D) : (a == 1 ? (unsigned)(g.G[2] * g.D) : (unsigned)g.D);
      AxisTaps<LANCZOS> t;
      axis_taps<LANCZOS>(t, (query[(base + pt) * 3 + a] - g.mn[a]) * sc, G1);
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_w[pt][a][r] = t.w[r]; s_dw[pt][a][r] = t.dw[r]; s_off[pt][a][r] = t.idx[r] * stride; }
      s_sc[pt][a] = sc;
      s_gm[pt][a] = t.gm;
    }
    __syncthreads();
    const int pt = tid >> 2, v = tid & 3;
    const bool live = pt < npts;                       // (4 lanes of a point share their wave: the shuffles below stay uniform)
    const long long p = base + (live ? pt : 0);
    float ga[3] = {0.f, 0.f, 0.f};
    if (live) {
      const float* fd = feature + v;
      float w0[4], w1[4], w2[4], d0[4], d1[4], d2[4], f[64];
      unsigned o0[4], o1[4], o2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        w0[r] = s_w[pt][0][r]; w1[r] = s_w[pt][1][r]; w2[r] = s_w[pt][2][r];
        d0[r] = s_dw[pt][0][r]; d1[r] = s_dw[pt][1][r]; d2[r] = s_dw[pt][2][r];
        o0[r] = s_off[pt][0][r]; o1[r] = s_off[pt][1][r]; o2[r] = s_off[pt][2][r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) f[(i * 4 + j) * 4 + k] = fd[(long long)(o0[i] + o1[j] + o2[k])];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float fv = f[(i * 4 + j) * 4 + k];
            ga[0] += ((d0[i] * w1[j]) * w2[k]) * fv;        // tap_dw(st, a, i, j, k)
            ga[1] += ((w0[i] * d1[j]) * w2[k]) * fv;
            ga[2] += ((w0[i] * w1[j]) * d2[k]) * fv;
          }
    }
    if constexpr (MODE == 0) {
      // gq[a] += og[v] * scale[a] * gm[a] * ga[a][v], v = 0 ... 3 in turn, on the point's first lane
      float gq[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int vv = 0; vv < 4; ++vv) {
        const float og = src[p * 4 + vv];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float gav = __shfl(ga[a], (tid & 60) + vv);
          gq[a] += og * s_sc[live ? pt : 0][a] * s_gm[live ? pt : 0][a] * gav;
        }
      }
      if (live && v == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) dst[p * 3 + a] += gq[a];
      }
    } else {
      if (live) {
        float r = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) r += src[p * 3 + a] * s_sc[pt][a] * s_gm[pt][a] * ga[a];
        dst[p * 4 + v] = ACCUM ? dst[p * 4 + v] + r : r;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// scatter kernels:
//  MODE 0 grad_feature:            gf[cell,d] += og[d] * w                        (:230-286)
//  MODE 1 grad_query_grad_feature: gf[cell,d] += og[d] * sum_a gg[a]*scale*gm*dw  (:548-614)
// ------------------------------------------------------------------------------------------------
template <int TOPO, int I, int VW, int MODE>
__global__ void __launch_bounds__(256) k_scatter(long long P, float* __restrict__ gf, const float* __restrict__ gg_query,
                                                 const float* __restrict__ grad_output, const float* __restrict__ query,
                                                 GridDesc g) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  NDJIR_GRID_THREAD_PROLOGUE
  float ggs[ND];
  if constexpr (MODE == 1) {
#pragma unroll
    for (int a = 0; a < ND; ++a) ggs[a] = gg_query[b * 3 + st.axis[a]] * st.scale[a] * st.ax[a].gm;
  }
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    float og[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) og[v] = grad_output[out_index<TOPO>(g, P, b, s, d0 + v)];
    NDJIR_FOR_TAPS(ND, NT) {
      float w;
      if constexpr (MODE == 0) {
        w = tap_w(st, i, j, k);
      } else {
        w = 0.f;
#pragma unroll
        for (int a = 0; a < ND; ++a) w += ggs[a] * tap_dw(st, a, i, j, k);
      }
      float* p = gf + cell_offset(st, i, j, k) + d0;
#pragma unroll
      for (int v = 0; v < VW; ++v) atomicAdd(p + v, og[v] * w);
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

// ------------------------------------------------------------------------------------------------
// The same scatters for the dense topologies, pre-aggregated per workgroup.  The fp32 atomics of a scatter resolve in
// the memory-side cache (the 8 XCD L2s are not coherent) at a rate that bounds the kernel, and that rate is per REQUEST
// of up to 64 contiguous, 64-byte-aligned bytes, not per float (tools/ubench/atomics_shape.hip, profiles/r06_ubench.txt:
// 20 G requests/s at 4 ... 64 bytes, bytes-bound beyond).  A stencil is a set of RUNS along the fastest axis -- 2 (linear)
// or 4 (Lanczos) consecutive cells of D floats, 32 ... 128 contiguous bytes -- and consecutive samples of a ray revisit
// cells.  Each workgroup therefore accumulates its points in an LDS hash table whose entries are the 64-byte-aligned
// blocks of the gradient tensor (LDS atomics), and flushes every touched block with one lane per float: one request
// per block.  Per point that is 4 x 1.25 requests (linear voxel, D = 4) instead of 8, 16 x 1.75 instead of 64 (Lanczos).
// ------------------------------------------------------------------------------------------------
#ifndef NDJIR_AGG_ENT
#define NDJIR_AGG_ENT 256
#endif
constexpr int AGG_ENT = NDJIR_AGG_ENT;     // table entries of 16 floats.  Best effort: a run that finds no slot within 8 probes is listed
                                           // and flushed as it is (spread points have nothing to merge anyway); the table is kept
                                           // small because a pass clears and scans all of it (1024 entries: 26 us per launch of the
                                           // step's 65 536 points, 256: 16 us, 128: 35 us -- profiles/r06_scatter.txt)
constexpr int AGG_OVER = 1024;             // list capacity in float4 chunks: the launcher keeps lanes x chunks per run within it

template <int NT, class T>
__device__ __forceinline__ T pick_tap(const T (&a)[NT], int i) {      // selected, not indexed: the taps live in registers
  if constexpr (NT == 2) return i == 0 ? a[0] : a[1];
  else return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3];
}

// One lane per RUN: LPP = NT^(ND - 1) lanes share a point (4 linear voxel, 16 Lanczos voxel, 2 tri-plane, 1 tri-line), every
// one of them evaluates the point's stencil and keeps its own (i, j); ppp = points per pass and workgroup (<= 256 / LPP).
template <int TOPO, int I, int MODE>
__global__ void __launch_bounds__(256) k_scatter_agg(long long P, float* __restrict__ gf, const float* __restrict__ gg_query,
                                                     const float* __restrict__ grad_output, const float* __restrict__ query,
                                                     GridDesc g, int ppp) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  constexpr int LPP = ND == 3 ? NT * NT : (ND == 2 ? NT : 1);
  __shared__ int keys[AGG_ENT];
  __shared__ float vals[AGG_ENT * 16];
  __shared__ int over_key[AGG_OVER];         // float4 chunks that found no slot: chunk index in gf (-1: none) ...
  __shared__ float over_val[AGG_OVER * 4];   // ... and value
  __shared__ int n_over;
  const long long total = P * g.S;
  const long long per_pass = (long long)gridDim.x * ppp;
  const int D4 = g.D >> 2;
  for (int t = threadIdx.x; t < AGG_OVER; t += 256) over_key[t] = -1;
  if (threadIdx.x == 0) n_over = 0;
  // blocks are aligned in MEMORY, not relative to gf (a gradient tensor may be a slice of a flat buffer); a base that is not
  // 16-byte aligned keeps the relative blocks -- still correct, a float4 chunk must not straddle two entries
  int a0 = (int)((reinterpret_cast<uintptr_t>(gf) >> 2) & 15);
  if (a0 & 3) a0 = 0;
  for (long long base = (long long)blockIdx.x * ppp; base < total; base += per_pass) {   // uniform per workgroup
    for (int t = threadIdx.x; t < AGG_ENT * 4; t += 256) {
      if (t < AGG_ENT) keys[t] = -1;
      *reinterpret_cast<float4*>(vals + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int pt = (int)threadIdx.x / LPP, sub = (int)threadIdx.x % LPP;
    const long long tid = base + pt;
    if (pt < ppp && tid < total) {
      const int s = (int)(tid / P);
      const long long b = tid - (long long)s * P;
      float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, s, q);
      float ggs[ND];
      if constexpr (MODE == 1) {
#pragma unroll
        for (int a = 0; a < ND; ++a) ggs[a] = gg_query[b * 3 + st.axis[a]] * st.scale[a] * st.ax[a].gm;
      }
      if constexpr (ND >= 2) {                   // this lane's run: taps (i0, j0, all k) / (i0, all j)
        const int i0 = ND == 3 ? sub / NT : sub;
        const float w = pick_tap<NT>(st.ax[0].w, i0), dw = pick_tap<NT>(st.ax[0].dw, i0);
        const unsigned ix = pick_tap<NT>(st.ax[0].idx, i0);
        st.ax[0].w[0] = w; st.ax[0].dw[0] = dw; st.ax[0].idx[0] = ix;
      }
      if constexpr (ND == 3) {
        const int j0 = sub % NT;
        const float w = pick_tap<NT>(st.ax[1].w, j0), dw = pick_tap<NT>(st.ax[1].dw, j0);
        const unsigned ix = pick_tap<NT>(st.ax[1].idx, j0);
        st.ax[1].w[0] = w; st.ax[1].dw[0] = dw; st.ax[1].idx[0] = ix;
      }
      int last_ek = -1;
      int slot = -1;
      int rec = -1;                               // first record of this run in the overflow list, once a chunk found no slot
      for (int d0 = 0; d0 < g.D; d0 += 4) {       // a cell of D floats = D / 4 float4 chunks
        float og[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) og[v] = grad_output[out_index<TOPO>(g, P, b, s, d0 + v)];
#pragma unroll
        for (int r = 0; r < NT; ++r) {
          const int i = ND == 1 ? r : 0, j = ND == 2 ? r : 0, k = ND == 3 ? r : 0;
          float w;
          if constexpr (MODE == 0) {
            w = tap_w(st, i, j, k);
          } else {
            w = 0.f;
#pragma unroll
            for (int a = 0; a < ND; ++a) w += ggs[a] * tap_dw(st, a, i, j, k);
          }
          const long long f = cell_offset(st, i, j, k) + d0 + a0;           // float index, block-aligned
          const int ek = (int)(f >> 4), ce = (int)(f >> 2) & 3;
          if (rec < 0 && ek != last_ek) {          // the taps of a run mostly share their block
            unsigned h = ((unsigned)ek * 2654435761u) >> (32 - __builtin_ctz(AGG_ENT));
            slot = -1;
            for (int probe = 0; probe < 8; ++probe) {
              const int old = atomicCAS(&keys[h], -1, ek);
              if (old == -1 || old == ek) { slot = (int)h; break; }
              h = (h + 1) & (AGG_ENT - 1);
            }
            last_ek = ek;
            // no slot within 8 probes: this and the remaining chunks of the run go to the list, in the run's memory order
            // (record r D4 + d0 / 4), so that the flush addresses them as the contiguous bytes they are
            if (slot < 0) rec = atomicAdd(&n_over, NT * D4);
          }
          if (rec < 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) atomicAdd(&vals[16 * slot + 4 * ce + v], og[v] * w);
          } else {
            const int e = rec + r * D4 + (d0 >> 2);
            over_key[e] = (int)((f - a0) >> 2);
#pragma unroll
            for (int v = 0; v < 4; ++v) over_val[4 * e + v] = og[v] * w;
          }
        }
      }
    }
    __syncthreads();
    // one lane per FLOAT, 16 lanes per block: one request.  Floats that received nothing are exact zeros and are skipped --
    // a block may reach past either end of the tensor, its untouched part is never addressed.
    for (int t = threadIdx.x; t < AGG_ENT * 16; t += 256) {
      const int ek = keys[t >> 4];
      const float v = vals[t];
      if (ek >= 0 && v != 0.f) atomicAdd(gf + (((long long)ek << 4) - a0 + (t & 15)), v);
    }
    // the list: 4 lanes per chunk, the chunks of a run side by side
    const int no = n_over;
    for (int t = threadIdx.x; t < no * 4; t += 256) {
      const int e = t >> 2, key = over_key[e];
      if (key >= 0) atomicAdd(gf + ((long long)key << 2) + (t & 3), over_val[t]);
      if ((t & 3) == 0) over_key[e] = -1;          // (same wave instruction as the reads of its three neighbours)
    }
    __syncthreads();
    if (threadIdx.x == 0) n_over = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// Lanczos voxel scatters (D = 4), one lane per FLOAT of the stencil.  A point's 4 x 4 x 4 taps of 4 floats are 16 runs of 64
// contiguous bytes (4 cells along the fastest axis): 256 floats, one per lane of the workgroup -- lane (i, j, k, v) keeps its
// place in the stencil and walks the workgroup's 64 points.  What the generic table kernel above paid for these stencils
// (profiles/r06_scatter.txt: 185 / 230 us per launch of the step's 65 536 points, none of it in the flush):
//   - every lane of a point repeated the 24 software sines of the stencil: here one lane per (point, axis) evaluates the
//     axis taps once and hands them over in LDS;
//   - a lane per tap inserted with 1 CAS + 4 LDS atomics whose lanes collide on the cells consecutive samples share: here
//     a wave instruction is one point's slab of 4 runs, 64 different floats; the one or two 64-byte blocks of a run are
//     claimed by its leader lanes only and the slot is passed on with a wave shuffle;
//   - consecutive samples of a ray that stay on the same cells are summed in the lane's register first.
// The table is best effort (256 blocks, 8 probes): a run that finds no slot goes to memory directly, which is also all a set of
// spread points can do -- there a run is 1.75 requests of the memory-side atomic unit where a lane per tap issued 16.
// Values are those of k_scatter: og * ((w0 * w1) * w2), resp. og * sum_a gg_a * tap_dw_a in the same order.
// ------------------------------------------------------------------------------------------------
#ifndef NDJIR_LZ_PTS
#define NDJIR_LZ_PTS 64
#endif
#ifndef NDJIR_LZ_ENT
#define NDJIR_LZ_ENT 256
#endif
constexpr int LZ_PTS = NDJIR_LZ_PTS;        // points per pass and workgroup
constexpr int LZ_ENT = NDJIR_LZ_ENT;        // table entries of 16 floats
static_assert(4 * LZ_PTS <= 256, "one lane per (point, axis) + one per point");

template <int MODE>
__global__ void __launch_bounds__(256) k_scatter_lanczos_voxel(long long P, float* __restrict__ gf, const float* __restrict__ gg_query,
                                                               const float* __restrict__ grad_output, const float* __restrict__ query,
                                                               GridDesc g) {
  __shared__ float s_w[LZ_PTS][3][4], s_dw[LZ_PTS][3][4], s_gg[LZ_PTS][3], s_og[LZ_PTS][4];
  __shared__ unsigned s_off[LZ_PTS][3][4];
  __shared__ int keys[LZ_ENT];
  __shared__ float vals[LZ_ENT * 16];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int i = tid >> 6, j = (tid >> 4) & 3, k = (tid >> 2) & 3, v = tid & 3;
  int a0 = (int)((reinterpret_cast<uintptr_t>(gf) >> 2) & 15);      // blocks are aligned in memory (see k_scatter_agg)
  if (a0 & 3) a0 = 0;
  for (long long base = (long long)blockIdx.x * LZ_PTS; base < P; base += (long long)gridDim.x * LZ_PTS) {   // uniform per workgroup
    for (int t = tid; t < LZ_ENT * 4; t += 256) {
      if (t < LZ_ENT) keys[t] = -1;
      *reinterpret_cast<float4*>(vals + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 3 * LZ_PTS) {
      const int pt = tid / 3, a = tid - 3 * pt;
      const long long b = base + pt;
      if (b < P) {
        const float G1 = (float)g.G[a] - 1.f, sc = G1 / (g.mx[a] - g.mn[a]);        // make_stencil<VOXEL>, axis a
        const unsigned stride = a == 0 ? (unsigned)(g.G[1] * g.G[2] * g.D) : (a == 1 ? (unsigned)(g.G[2] * g.D) : (unsigned)g.D);
        AxisTaps<LANCZOS> t;
        axis_taps<LANCZOS>(t, (query[b * 3 + a] - g.mn[a]) * sc, G1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_w[pt][a][r] = t.w[r]; s_dw[pt][a][r] = t.dw[r]; s_off[pt][a][r] = t.idx[r] * stride; }
        if constexpr (MODE == 1) s_gg[pt][a] = gg_query[b * 3 + a] * sc * t.gm;
      }
    } else {
      const int pt = tid - 3 * LZ_PTS;
      if (base + pt < P) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_og[pt][c] = grad_output[(base + pt) * 4 + c];
      }
    }
    __syncthreads();
    const int npts = (int)((P - base) < (long long)LZ_PTS ? (P - base) : (long long)LZ_PTS);
    // (cur, acc): the run of points whose stencil sits on the same cells; handed to the table when the wave moves on
    auto hand_over = [&](unsigned o, float val) {
      const long long f = (long long)o + a0;
      const int ek = (int)(f >> 4), within = ((int)(f >> 2) & 3) * 4 + v;
      // the blocks of a run: the cells of k = 0 ... kb - 1 in the first, the rest in the second; leaders = their first lanes
      const int run0 = lane & ~15;
      const int ek0 = __shfl(ek, run0);
      const bool same0 = ek == ek0;
      const unsigned long long m = __ballot(same0);
      const int kb = __popcll((m >> run0) & 0x1111ull);
      const int leader = same0 ? run0 : run0 + 4 * kb;
      int slot = -1;
      if (lane == leader) {
        unsigned h = ((unsigned)ek * 2654435761u) >> (32 - __builtin_ctz(LZ_ENT));
        for (int probe = 0; probe < 8; ++probe) {
          const int old = atomicCAS(&keys[h], -1, ek);
          if (old == -1 || old == ek) { slot = (int)h; break; }
          h = (h + 1) & (LZ_ENT - 1);
        }
      }
      slot = __shfl(slot, leader);
      if (slot >= 0) atomicAdd(&vals[16 * slot + within], val);
      else atomicAdd(gf + (long long)o + v, val);
    };
    unsigned cur = 0xffffffffu;
    float acc = 0.f;
    for (int pt = 0; pt < npts; ++pt) {
      const float w0 = s_w[pt][0][i], w1 = s_w[pt][1][j], w2 = s_w[pt][2][k];
      float w;
      if constexpr (MODE == 0) {
        w = w0 * w1;
        w = w * w2;
      } else {
        const float d0w = s_dw[pt][0][i], d1w = s_dw[pt][1][j], d2w = s_dw[pt][2][k];
        w = 0.f;
        w += s_gg[pt][0] * ((d0w * w1) * w2);
        w += s_gg[pt][1] * ((w0 * d1w) * w2);
        w += s_gg[pt][2] * ((w0 * w1) * d2w);
      }
      const unsigned o = s_off[pt][0][i] + s_off[pt][1][j] + s_off[pt][2][k];
      if (__any(o != cur)) {                       // wave-uniform: the shuffles of hand_over need every lane
        if (pt > 0) hand_over(cur, acc);
        cur = o;
        acc = 0.f;
      }
      acc += s_og[pt][v] * w;
    }
    if (npts > 0) hand_over(cur, acc);
    __syncthreads();
    // one lane per FLOAT, 16 lanes per block: one request; untouched floats are exact zeros and are skipped (see k_scatter_agg)
    for (int t = tid; t < LZ_ENT * 16; t += 256) {
      const int ek = keys[t >> 4];
      const float val = vals[t];
      if (ek >= 0 && val != 0.f) atomicAdd(gf + (((long long)ek << 4) - a0 + (t & 15)), val);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Hash-grid scatters through a table image in LDS.  A level's table is small (T <= 2^15 ... 2^19 entries of D floats:
// 256 KB at the reference bench's T0 = 2^15, D = 2) and EVERY point writes 8 (Lanczos: 64) pseudo-random entries of it:
// nothing to merge inside a wave, and the device-wide fp32 atomic rate (~20 G/s, tools/ubench/atomics.hip) prices the
// plain scatter at 256 atomics per point.  Here a workgroup owns one slice of one level's table (<= 32768 floats, 128 KB
// of LDS), walks a share of the points, adds the taps that fall into its slice with LDS atomics and flushes the slice
// once: global atomics drop from taps x D per (point, level) to (slice floats) per workgroup.
//   grid = levels x slices per level x point shares ;  workgroup (s, c, k): level s, entries [c E, (c+1) E), points k, k+K, ...
// Every workgroup evaluates all taps of its points and keeps those of its slice, so the hash arithmetic is repeated once
// per slice of the level -- the launcher uses this path while a level has at most HASH_LDS_MAX_SLICES slices.
// ------------------------------------------------------------------------------------------------
constexpr int HASH_LDS_FLOATS = 32768;
constexpr int HASH_LDS_MAX_SLICES = 8;

template <int I, int MODE>
__global__ void __launch_bounds__(256) k_scatter_hash_lds(long long P, float* __restrict__ gf, const float* __restrict__ gg_query,
                                                          const float* __restrict__ grad_output, const float* __restrict__ query,
                                                          GridDesc g, int slices, int shares, int entries_per_slice) {
  constexpr int TOPO = HASH;
  constexpr int ND = 3, NT = NTaps<I>::v;
  extern __shared__ float hs_tab[];
  const int share = blockIdx.x % shares;
  const int c = (blockIdx.x / shares) % slices;
  const int s = blockIdx.x / (shares * slices);
  const int T = g.lvlT[s], D = g.D;
  const int lo = c * entries_per_slice;
  int n_ent = T - lo;
  if (n_ent > entries_per_slice) n_ent = entries_per_slice;
  if (n_ent <= 0) return;
  const int n_fl = n_ent * D;
  for (int t = threadIdx.x; t < n_fl; t += 256) hs_tab[t] = 0.f;
  __syncthreads();
  for (long long b = (long long)share * 256 + threadIdx.x; b < P; b += (long long)shares * 256) {
    float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
    Stencil<TOPO, I> st;
    make_stencil<TOPO, I>(st, g, s, q);
    float ggs[ND];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int a = 0; a < ND; ++a) ggs[a] = gg_query[b * 3 + st.axis[a]] * st.scale[a] * st.ax[a].gm;
    }
    float og[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) og[d] = d < D ? grad_output[out_index<TOPO>(g, P, b, s, d)] : 0.f;
    NDJIR_FOR_TAPS(ND, NT) {
      const int e = (int)hash3(st.ax[0].idx[i], st.ax[1].idx[j], st.ax[2].idx[k], T) - lo;
      if (e >= 0 && e < n_ent) {
        float w;
        if constexpr (MODE == 0) {
          w = tap_w(st, i, j, k);
        } else {
          w = 0.f;
#pragma unroll
          for (int a = 0; a < ND; ++a) w += ggs[a] * tap_dw(st, a, i, j, k);
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) if (d < D) atomicAdd(&hs_tab[e * D + d], og[d] * w);
      }
    }
  }
  __syncthreads();
  float* dst = gf + g.lvlOff[s] + (long long)lo * D;
  if (shares == 1) {
    for (int t = threadIdx.x; t < n_fl; t += 256) dst[t] += hs_tab[t];          // sole owner of the slice
  } else {
    for (int t = threadIdx.x; t < n_fl; t += 256) {
      const float v = hs_tab[t];
      if (v != 0.f) atomicAdd(dst + t, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Tri-line scatters through an LDS image of a whole line (G x D floats: 64 KB at G = 2048, D = 8), the 1-D case of the hash
// path above: workgroup (s, k) owns line s for the points k, k + K, ..., adds their taps with LDS atomics and flushes the
// line once.  Pays when the points outnumber the line's cells by far (the micro-benchmark's 2^19 points: 25 M global
// atomics become 3 x K x 16384); the launcher keeps the aggregated path for a training step's 65 536 samples.
// ------------------------------------------------------------------------------------------------
template <int I, int MODE>
__global__ void __launch_bounds__(256) k_scatter_line_lds(long long P, float* __restrict__ gf, const float* __restrict__ gg_query,
                                                          const float* __restrict__ grad_output, const float* __restrict__ query,
                                                          GridDesc g, int shares) {
  constexpr int TOPO = TRILINE;
  constexpr int ND = 1, NT = NTaps<I>::v;
  extern __shared__ float hs_tab[];
  const int share = blockIdx.x % shares, s = blockIdx.x / shares;
  const int D = g.D, n_fl = g.G[0] * D;
  for (int t = threadIdx.x; t < n_fl; t += 256) hs_tab[t] = 0.f;
  __syncthreads();
  for (long long b = (long long)share * 256 + threadIdx.x; b < P; b += (long long)shares * 256) {
    float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
    Stencil<TOPO, I> st;
    make_stencil<TOPO, I>(st, g, s, q);
    float ggs = 0.f;
    if constexpr (MODE == 1) ggs = gg_query[b * 3 + st.axis[0]] * st.scale[0] * st.ax[0].gm;
    float og[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) og[d] = d < D ? grad_output[out_index<TOPO>(g, P, b, s, d)] : 0.f;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const float w = (MODE == 0) ? st.ax[0].w[i] : ggs * st.ax[0].dw[i];
      const int e = (int)st.ax[0].idx[i] * D;
#pragma unroll
      for (int d = 0; d < 8; ++d) if (d < D) atomicAdd(&hs_tab[e + d], og[d] * w);
    }
  }
  __syncthreads();
  float* dst = gf + (long long)s * n_fl;
  for (int t = threadIdx.x; t < n_fl; t += 256) {
    const float v = hs_tab[t];
    if (v != 0.f) {
      if (shares == 1) dst[t] += v;
      else atomicAdd(dst + t, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// zero the cells a set of queries touches (every tap of the stencil): re-arms an accumulate-in-place
// gradient buffer after use without rewriting all of it (2 GiB for the default 512^3 x 4 grid).
// Covers grad_feature, grad_query_grad_feature and the TV backward (its cells are a subset of the taps).
// ------------------------------------------------------------------------------------------------
// CHECK: instead of clearing, raise *flag when a touched cell holds an inf or nan (the finite-gradient guard of
// python/solver.py:67-69 restricted to the cells that can hold a gradient at all).
template <int TOPO, int I, int VW, bool CHECK>
__global__ void __launch_bounds__(256) k_zero_touched(long long P, float* __restrict__ gf, const float* __restrict__ query,
                                                      GridDesc g, int* __restrict__ flag) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  bool bad = false;
  NDJIR_GRID_THREAD_PROLOGUE
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    NDJIR_FOR_TAPS(ND, NT) {
      float* p = gf + cell_offset(st, i, j, k) + d0;
      if (CHECK) {
        float x[VW];
        vload<VW>(x, p);
#pragma unroll
        for (int v = 0; v < VW; ++v) bad |= !isfinite(x[v]);
      } else {
#pragma unroll
        for (int v = 0; v < VW; ++v) p[v] = 0.f;
      }
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
  if (CHECK && bad) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------
// Sparse gradient exchange (multi-GPU, no reference counterpart): append the NON-ZERO rows of the cells that the query
// points touch in an accumulate-in-place gradient buffer (D = 4: one float4 per cell) to a packed list (cell id, row),
// each cell once -- a persistent bitmap (1 bit per cell) marks the cells already listed.  `count` keeps counting past
// `capacity` (the caller sees the overflow and repeats with a larger list).
// ------------------------------------------------------------------------------------------------
// one bit per cell (D = 4: per float4 of the gradient buffer) for every cell the query points touch
__global__ void __launch_bounds__(256) k_mark_touched(long long P, const float* __restrict__ query, GridDesc g,
                                                      unsigned* __restrict__ bitmap) {
  constexpr int TOPO = VOXEL, I = LINEAR;
  constexpr int ND = 3, NT = 2;
  NDJIR_GRID_THREAD_PROLOGUE
  NDJIR_FOR_TAPS(ND, NT) {
    const unsigned cell = (unsigned)(cell_offset(st, i, j, k) >> 2);
    const unsigned bit = 1u << (cell & 31);
    if (!(bitmap[cell >> 5] & bit)) atomicOr(bitmap + (cell >> 5), bit);
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

template <int TOPO, int I, int D4>      // D4 = float4 chunks per row (D = 4 or 8)
__global__ void __launch_bounds__(256) k_pack_rows(long long P, const float* __restrict__ gf, const float* __restrict__ query,
                                                   GridDesc g, unsigned* __restrict__ bitmap, int* __restrict__ ids,
                                                   float4* __restrict__ rows, int* __restrict__ count, int capacity) {
  constexpr int ND = NDims<TOPO>::v, NT = NTaps<I>::v;
  NDJIR_GRID_THREAD_PROLOGUE
  NDJIR_FOR_TAPS(ND, NT) {
    const long long off = cell_offset(st, i, j, k);
    float4 v[D4];
    bool nz = false;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
      v[c] = *reinterpret_cast<const float4*>(gf + off + 4 * c);
      nz |= (v[c].x != 0.f || v[c].y != 0.f || v[c].z != 0.f || v[c].w != 0.f);
    }
    if (nz) {
      const unsigned cell = (unsigned)(off / (4 * D4));
      const unsigned bit = 1u << (cell & 31);
      const unsigned old = atomicOr(bitmap + (cell >> 5), bit);
      if (!(old & bit)) {
        const int slot = atomicAdd(count, 1);
        if (slot < capacity) {
          ids[slot] = (int)cell;
#pragma unroll
          for (int c = 0; c < D4; ++c) rows[(long long)slot * D4 + c] = v[c];
        } else {
          // no room in the list: the cell is NOT listed, so its bit must not stay set -- k_rows_clear_bitmap clears the bits of
          // listed cells only, and a cell whose bit survived would never be listed (or exchanged) again.  *count keeps counting
          // (another lane of the same cell may add once more): any value above the capacity reads "rows were dropped" -- the
          // overflow flag vetoes the step and k_rows_zero clears the whole buffer (sparse_rows.hip)
          atomicAnd(bitmap + (cell >> 5), ~bit);
        }
      }
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

// ------------------------------------------------------------------------------------------------
// grad_query_grad_query, linear dense voxel only (voxel_feature_cuda.cu:440-520). Accumulates.
// ------------------------------------------------------------------------------------------------
template <int VW>
__global__ void __launch_bounds__(256) k_voxel_gq_gq(long long P, float* __restrict__ gq, const float* __restrict__ gg_query,
                                                     const float* __restrict__ grad_output, const float* __restrict__ query,
                                                     const float* __restrict__ feature, GridDesc g) {
  constexpr int TOPO = VOXEL, I = LINEAR;
  NDJIR_GRID_THREAD_PROLOGUE
  float p0 = st.ax[0].w[0], p1 = st.ax[0].w[1], q0 = st.ax[1].w[0], q1 = st.ax[1].w[1], r0 = st.ax[2].w[0], r1 = st.ax[2].w[1];
  float sx = st.scale[0], sy = st.scale[1], sz = st.scale[2];
  float ggx = gg_query[b * 3], ggy = gg_query[b * 3 + 1], ggz = gg_query[b * 3 + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    float f[2][2][2][VW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) vload<VW>(f[i][j][k], feature + cell_offset(st, i, j, k) + d0);
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      float go = grad_output[b * g.D + d0 + v];
      float ti = go * sy * sz * (p0 * (f[0][0][0][v] - f[0][0][1][v] - f[0][1][0][v] + f[0][1][1][v]) +
                                 p1 * (f[1][0][0][v] - f[1][0][1][v] - f[1][1][0][v] + f[1][1][1][v]));
      float tj = go * sx * sz * (q0 * (f[0][0][0][v] - f[0][0][1][v] - f[1][0][0][v] + f[1][0][1][v]) +
                                 q1 * (f[0][1][0][v] - f[0][1][1][v] - f[1][1][0][v] + f[1][1][1][v]));
      float tk = go * sx * sy * (r0 * (f[0][0][0][v] - f[0][1][0][v] - f[1][0][0][v] + f[1][1][0][v]) +
                                 r1 * (f[0][0][1][v] - f[0][1][1][v] - f[1][0][1][v] + f[1][1][1][v]));
      ax += ggy * tk + ggz * tj;
      ay += ggz * ti + ggx * tk;
      az += ggx * tj + ggy * ti;
    }
  }
  gq[b * 3] += ax; gq[b * 3 + 1] += ay; gq[b * 3 + 2] += az;
  NDJIR_GRID_THREAD_EPILOGUE
}

// ------------------------------------------------------------------------------------------------
// sampled total-variation loss (total_variation_loss*_cuda.cu); backward always accumulates.
// ------------------------------------------------------------------------------------------------
template <int TOPO, int VW, bool BWD>
__global__ void __launch_bounds__(256) k_tv(long long P, float* __restrict__ dst, const float* __restrict__ grad_output,
                                            const float* __restrict__ query, const float* __restrict__ feature,
                                            GridDesc g, int sym_backward) {
  constexpr int I = LINEAR;
  constexpr int ND = NDims<TOPO>::v;
  NDJIR_GRID_THREAD_PROLOGUE
  long long o0 = cell_offset(st, 0, 0, 0);
  long long oa[ND];
  oa[0] = cell_offset(st, 1, 0, 0);
  if constexpr (ND > 1) oa[1] = cell_offset(st, 0, 1, 0);
  if constexpr (ND > 2) oa[2] = cell_offset(st, 0, 0, 1);
  for (int d0 = 0; d0 < g.D; d0 += VW) {
    float f0[VW], fa[ND][VW];
    vload<VW>(f0, feature + o0 + d0);
#pragma unroll
    for (int a = 0; a < ND; ++a) vload<VW>(fa[a], feature + oa[a] + d0);
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      float del[ND], s2 = 0.f;
#pragma unroll
      for (int a = 0; a < ND; ++a) { del[a] = fa[a][v] - f0[v]; s2 += del[a] * del[a]; }
      long long o = out_index<TOPO>(g, P, b, s, d0 + v);
      if constexpr (!BWD) {
        dst[o] = sqrtf(s2);
      } else {
        // total_variation_loss_cuda.cu:158-170: rsqrt(.. + 1e-12) evaluated in double
        double common = (double)grad_output[o] * (1.0 / sqrt((double)s2 + 1e-12));
        double gsum = 0.0;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          double ga = common * (double)del[a];
          gsum += ga;
          atomicAdd(dst + oa[a] + d0 + v, (float)ga);
        }
        if (sym_backward) atomicAdd(dst + o0 + d0 + v, (float)(-gsum));
      }
    }
  }
  NDJIR_GRID_THREAD_EPILOGUE
}

// TV backward with a workgroup-level LDS aggregation like k_scatter_agg's (dense topologies, D = 4 / 8).  Its four cells per point
// are not a run and its points revisit cells heavily, so it keeps the exact table: cells (16 / 32 bytes) as entries, 4096 slots, at
// most 256 x 4 x D / 4 insertions per pass (the small best-effort table of k_scatter_agg was 3 x slower here: 22 -> 80 us)
constexpr int AGG_HT = 4096;
template <int TOPO>
__global__ void __launch_bounds__(256) k_tv_bwd_agg(long long P, float* __restrict__ dst, const float* __restrict__ grad_output,
                                                    const float* __restrict__ query, const float* __restrict__ feature,
                                                    GridDesc g, int sym_backward) {
  constexpr int I = LINEAR, ND = NDims<TOPO>::v;
  __shared__ int keys[AGG_HT];
  __shared__ float vals[AGG_HT * 4];
  const long long total = P * g.S;
  const long long per_pass = (long long)gridDim.x * 256;
  const int cw_log = (g.D == 8) ? 1 : 0;          // float4 chunks per table entry (see k_scatter_agg)
  auto add = [&](long long off, const float* v4) {
    const int key = (int)(off >> 2);
    const int ek = key >> cw_log, ce = key & ((1 << cw_log) - 1);
    unsigned slot = ((unsigned)ek * 2654435761u) >> (20 + cw_log);
    while (true) {
      const int old = atomicCAS(&keys[slot], -1, ek);
      if (old == -1 || old == ek) break;
      slot = (slot + 1) & ((AGG_HT >> cw_log) - 1);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) atomicAdd(&vals[4 * ((slot << cw_log) + ce) + v], v4[v]);
  };
  for (long long base = (long long)blockIdx.x * 256; base < total; base += per_pass) {
    for (int t = threadIdx.x; t < AGG_HT; t += 256) {
      keys[t] = -1;
      *reinterpret_cast<float4*>(vals + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const long long tid = base + threadIdx.x;
    if (tid < total) {
      const int s = (int)(tid / P);
      const long long b = tid - (long long)s * P;
      float q[3] = {query[b * 3], query[b * 3 + 1], query[b * 3 + 2]};
      Stencil<TOPO, I> st;
      make_stencil<TOPO, I>(st, g, s, q);
      const long long o0 = cell_offset(st, 0, 0, 0);
      long long oa[ND];
      oa[0] = cell_offset(st, 1, 0, 0);
      if constexpr (ND > 1) oa[1] = cell_offset(st, 0, 1, 0);
      if constexpr (ND > 2) oa[2] = cell_offset(st, 0, 0, 1);
      for (int d0 = 0; d0 < g.D; d0 += 4) {
        float f0[4], fa[ND][4], ga[ND][4], g0[4];
        vload<4>(f0, feature + o0 + d0);
#pragma unroll
        for (int a = 0; a < ND; ++a) vload<4>(fa[a], feature + oa[a] + d0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float del[ND], s2 = 0.f;
#pragma unroll
          for (int a = 0; a < ND; ++a) { del[a] = fa[a][v] - f0[v]; s2 += del[a] * del[a]; }
          // total_variation_loss_cuda.cu:158-170: rsqrt(.. + 1e-12) evaluated in double
          const double common = (double)grad_output[out_index<TOPO>(g, P, b, s, d0 + v)] * (1.0 / sqrt((double)s2 + 1e-12));
          double gsum = 0.0;
#pragma unroll
          for (int a = 0; a < ND; ++a) { const double t = common * (double)del[a]; gsum += t; ga[a][v] = (float)t; }
          g0[v] = (float)(-gsum);
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) add(oa[a] + d0, ga[a]);
        if (sym_backward) add(o0 + d0, g0);
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < AGG_HT * 4; t += 256) {       // (one lane per float: the lanes of a cell form one request)
      const int ek = keys[t >> (2 + cw_log)];
      if (ek >= 0) atomicAdd(dst + ((long long)ek << (2 + cw_log)) + (t & ((4 << cw_log) - 1)), vals[t]);
    }
    __syncthreads();
  }
}

// hash_index (voxel_hash_feature_cuda.cu:54-100): 8 hashed corner indices as floats
__global__ void __launch_bounds__(256) k_hash_index(long long P, float* __restrict__ out, const float* __restrict__ query,
                                                    GridDesc g) {
  constexpr int TOPO = HASH, I = LINEAR;
  NDJIR_GRID_THREAD_PROLOGUE
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        out[b * 8 + i * 4 + j * 2 + k] = (float)hash3(st.ax[0].idx[i], st.ax[1].idx[j], st.ax[2].idx[k], st.T);
  NDJIR_GRID_THREAD_EPILOGUE
}

__global__ void __launch_bounds__(256) k_zero(long long n, float* __restrict__ p) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  long long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
  if (aligned) {
    for (long long j = i; j < n4; j += stride) p4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long j = (n4 << 2) + i; j < n; j += stride) p[j] = 0.f;
  } else {
    for (long long j = i; j < n; j += stride) p[j] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
static inline int grid_blocks(long long threads) {
  long long b = (threads + 255) / 256;
  const long long cap = 256LL * 16;   // 256 CUs x 16 blocks; grid-stride beyond that
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

static inline int pick_vw(int D) { return (D % 4 == 0) ? 4 : ((D % 2 == 0) ? 2 : 1); }

void zero_fill(float* p, long long n, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_zero, dim3(grid_blocks((n + 3) / 4)), dim3(256), 0, stream, n, p);
}

#define NDJIR_DISPATCH_VW(VWV, ...)                    \
  switch (VWV) {                                       \
    case 4: { constexpr int VW = 4; __VA_ARGS__; } break; \
    case 2: { constexpr int VW = 2; __VA_ARGS__; } break; \
    default: { constexpr int VW = 1; __VA_ARGS__; } break; \
  }

#define NDJIR_DISPATCH_TI(TOPOV, INTERPV, ...)                                              \
  switch ((TOPOV) * 3 + (INTERPV)) {                                                        \
    case 0: { constexpr int TOPO = VOXEL, I = LINEAR; __VA_ARGS__; } break;                 \
    case 1: { constexpr int TOPO = VOXEL, I = COSINE; __VA_ARGS__; } break;                 \
    case 2: { constexpr int TOPO = VOXEL, I = LANCZOS; __VA_ARGS__; } break;                \
    case 3: { constexpr int TOPO = TRIPLANE, I = LINEAR; __VA_ARGS__; } break;              \
    case 4: { constexpr int TOPO = TRIPLANE, I = COSINE; __VA_ARGS__; } break;              \
    case 5: { constexpr int TOPO = TRIPLANE, I = LANCZOS; __VA_ARGS__; } break;             \
    case 6: { constexpr int TOPO = TRILINE, I = LINEAR; __VA_ARGS__; } break;               \
    case 7: { constexpr int TOPO = TRILINE, I = COSINE; __VA_ARGS__; } break;               \
    case 8: { constexpr int TOPO = TRILINE, I = LANCZOS; __VA_ARGS__; } break;              \
    case 9: { constexpr int TOPO = HASH, I = LINEAR; __VA_ARGS__; } break;                  \
    case 11: { constexpr int TOPO = HASH, I = LANCZOS; __VA_ARGS__; } break;                \
    default: return NDJIR_ERR_UNSUPPORTED;                                                  \
  }

int launch_query(int interp, const GridDesc& g, long long P, float* out, const float* query, const float* feature,
                 bool accum, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  static const bool no_rows = getenv("NDJIR_QUERY_NO_ROWS") != nullptr;      // A/B switch
  if ((g.topo == TRIPLANE || g.topo == TRILINE) && !accum && (g.D == 4 || g.D == 8) && !no_rows &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(feature) & 15) == 0) {
    // one lane per point emits the point's contiguous (D, 3) row
    const int rb = grid_blocks(P);
#define NDJIR_ROWS(T, II, DD) hipLaunchKernelGGL((k_query_rows<T, II, DD>), dim3(rb), dim3(256), 0, stream, P, out, query, feature, g)
#define NDJIR_ROWS_I(T, DD) { if (interp == LINEAR) NDJIR_ROWS(T, LINEAR, DD); else if (interp == COSINE) NDJIR_ROWS(T, COSINE, DD); else NDJIR_ROWS(T, LANCZOS, DD); }
    if (g.topo == TRIPLANE) { if (g.D == 4) NDJIR_ROWS_I(TRIPLANE, 4) else NDJIR_ROWS_I(TRIPLANE, 8) }
    else { if (g.D == 4) NDJIR_ROWS_I(TRILINE, 4) else NDJIR_ROWS_I(TRILINE, 8) }
#undef NDJIR_ROWS_I
#undef NDJIR_ROWS
    return ndjir_check_launch();
  }
  int blocks = grid_blocks(P * g.S);
  NDJIR_DISPATCH_TI(g.topo, interp, NDJIR_DISPATCH_VW(pick_vw(g.D), {
    if (accum) hipLaunchKernelGGL((k_query<TOPO, I, VW, true>), dim3(blocks), dim3(256), 0, stream, P, out, query, feature, g);
    else hipLaunchKernelGGL((k_query<TOPO, I, VW, false>), dim3(blocks), dim3(256), 0, stream, P, out, query, feature, g);
  }))
  return ndjir_check_launch();
}

// mode 0: grad_query (dst = gq (P,3), src = grad_output); mode 1: ggo (dst = (P,C), src = gg_query)
int launch_dquery(int interp, const GridDesc& g, long long P, int mode, float* dst, const float* src, const float* query,
                  const float* feature, bool accum, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  static const bool no_point = getenv("NDJIR_HASH_DQUERY_ATOMIC") != nullptr;    // A/B switch
  if (mode == 0 && g.topo == HASH && !no_point) {
    // one lane per point sums the levels on chip: no atomics, one write per point and axis
    const int pb = grid_blocks(P);
    const int vw = pick_vw(g.D);
#define NDJIR_HP(II, VV) hipLaunchKernelGGL((k_dquery_hash_point<II, VV>), dim3(pb), dim3(256), 0, stream, P, dst, src, query, feature, g, accum ? 1 : 0)
#define NDJIR_HP_V(II) { if (vw == 4) NDJIR_HP(II, 4); else if (vw == 2) NDJIR_HP(II, 2); else NDJIR_HP(II, 1); }
    if (interp == COSINE) return NDJIR_ERR_UNSUPPORTED;            // (no cosine hash family)
    if (interp == LANCZOS) NDJIR_HP_V(LANCZOS) else NDJIR_HP_V(LINEAR)
#undef NDJIR_HP_V
#undef NDJIR_HP
    return ndjir_check_launch();
  }
  static const bool no_rows = getenv("NDJIR_QUERY_NO_ROWS") != nullptr;      // A/B switch
  if ((g.topo == TRIPLANE || g.topo == TRILINE) && (g.D == 4 || g.D == 8) && !no_rows &&
      (reinterpret_cast<uintptr_t>(mode == 0 ? (const void*)src : (const void*)dst) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(feature) & 15) == 0) {
    const int rb = grid_blocks(P);
    const int acc = accum ? 1 : 0;
#define NDJIR_DR(T, II, DD, MM) hipLaunchKernelGGL((k_dquery_rows<T, II, DD, MM>), dim3(rb), dim3(256), 0, stream, P, dst, src, query, feature, g, acc)
#define NDJIR_DR_M(T, II, DD) { if (mode == 0) NDJIR_DR(T, II, DD, 0); else NDJIR_DR(T, II, DD, 1); }
#define NDJIR_DR_I(T, DD) { if (interp == LINEAR) NDJIR_DR_M(T, LINEAR, DD) else if (interp == COSINE) NDJIR_DR_M(T, COSINE, DD) else NDJIR_DR_M(T, LANCZOS, DD) }
    if (g.topo == TRIPLANE) { if (g.D == 4) NDJIR_DR_I(TRIPLANE, 4) else NDJIR_DR_I(TRIPLANE, 8) }
    else { if (g.D == 4) NDJIR_DR_I(TRILINE, 4) else NDJIR_DR_I(TRILINE, 8) }
#undef NDJIR_DR_I
#undef NDJIR_DR_M
#undef NDJIR_DR
    return ndjir_check_launch();
  }
  if (mode == 0 && !accum) zero_fill(dst, P * 3, stream);
  static const bool lz_point = getenv("NDJIR_LANCZOS_DQUERY_PER_POINT") != nullptr;      // A/B switch
  if (g.topo == VOXEL && interp == LANCZOS && g.D == 4 && !lz_point) {
    const long long want = (P + 63) / 64;
    const dim3 lb((unsigned)(want > 8192 ? 8192 : want));
    if (mode == 0) hipLaunchKernelGGL((k_dquery_lanczos_voxel<0, true>), lb, dim3(256), 0, stream, P, dst, src, query, feature, g);
    else if (accum) hipLaunchKernelGGL((k_dquery_lanczos_voxel<1, true>), lb, dim3(256), 0, stream, P, dst, src, query, feature, g);
    else hipLaunchKernelGGL((k_dquery_lanczos_voxel<1, false>), lb, dim3(256), 0, stream, P, dst, src, query, feature, g);
    return ndjir_check_launch();
  }
  int blocks = grid_blocks(P * g.S);
  NDJIR_DISPATCH_TI(g.topo, interp, NDJIR_DISPATCH_VW(pick_vw(g.D), {
    if (mode == 0) hipLaunchKernelGGL((k_dquery<TOPO, I, VW, 0, true>), dim3(blocks), dim3(256), 0, stream, P, dst, src, query, feature, g);
    else if (accum) hipLaunchKernelGGL((k_dquery<TOPO, I, VW, 1, true>), dim3(blocks), dim3(256), 0, stream, P, dst, src, query, feature, g);
    else hipLaunchKernelGGL((k_dquery<TOPO, I, VW, 1, false>), dim3(blocks), dim3(256), 0, stream, P, dst, src, query, feature, g);
  }))
  return ndjir_check_launch();
}

static inline long long NTapsOf(int interp) { return interp == LANCZOS ? 64 : 8; }

// mode 0: grad_feature ; mode 1: grad_query_grad_feature.  The caller zero-fills when !accum.
int launch_scatter(int interp, const GridDesc& g, long long P, int mode, float* gf, const float* gg_query,
                   const float* grad_output, const float* query, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  int blocks = grid_blocks(P * g.S);
  static const bool no_agg = getenv("NDJIR_SCATTER_NO_AGG") != nullptr;      // A/B switch
  // workgroup-aggregated path (dense cell index must fit 31 bits: 2^33 floats); insertions per point = taps x D / 4
  const int nd = g.topo == VOXEL ? 3 : (g.topo == TRIPLANE ? 2 : 1), nt = interp == LANCZOS ? 4 : 2;
  int taps = 1;
  for (int a = 0; a < nd; ++a) taps *= nt;
  const int per_point = taps * (g.D / 4);
  if (g.topo == TRILINE && g.D <= 8 && g.G[0] * g.D <= HASH_LDS_FLOATS && P >= 64LL * g.G[0] && !no_agg) {
    long long shares = 170;                                           // 3 lines x 170 = 510 workgroups
    // >= `div` points per cell of the line and workgroup.  Round 5: 8 -> 2.  At the step's 131 072 points and G = 2048 the old
    // bound left 8 shares = 24 workgroups on a 256-CU chip (382 us per launch, 0.02 of the HBM roof: 16 384 points x 16 LDS
    // atomics each, serially per workgroup); the flush it was protecting is cheap -- a line image is 2 048 requests of 32 bytes
    static const long long div = [] { const char* e = getenv("NDJIR_LINE_SHARE_POINTS"); const int v = e ? atoi(e) : 2; return (long long)(v > 0 ? v : 2); }();
    const long long max_shares = P / (div * g.G[0]);
    if (shares > max_shares) shares = max_shares;
    if (shares < 1) shares = 1;
    const size_t lds = (size_t)g.G[0] * g.D * sizeof(float);
    static bool attr = false;
    if (!attr) {
      const int mx = HASH_LDS_FLOATS * (int)sizeof(float);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<LINEAR, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<LINEAR, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<COSINE, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<COSINE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<LANCZOS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_line_lds<LANCZOS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, mx);
      attr = true;
    }
#define NDJIR_LINE_LDS(IV) { if (mode == 0) hipLaunchKernelGGL((k_scatter_line_lds<IV, 0>), dim3((unsigned)(3 * shares)), dim3(256), lds, stream, P, gf, gg_query, grad_output, query, g, (int)shares); \
                            else hipLaunchKernelGGL((k_scatter_line_lds<IV, 1>), dim3((unsigned)(3 * shares)), dim3(256), lds, stream, P, gf, gg_query, grad_output, query, g, (int)shares); }
    if (interp == LINEAR) NDJIR_LINE_LDS(LINEAR)
    else if (interp == COSINE) NDJIR_LINE_LDS(COSINE)
    else NDJIR_LINE_LDS(LANCZOS)
#undef NDJIR_LINE_LDS
    return ndjir_check_launch();
  }
  // (Round 2 also binned large spread point sets by tiles -- tri-plane, Lanczos voxel -- so that every cell had one owner;
  // with the request-shaped flushes of round 3 the aggregated path is within 10 % of it on the tri-plane, a training step
  // never reached its threshold, and its library-owned scratch was the one piece of cross-stream state: removed in round 4.)
  static const bool lanczos_agg = getenv("NDJIR_LANCZOS_AGG") != nullptr;      // A/B switch: the generic table kernel
  if (g.topo == VOXEL && interp == LANCZOS && g.D == 4 && !no_agg && !lanczos_agg) {
    const long long want = (P + LZ_PTS - 1) / LZ_PTS;
    const int lblocks = (int)(want > 8192 ? 8192 : want);
    if (mode == 0) hipLaunchKernelGGL((k_scatter_lanczos_voxel<0>), dim3(lblocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g);
    else hipLaunchKernelGGL((k_scatter_lanczos_voxel<1>), dim3(lblocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g);
    return ndjir_check_launch();
  }
  const int lpp = taps / nt;                 // one lane per run along the fastest axis
  if (g.topo != HASH && (g.D & 3) == 0 && per_point <= 128 && !no_agg) {
    int ppp = 256 / lpp;
    while (ppp * per_point > AGG_OVER) ppp >>= 1;               // every chunk of a pass must fit the overflow list
    const long long want = (P * g.S + ppp - 1) / ppp;
    const int ablocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
#define NDJIR_AGG_CASE(T, IV)                                                                                              \
    { if (mode == 0) hipLaunchKernelGGL((k_scatter_agg<T, IV, 0>), dim3(ablocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g, ppp); \
      else hipLaunchKernelGGL((k_scatter_agg<T, IV, 1>), dim3(ablocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g, ppp); }
#define NDJIR_AGG_TOPO(IV)                                   \
    { if (g.topo == VOXEL) NDJIR_AGG_CASE(VOXEL, IV)          \
      else if (g.topo == TRIPLANE) NDJIR_AGG_CASE(TRIPLANE, IV) \
      else NDJIR_AGG_CASE(TRILINE, IV) }
    if (interp == LINEAR) NDJIR_AGG_TOPO(LINEAR)
    else if (interp == COSINE) NDJIR_AGG_TOPO(COSINE)
    else NDJIR_AGG_TOPO(LANCZOS)
#undef NDJIR_AGG_TOPO
#undef NDJIR_AGG_CASE
    return ndjir_check_launch();
  }
  if (g.topo == HASH && (interp == LINEAR || interp == LANCZOS) && !no_agg) {
    int Tmax = 0;
    for (int l = 0; l < g.S; ++l) Tmax = g.lvlT[l] > Tmax ? g.lvlT[l] : Tmax;
    const int per_slice = HASH_LDS_FLOATS / g.D;                   // entries
    const int slices = (Tmax + per_slice - 1) / per_slice;
    // enough points per workgroup that the slice flush (per_slice x D global atomics) stays small beside the taps it replaces
    if (g.D <= 8 && slices <= HASH_LDS_MAX_SLICES && P * NTapsOf(interp) >= 8LL * per_slice) {
      long long shares = 1024 / ((long long)g.S * slices);
      const long long max_shares = (P + 2047) / 2048;              // at least 2048 points per workgroup
      if (shares > max_shares) shares = max_shares;
      if (shares < 1) shares = 1;
      const int nblk = (int)(g.S * slices * shares);
      const size_t lds = (size_t)HASH_LDS_FLOATS * sizeof(float);
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_hash_lds<LINEAR, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_hash_lds<LINEAR, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_hash_lds<LANCZOS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scatter_hash_lds<LANCZOS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
      }
#define NDJIR_HASH_LDS(IV, MV) hipLaunchKernelGGL((k_scatter_hash_lds<IV, MV>), dim3(nblk), dim3(256), lds, stream, P, gf, gg_query, \
                                                  grad_output, query, g, slices, (int)shares, per_slice)
      if (interp == LINEAR) { if (mode == 0) NDJIR_HASH_LDS(LINEAR, 0); else NDJIR_HASH_LDS(LINEAR, 1); }
      else { if (mode == 0) NDJIR_HASH_LDS(LANCZOS, 0); else NDJIR_HASH_LDS(LANCZOS, 1); }
#undef NDJIR_HASH_LDS
      return ndjir_check_launch();
    }
  }
  NDJIR_DISPATCH_TI(g.topo, interp, NDJIR_DISPATCH_VW(pick_vw(g.D), {
    if (mode == 0) hipLaunchKernelGGL((k_scatter<TOPO, I, VW, 0>), dim3(blocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g);
    else hipLaunchKernelGGL((k_scatter<TOPO, I, VW, 1>), dim3(blocks), dim3(256), 0, stream, P, gf, gg_query, grad_output, query, g);
  }))
  return ndjir_check_launch();
}

int launch_mark_touched(const GridDesc& g, long long P, const float* query, unsigned* bitmap, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (g.topo != VOXEL || g.D != 4) return NDJIR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_mark_touched, dim3(grid_blocks(P * g.S)), dim3(256), 0, stream, P, query, g, bitmap);
  return ndjir_check_launch();
}

int launch_pack_rows(int interp, const GridDesc& g, long long P, const float* gf, const float* query, unsigned* bitmap, int* ids,
                     float* rows, int* count, int capacity, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (g.topo == HASH || (g.D != 4 && g.D != 8)) return NDJIR_ERR_UNSUPPORTED;
  const int blocks = grid_blocks(P * g.S);
  NDJIR_DISPATCH_TI(g.topo, interp, {
    if (g.D == 4) hipLaunchKernelGGL((k_pack_rows<TOPO, I, 1>), dim3(blocks), dim3(256), 0, stream, P, gf, query, g, bitmap, ids,
                                     reinterpret_cast<float4*>(rows), count, capacity);
    else hipLaunchKernelGGL((k_pack_rows<TOPO, I, 2>), dim3(blocks), dim3(256), 0, stream, P, gf, query, g, bitmap, ids,
                            reinterpret_cast<float4*>(rows), count, capacity);
  })
  return ndjir_check_launch();
}

int launch_voxel_query_encode(int interp, const GridDesc& g, long long P, int M, const float* query, const float* feature, float* e,
                              int lde, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (g.topo != VOXEL) return NDJIR_ERR_UNSUPPORTED;
  const int blocks = grid_blocks(P * (3 + 6 * M + g.D));
  if (interp == LINEAR) hipLaunchKernelGGL((k_voxel_query_encode<LINEAR>), dim3(blocks), dim3(256), 0, stream, P, M, query, feature, g, e, lde);
  else if (interp == COSINE) hipLaunchKernelGGL((k_voxel_query_encode<COSINE>), dim3(blocks), dim3(256), 0, stream, P, M, query, feature, g, e, lde);
  else if (interp == LANCZOS) {
    static const bool per_column = getenv("NDJIR_LANCZOS_ENCODE_PER_COLUMN") != nullptr;      // A/B switch
    // 64 points per workgroup; 16 while that leaves the chip short of workgroups (the sampler's rounds of 8 192 points: 18 -> 9 us)
    const long long want = (P + 15) / 16, want64 = (P + 63) / 64;
    if (per_column) hipLaunchKernelGGL((k_voxel_query_encode<LANCZOS>), dim3(blocks), dim3(256), 0, stream, P, M, query, feature, g, e, lde);
    else if (want64 >= 512) hipLaunchKernelGGL((k_voxel_query_encode_lanczos<64>), dim3((unsigned)(want64 > 8192 ? 8192 : want64)), dim3(256), 0, stream, P, M, query, feature, g, e, lde);
    else hipLaunchKernelGGL((k_voxel_query_encode_lanczos<16>), dim3((unsigned)want), dim3(256), 0, stream, P, M, query, feature, g, e, lde);
  }
  else return NDJIR_ERR_UNSUPPORTED;
  return ndjir_check_launch();
}

int launch_tri_query_encode(int interp, const GridDesc& gp, const GridDesc& gl, long long P, int M, const float* query,
                            const float* plane, const float* line, float* e, int lde, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (gp.topo != TRIPLANE || gl.topo != TRILINE) return NDJIR_ERR_UNSUPPORTED;
  const int blocks = grid_blocks(P * (3 + 6 * M + 3 * gp.D + 3 * gl.D));
  if (interp == LINEAR) hipLaunchKernelGGL((k_tri_query_encode<LINEAR>), dim3(blocks), dim3(256), 0, stream, P, M, query, plane, gp, line, gl, e, lde);
  else if (interp == COSINE) hipLaunchKernelGGL((k_tri_query_encode<COSINE>), dim3(blocks), dim3(256), 0, stream, P, M, query, plane, gp, line, gl, e, lde);
  else if (interp == LANCZOS) hipLaunchKernelGGL((k_tri_query_encode<LANCZOS>), dim3(blocks), dim3(256), 0, stream, P, M, query, plane, gp, line, gl, e, lde);
  else return NDJIR_ERR_UNSUPPORTED;
  return ndjir_check_launch();
}

int launch_zero_touched(int interp, const GridDesc& g, long long P, float* gf, const float* query, int* nonfinite_flag,
                        hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  int blocks = grid_blocks(P * g.S);
  if (nonfinite_flag) {
    NDJIR_DISPATCH_TI(g.topo, interp, NDJIR_DISPATCH_VW(pick_vw(g.D), {
      hipLaunchKernelGGL((k_zero_touched<TOPO, I, VW, true>), dim3(blocks), dim3(256), 0, stream, P, gf, query, g, nonfinite_flag);
    }))
  } else {
    NDJIR_DISPATCH_TI(g.topo, interp, NDJIR_DISPATCH_VW(pick_vw(g.D), {
      hipLaunchKernelGGL((k_zero_touched<TOPO, I, VW, false>), dim3(blocks), dim3(256), 0, stream, P, gf, query, g, nonfinite_flag);
    }))
  }
  return ndjir_check_launch();
}

int launch_voxel_gq_gq(const GridDesc& g, long long P, float* gq, const float* gg_query, const float* grad_output,
                       const float* query, const float* feature, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  int blocks = grid_blocks(P);
  NDJIR_DISPATCH_VW(pick_vw(g.D), {
    hipLaunchKernelGGL((k_voxel_gq_gq<VW>), dim3(blocks), dim3(256), 0, stream, P, gq, gg_query, grad_output, query, feature, g);
  })
  return ndjir_check_launch();
}

int launch_tv(const GridDesc& g, long long P, bool bwd, float* dst, const float* grad_output, const float* query,
              const float* feature, int sym_backward, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  int blocks = grid_blocks(P * g.S);
  static const bool no_agg = getenv("NDJIR_SCATTER_NO_AGG") != nullptr;
  if (bwd && (g.D == 4 || (g.topo != VOXEL && g.D == 8)) && g.topo != HASH && !no_agg) {
    if (g.topo == VOXEL) hipLaunchKernelGGL((k_tv_bwd_agg<VOXEL>), dim3(blocks), dim3(256), 0, stream, P, dst, grad_output, query, feature, g, sym_backward);
    else if (g.topo == TRIPLANE) hipLaunchKernelGGL((k_tv_bwd_agg<TRIPLANE>), dim3(blocks), dim3(256), 0, stream, P, dst, grad_output, query, feature, g, sym_backward);
    else hipLaunchKernelGGL((k_tv_bwd_agg<TRILINE>), dim3(blocks), dim3(256), 0, stream, P, dst, grad_output, query, feature, g, sym_backward);
    return ndjir_check_launch();
  }
#define NDJIR_TV_CASE(T)                                                                                         \
  NDJIR_DISPATCH_VW(pick_vw(g.D), {                                                                              \
    if (bwd) hipLaunchKernelGGL((k_tv<T, VW, true>), dim3(blocks), dim3(256), 0, stream, P, dst, grad_output, query, feature, g, sym_backward); \
    else hipLaunchKernelGGL((k_tv<T, VW, false>), dim3(blocks), dim3(256), 0, stream, P, dst, grad_output, query, feature, g, sym_backward);    \
  })
  switch (g.topo) {
    case VOXEL: NDJIR_TV_CASE(VOXEL) break;
    case TRIPLANE: NDJIR_TV_CASE(TRIPLANE) break;
    case TRILINE: NDJIR_TV_CASE(TRILINE) break;
    case HASH: NDJIR_TV_CASE(HASH) break;
    default: return NDJIR_ERR_UNSUPPORTED;
  }
#undef NDJIR_TV_CASE
  return ndjir_check_launch();
}

int launch_hash_index(const GridDesc& g, long long P, float* out, const float* query, hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  hipLaunchKernelGGL(k_hash_index, dim3(grid_blocks(P)), dim3(256), 0, stream, P, out, query, g);
  return ndjir_check_launch();
}

}  // namespace ndjir
