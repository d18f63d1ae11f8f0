// mlp.hip -- fused whole-MLP "chain" kernel on the fp32 matrix cores of gfx950.
//
// The reference evaluates every small MLP (python/network.py) as a sequence of cuBLAS affine
// calls with separate elementwise kernels in between.  Here ONE launch pushes a tile of TM = 64
// points through ALL layers of a net: activations never leave the CU (LDS ping-pong), weights
// stream from L2 in MFMA-fragment order, bias + softplus (forward) or the softplus-derivative
// product (backward) are fused into the epilogue of each layer.
//
//   forward  chain:  h_l = softplus_beta(h_{l-1} W_l + b_l),  y = h_{L-1} W_L + b_L
//   backward chain:  delta_{l-1} = (delta_l W_l^T) * softplus'(z_{l-1}),  softplus'(z) = 1 - exp(-beta h)
//                    (needs only the stored forward activations h, not z)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains; 64 FLOP/clk/SIMD = the fp32 peak).
// Tiling: 512 threads = 8 waves; a wave owns both 32-row blocks of one or more 32-column blocks,
// i.e. 2 accumulators (32 VGPRs) per column block.  Per 8 k-values a wave issues 2 ds_read_b128
// (A fragments, conflict-free layout), 1 global_load_dwordx4 (B fragment, 1 KiB coalesced) and
// 8 MFMAs (512 matrix-pipe cycles): the matrix pipe is the only busy unit by construction.
//
// LDS activation layout: element (row m, feature k) at dword (k>>2)*GP + m*4 + (k&3), GP = 4*TM+4.
//   - A-fragment of MFMA step j for lane (r = lane&31, h = lane>>5) is feature k = 8*kb + 4*h + j:
//     the 4 steps of a k-block are ONE 16-byte read; 16-lane groups hit 16 distinct 4-bank slots.
//   - The +4 pad makes the epilogue's ds_write_b32 of a 32x32 accumulator conflict-free.
// Packed weight layout (pack kernel below): Wp[nb][kb][lane][j] = W[8*kb + 4*(lane>>5) + j][32*nb + (lane&31)].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "mlp.h"

namespace ndjir {

constexpr int TM = 64;
constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
constexpr int GP = TM * 4 + 4;   // dwords per group of 4 features

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float softplus_beta(float z, float beta) {
  float bz = beta * z;
  return bz > 20.f ? z : log1pf(__expf(bz)) / beta;
}

// row of accumulator register i for half-wave h (standard 32x32 C/D map)
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- weight packing -----------------------------------------------------------------------------
// src: W (K x N) row-major (nnabla layout, y = x W).  transpose=1 packs W^T (N x K) instead.
// dst: [Np/32][Kp/8][64][4], zero padded.  (Kp, Np) are the padded dims of the packed matrix.
__global__ void __launch_bounds__(256) k_pack(const float* __restrict__ W, float* __restrict__ dst, int K, int N,
                                              int transpose, int Kp, int Np) {
  long long total = (long long)Kp * Np;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int j = (int)(t & 3);
    int lane = (int)((t >> 2) & 63);
    long long rest = t >> 8;
    int KB = Kp >> 3;
    int kb = (int)(rest % KB);
    int nb = (int)(rest / KB);
    int k = kb * 8 + 4 * (lane >> 5) + j;
    int n = nb * 32 + (lane & 31);
    float v = 0.f;
    if (!transpose) { if (k < K && n < N) v = W[(long long)k * N + n]; }
    else { if (k < N && n < K) v = W[(long long)n * N + k]; }   // packed matrix = W^T: rows index N, cols index K
    dst[t] = v;
  }
}

// ---- the chain kernel ---------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(NTHREADS, 2) k_mlp_chain(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;                             // input-side buffer (sized for Kmax0)
  float* bufB = lds + (size_t)(a.lds_split);     // second buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const float beta = a.beta;

  for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = tile * TM;
    const int rows = (int)((a.P - row0) < TM ? (a.P - row0) : TM);

    // ---- load the chain input tile into bufA (zero padded to K0p) ----
    {
      const int K0p = a.K0p, K0 = a.K0;
      const float* X = a.X + row0 * a.ldx;
      const int groups = K0p >> 2;
      if ((a.ldx & 3) == 0 && ((uintptr_t)a.X & 15) == 0) {
        for (int t = tid; t < groups * TM; t += NTHREADS) {
          int g = t % groups, m = t / groups;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (m < rows) {
            int k = g * 4;
            if (k + 3 < K0) v = *reinterpret_cast<const f32x4*>(X + (long long)m * a.ldx + k);
            else { for (int q = 0; q < 4; ++q) if (k + q < K0) v[q] = X[(long long)m * a.ldx + k + q]; }
          }
          *reinterpret_cast<f32x4*>(bufA + g * GP + m * 4) = v;
        }
      } else {
        for (int t = tid; t < K0p * TM; t += NTHREADS) {
          int k = t % K0p, m = t / K0p;
          float v = (m < rows && k < K0) ? X[(long long)m * a.ldx + k] : 0.f;
          bufA[(k >> 2) * GP + m * 4 + (k & 3)] = v;
        }
      }
    }
    __syncthreads();

    float* cur = bufA;
    float* nxt = bufB;
    for (int li = 0; li < a.L; ++li) {
      const ChainLayer& ly = a.layers[li];
      const int KB = ly.Kp >> 3;
      const int NB = ly.Np >> 5;
      const bool last = a.has_output && (li == a.L - 1);

      if (NB == 1) {
        // ---- narrow output (N <= 32): split K over 4 wave groups, reduce through LDS ----
        const int rb = wave & 1, ks = wave >> 1;
        const int kb0 = (KB * ks) / 4, kb1 = (KB * (ks + 1)) / 4;
        f32x16 acc = {0};
        const f32x4* Bp = reinterpret_cast<const f32x4*>(ly.Wp) + lane;
        for (int kb = kb0; kb < kb1; ++kb) {
          f32x4 b = Bp[(long long)kb * 64];
          f32x4 av = *reinterpret_cast<const f32x4*>(cur + (kb * 2 + h) * GP + (rb * 32 + r) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[j], acc, 0, 0, 0);
        }
        // partials: nxt[ks][m][n] (4 x 64 x 32 floats = 32 KiB)
#pragma unroll
        for (int i = 0; i < 16; ++i) nxt[(ks * TM + rb * 32 + acc_row(i, h)) * 32 + r] = acc[i];
        __syncthreads();
        for (int t = tid; t < TM * 32; t += NTHREADS) {
          int n = t & 31, m = t >> 5;
          float z = nxt[t] + nxt[TM * 32 + t] + nxt[2 * TM * 32 + t] + nxt[3 * TM * 32 + t];
          if (n < ly.N && m < rows) {
            if (!BWD) {
              z += ly.bias ? ly.bias[n] : 0.f;
              if (!last) z = softplus_beta(z, beta);
            }
            if (last) {
              float* y = a.Y + (row0 + m) * a.ldy + n;
              *y = a.accum_y ? *y + z : z;
            }
          }
          // a narrow layer is always the last one of the chains this kernel serves
        }
        __syncthreads();
        continue;
      }

      // ---- general layer: each wave owns column blocks nb = wave, wave + 8, ... ----
      for (int nb = wave; nb < NB; nb += NWAVES) {
        f32x16 acc0 = {0}, acc1 = {0};
        const f32x4* Bp = reinterpret_cast<const f32x4*>(ly.Wp) + (long long)nb * KB * 64 + lane;
        const float* A0 = cur + h * GP + r * 4;
        f32x4 b = Bp[0];
        f32x4 a0 = *reinterpret_cast<const f32x4*>(A0);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(A0 + 32 * 4);
        for (int kb = 0; kb < KB; ++kb) {
          f32x4 bn = b, a0n = a0, a1n = a1;
          if (kb + 1 < KB) {                     // prefetch the next k-block
            bn = Bp[(long long)(kb + 1) * 64];
            const float* An = A0 + (kb + 1) * 2 * GP;
            a0n = *reinterpret_cast<const f32x4*>(An);
            a1n = *reinterpret_cast<const f32x4*>(An + 32 * 4);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[j], acc1, 0, 0, 0);
          }
          b = bn; a0 = a0n; a1 = a1n;
        }

        // ---- epilogue, pass 1: raw accumulators -> LDS (activation layout, conflict-free) ----
        {
          // the output layer only stages through LDS: its blocks use a compact per-wave slot
          const int n = (last ? wave : nb) * 32 + r;
          float* dst = nxt + (n >> 2) * GP + (n & 3);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            dst[acc_row(i, h) * 4] = acc0[i];
            dst[(32 + acc_row(i, h)) * 4] = acc1[i];
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- pass 2: this wave's 64 x 32 block, one float4 (4 columns of one row) per lane-step ----
        {
          const int g = lane & 7;                 // column group inside the block
          const int n4 = nb * 32 + g * 4;         // first of the 4 columns
          f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
          if (!BWD && ly.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n4 + q < ly.N) bias4[q] = ly.bias[n4 + q];
          }
          f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
          const bool is_skip = (li == a.skip_layer);
          const float sc = is_skip ? a.skip_scale : 1.f;
          for (int it = 0; it < 8; ++it) {
            const int m = (lane >> 3) + 8 * it;
            float* lp = nxt + ((last ? wave * 32 + g * 4 : n4) >> 2) * GP + m * 4;
            f32x4 z = *reinterpret_cast<f32x4*>(lp);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const bool mrow = m < rows;
            const long long grow = row0 + m;
            if (!BWD) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float t = z[q] + bias4[q];
                if (!last) t = softplus_beta(t, beta) * sc;
                v[q] = (n4 + q < ly.N && mrow) ? t : 0.f;
              }
              if (mrow) {
                if (last) {
                  float* y = a.Y + grow * a.ldy + n4;
#pragma unroll
                  for (int q = 0; q < 4; ++q) if (n4 + q < ly.N) y[q] = a.accum_y ? y[q] + v[q] : v[q];
                } else if (ly.side_out) {
                  float* o = ly.side_out + grow * ly.ld_side + n4;
                  if (n4 + 3 < ly.N && (ly.ld_side & 3) == 0) *reinterpret_cast<f32x4*>(o) = v;
                  else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (n4 + q < ly.N) o[q] = v[q];
                  }
                }
              }
            } else if (last) {
              if (mrow) {
                float* y = a.Y + grow * a.ldy + n4;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (n4 + q < ly.N) y[q] = a.accum_y ? y[q] + z[q] : z[q];
              }
            } else {
              // this GEMM produced dL/dh of the layer below; its stored activation gives softplus'
              if (mrow) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int n = n4 + q;
                  if (n >= ly.N) continue;
                  if (is_skip && n >= a.skip_split) {
                    // gradient of the concatenated chain input: stash (scaled) for the final dL/dX
                    if (a.Xskip) a.Xskip[grow * a.ld_xskip + (n - a.skip_split)] = z[q] * sc;
                    continue;
                  }
                  // stored activation is h * skip_scale on the skip layer
                  float hs = ly.side_in[grow * ly.ld_side + n];
                  float sp = (1.f - __expf(-beta * hs / sc)) * sc;
                  v[q] = z[q] * sp;
                }
                if (ly.side_out) {
                  const int nlim = is_skip ? a.skip_split : ly.N;
                  float* o = ly.side_out + grow * ly.ld_side + n4;
                  if (n4 + 3 < nlim && (ly.ld_side & 3) == 0) *reinterpret_cast<f32x4*>(o) = v;
                  else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (n4 + q < nlim) o[q] = v[q];
                  }
                }
              }
              colsum += v;
            }
            if (!last) *reinterpret_cast<f32x4*>(lp) = v;
          }
          if (BWD && !last && ly.bgrad) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float c = colsum[q];
              c += __shfl_xor(c, 8);
              c += __shfl_xor(c, 16);
              c += __shfl_xor(c, 32);
              if (lane < 8 && n4 + q < (is_skip ? a.skip_split : ly.N)) atomicAdd(ly.bgrad + n4 + q, c);
            }
          }
        }
      }

      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output ----
      if (!BWD && li == a.skip_layer) {
        __syncthreads();   // the epilogues above zero-filled the padding columns this overwrites
        const int K0 = a.K0, base = ly.N;
        const float* X = a.X + row0 * a.ldx;
        for (int t = tid; t < K0 * TM; t += NTHREADS) {
          int k = t % K0, m = t / K0;
          float v = (m < rows) ? X[(long long)m * a.ldx + k] * a.skip_scale : 0.f;
          int kk = base + k;
          nxt[(kk >> 2) * GP + m * 4 + (kk & 3)] = v;
          if (m < rows && ly.side_out) ly.side_out[(row0 + m) * ly.ld_side + kk] = v;
        }
      }
      __syncthreads();
      float* t = cur; cur = nxt; nxt = t;
    }
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

int launch_pack(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream) {
  int Kp = round_up(transpose ? N : K, 8), Np = round_up(transpose ? K : N, 32);
  long long total = (long long)Kp * Np;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, stream, W, dst, K, N, transpose, Kp, Np);
  return ndjir_check_launch();
}

int launch_chain(const ChainArgs& a, bool bwd, hipStream_t stream) {
  if (a.P <= 0) return NDJIR_OK;
  // LDS: bufA holds the chain input (K0p) or any hidden width; bufB any hidden width / partials
  int wmax = 0;
  // every general-path layer (incl. the output layer: its epilogue also passes through LDS)
  for (int i = 0; i < a.L; ++i) if (a.layers[i].Np > 32) {
    int w = a.layers[i].Np;
    if (a.has_output && i == a.L - 1 && w > NWAVES * 32) w = NWAVES * 32;   // output layer: per-wave slots
    if (w > wmax) wmax = w;
  }
  if (a.skip_layer >= 0 && !bwd) { int w = round_up(a.layers[a.skip_layer].N + a.K0, 8); if (w > wmax) wmax = w; }
  int wa = a.K0p > wmax ? a.K0p : wmax;
  size_t szA = (size_t)(wa / 4) * GP * 4;
  size_t szB = (size_t)(wmax / 4) * GP * 4;
  const size_t partials = (size_t)4 * TM * 32 * 4;                     // narrow-layer partial sums
  if (szA < partials) szA = partials;
  if (szB < partials) szB = partials;
  ChainArgs b = a;
  b.lds_split = (int)(szA / 4);
  b.n_tiles = (a.P + TM - 1) / TM;
  size_t lds_bytes = szA + szB;
  if (lds_bytes > 160 * 1024) return NDJIR_ERR_UNSUPPORTED;
  long long blocks = b.n_tiles;
  if (blocks > 256LL * 8) blocks = 256LL * 8;
  static bool attr_set[2] = {false, false};
  if (!attr_set[bwd]) {
    if (bwd) hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    else hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set[bwd] = true;
  }
  if (bwd) hipLaunchKernelGGL((k_mlp_chain<true>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, b);
  else hipLaunchKernelGGL((k_mlp_chain<false>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, b);
  return ndjir_check_launch();
}

}  // namespace ndjir
