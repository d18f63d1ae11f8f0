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
#include <stdio.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "mlp.h"

namespace ndjir {

constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float softplus_beta(float z, float beta) {
  float bz = beta * z;
  return bz > 20.f ? z : log1pf(__expf(bz)) / beta;
}

// keep a wave-uniform value in scalar registers (opaque to rematerialisation)
__device__ __forceinline__ int pin(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ float pin(float v) { return __int_as_float(pin(__float_as_int(v))); }
// pointers from the argument block: keep them in SGPRs AND in the global address space (a pointer that
// went through an asm barrier is generic to the compiler -> flat_load/flat_store, which also count
// on lgkmcnt and so serialise with every LDS wait)
template <class T>
using gptr = T __attribute__((address_space(1)))*;
template <class T>
__device__ __forceinline__ gptr<T> pin(T* p) {
  asm volatile("" : "+s"(p));
  return (gptr<T>)p;
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
// MODE 0: forward (bias + softplus).  MODE 1: backward data path (x softplus' from the stored
// activation, + optional extra adjoint).  MODE 2: tangent chain of the double backward (forward
// direction, no bias: x softplus', and emits beta * z * s * exp(-beta h) as the extra adjoint).
template <int MODE, int TM>
__global__ void __launch_bounds__(NTHREADS, ((TM == 64 || MODE != 0) ? 2 : 4)) k_mlp_chain(ChainArgs a) {
  constexpr bool BWD = (MODE == 1);
  constexpr int GP = TM * 4 + 4;   // dwords per group of 4 features
  constexpr int RB = TM / 32;      // 32-row blocks per tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bufA = lds;                             // input-side buffer (sized for Kmax0)
  float* bufB = lds + (size_t)(a.lds_split);     // second buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const float beta = a.beta;
  // diagnostics: shader-clock stamps of workgroup 0's first tile, [layer][phase 0..4][wave]
  auto stamp = [&](int li, int phase) {
    if (a.timeline && blockIdx.x == 0 && lane == 0) a.timeline[(li * 5 + phase) * NWAVES + wave] = (long long)__builtin_amdgcn_s_memtime();
  };

  // bias gradients: column sums of the deltas accumulate in LDS over all tiles of this workgroup and
  // leave as one partial row per workgroup (device-scope float atomics on a few hundred addresses
  // from every tile serialise in the memory-side cache: measured 4x on the 128-wide nets)
  float* bsum = lds + a.bg_lds;
  if (MODE != 0) for (int i = tid; i < a.bg_total; i += NTHREADS) bsum[i] = 0.f;
  if (MODE != 0 && a.in_bgrad) __syncthreads();    // the first tile's input load already accumulates into bsum

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
          if (MODE != 0 && a.in_bgrad && m < rows) {    // bias gradient of the output layer: column sums of the input
#pragma unroll
            for (int q = 0; q < 4; ++q) if (g * 4 + q < K0) atomicAdd(bsum + a.in_bg_off + g * 4 + q, v[q]);
          }
        }
      } else {
        for (int t = tid; t < K0p * TM; t += NTHREADS) {
          int k = t % K0p, m = t / K0p;
          float v = (m < rows && k < K0) ? X[(long long)m * a.ldx + k] : 0.f;
          bufA[(k >> 2) * GP + m * 4 + (k & 3)] = v;
          if (MODE != 0 && a.in_bgrad && m < rows && k < K0) atomicAdd(bsum + a.in_bg_off + k, v);
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
      stamp(li, 0);
      // Per-layer parameters -> SGPRs, once.  The argument block lives in kernarg memory: left to itself
      // the compiler re-loads fields inside the epilogue loop, and every scalar-load wait also drains
      // the LDS queue (lgkmcnt is shared).
      const gptr<const float> p_wp = pin(ly.Wp);
      const gptr<const float> p_bias = pin(ly.bias);
      // forward, first layer only: per-row-group additive term (the part of x W_0 that is constant over a group)
      const gptr<const float> p_rowbias = pin((MODE == 0 && li == 0) ? a.row_bias : nullptr);
      const int rb_div = pin(a.row_bias_div > 0 ? a.row_bias_div : 1);
      const gptr<const float> p_side_in = pin(ly.side_in);
      const gptr<const float> p_side_in2 = pin(ly.side_in2);
      const gptr<const float> p_side_add = pin(ly.side_add);
      const gptr<float> p_side_out = pin(ly.side_out);
      const gptr<float> p_side_out2 = pin(ly.side_out2);
      float* const p_bgrad = (MODE != 0 && ly.bgrad) ? bsum + pin(ly.bg_off) : nullptr;
      const int l_N = pin(ly.N);
      const int l_ld = pin(ly.ld_side);
      const bool is_skip = (li == a.skip_layer);
      const float sc = pin(is_skip ? a.skip_scale : 1.f);
      const int nlim = pin((BWD && is_skip) ? a.skip_split : l_N);   // columns that take the activation path

      if (NB == 1) {
        // ---- narrow output (N <= 32): split K over 4 wave groups, reduce through LDS ----
        // KS is the same for both tile heights (the summation order, hence the bits of the result, must
        // not depend on how many points a launch has); with 32-row tiles waves 4..7 only join the barriers
        constexpr int KS = 4;                    // K slices
        const int rb = wave % RB, ks = wave / RB;
        const int kb0 = ks < KS ? (KB * ks) / KS : 0, kb1 = ks < KS ? (KB * (ks + 1)) / KS : 0;
        f32x16 acc = {0};
        const f32x4* Bp = reinterpret_cast<const f32x4*>(ly.Wp) + lane;
        for (int kb = kb0; kb < kb1; ++kb) {
          f32x4 b = Bp[(long long)kb * 64];
          f32x4 av = *reinterpret_cast<const f32x4*>(cur + (kb * 2 + h) * GP + (rb * 32 + r) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b[j], acc, 0, 0, 0);
        }
        // partials: nxt[ks][m][n] (KS x TM x 32 floats = 32 KiB)
#pragma unroll
        for (int i = 0; i < 16; ++i) if (ks < KS) nxt[(ks * TM + rb * 32 + acc_row(i, h)) * 32 + r] = acc[i];
        __syncthreads();
        for (int t = tid; t < TM * 32; t += NTHREADS) {
          int n = t & 31, m = t >> 5;
          float z = 0.f;
#pragma unroll
          for (int q = 0; q < KS; ++q) z += nxt[q * TM * 32 + t];
          if (n < ly.N && m < rows) {
            if (MODE == 0) {
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

      // ---- general layer ----
      // A work unit = (column block nb, RBU consecutive 32-row blocks starting at rb0).  Column blocks
      // that fill whole rounds of 8 waves are processed with both row blocks by one wave (weights
      // fetched once per tile); the NB % 8 remainder blocks are split by row block over twice as many
      // waves, so e.g. a 257-wide output or a 43-wide input gradient does not leave 6-7 waves idle.
      auto process = [&](auto rbu_tag, const int nb, const int rb0) {
        constexpr int RBU = decltype(rbu_tag)::value;
        constexpr int ITS = RBU * 4;               // pass-2 steps (8 rows each)
        f32x16 acc[RBU];
#pragma unroll
        for (int q = 0; q < RBU; ++q) acc[q] = f32x16{0};
        const gptr<const f32x4> Bp = ((gptr<const f32x4>)(p_wp)) + (long long)nb * KB * 64 + lane;
        const float* A0 = cur + h * GP + (rb0 * 32 + r) * 4;
        const int mbase = rb0 * 32 + (lane >> 3);  // first row this lane handles in pass 2
        // output-layer staging slot (host sizes LDS for min(NB, 8) slots): the block's own index while
        // NB <= 8 (row-split units of one block share it on disjoint rows), else one slot per wave
        const int slot = (NB <= NWAVES) ? nb : wave;

        const int g = lane & 7;                   // column group inside the block
        const int n4 = nb * 32 + g * 4;           // first of the 4 columns
        // fast epilogue (wave-uniform): a hidden layer's full 32-column block of a full tile with
        // 16-byte aligned side rows -> no masks, no per-lane branches, vector loads/stores only
        const bool fast = !last && (nb * 32 + 31 < nlim) && (l_ld & 3) == 0 && rows == TM && (!p_rowbias || (l_N & 3) == 0);
        const long long off0 = (row0 + mbase) * l_ld + n4;    // this lane's first element in the side arrays

        // backward / tangent: the unit's stored activations are fetched from inside the k-loop (below)
        f32x4 hsv[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) hsv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_hsv = [&]() {
          if (fast) {
#pragma unroll
            for (int it = 0; it < ITS; ++it) hsv[it] = *((gptr<const f32x4>)(p_side_in + off0 + (long long)it * 8 * l_ld));
          }
        };

        // k-loop.  Weights are prefetched 4 steps ahead (an L2 round trip costs about 3 steps of matrix
        // work for the two waves of a SIMD), LDS operands one step ahead; static register slots, no
        // rotation moves.  Loads return in order, so the HBM-latency activation fetch of the
        // backward modes is issued right after the LAST weight prefetch: nothing queues behind it.
        f32x4 bq[4];
        f32x4 av[2][RBU];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bq[i] = Bp[(long long)(i < KB ? i : 0) * 64];
          __builtin_amdgcn_sched_barrier(0);     // issue order = slot order (the waits count on it)
        }
#pragma unroll
        for (int q = 0; q < RBU; ++q) { av[0][q] = *reinterpret_cast<const f32x4*>(A0 + q * 32 * 4); av[1][q] = av[0][q]; }
        __builtin_amdgcn_sched_barrier(0);
        const int hs_at = KB > 4 ? KB - 4 : 0;
        auto kstep = [&](auto itag, auto guard_tag, const int k) {
          constexpr int i = decltype(itag)::value;
          constexpr bool GUARD = decltype(guard_tag)::value;   // main loop: every prefetch is in range
          if (!GUARD || k + 1 < KB) {
            const float* An = A0 + (k + 1) * 2 * GP;
#pragma unroll
            for (int q = 0; q < RBU; ++q) av[(i + 1) & 1][q] = *reinterpret_cast<const f32x4*>(An + q * 32 * 4);
          }
          __builtin_amdgcn_sched_barrier(0);     // keep the software pipeline as written
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < RBU; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 1][q][j], bq[i][j], acc[q], 0, 0, 0);
          }
          if (!GUARD || k + 4 < KB) bq[i] = Bp[(long long)(k + 4) * 64];
          if (GUARD && MODE != 0 && !last && k == hs_at) load_hsv();
          __builtin_amdgcn_sched_barrier(0);
        };
        {
          using T = std::true_type;
          using F = std::false_type;
          int kb = 0;
          for (; kb + 8 <= KB; kb += 4) {        // k + 4 < KB for all four steps
            kstep(std::integral_constant<int, 0>{}, F{}, kb);
            kstep(std::integral_constant<int, 1>{}, F{}, kb + 1);
            kstep(std::integral_constant<int, 2>{}, F{}, kb + 2);
            kstep(std::integral_constant<int, 3>{}, F{}, kb + 3);
          }
          // last 4..7 steps (or all of a short K): guarded prefetches, activation fetch of the backward modes
          for (; kb < KB; kb += 4) {
            kstep(std::integral_constant<int, 0>{}, T{}, kb);
            if (kb + 1 < KB) kstep(std::integral_constant<int, 1>{}, T{}, kb + 1);
            if (kb + 2 < KB) kstep(std::integral_constant<int, 2>{}, T{}, kb + 2);
            if (kb + 3 < KB) kstep(std::integral_constant<int, 3>{}, T{}, kb + 3);
          }
        }

        stamp(li, 1);
        // ---- epilogue, pass 1: raw accumulators -> LDS (activation layout, conflict-free) ----
        {
          // the output layer only stages through LDS: its units use a compact slot
          const int n = (last ? slot : nb) * 32 + r;
          float* dst = nxt + (n >> 2) * GP + (n & 3) + rb0 * 32 * 4;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int q = 0; q < RBU; ++q) dst[(q * 32 + acc_row(i, h)) * 4] = acc[q][i];
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(li, 2);
        // ---- pass 2: this unit's (32 RBU) x 32 block, one float4 (4 columns of one row) per lane-step ----
        constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
        const float b2 = beta * LOG2E;             // softplus_beta(t) = log2(1 + 2^(b2 t)) * ln2 / beta
        const float ib2 = LN2 / beta;
        // stored activations are h * skip_scale on the forward skip layer
        const float hsc = (MODE != 0 && is_skip) ? 1.f / sc : 1.f;
        const float nb2 = -b2 * hsc;               // exp(-beta h) = 2^(nb2 h_stored)
        f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
        float* lp = nxt + ((last ? slot * 32 + g * 4 : n4) >> 2) * GP + mbase * 4;
        if (fast) {
          f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
          if (MODE == 0 && p_bias) bias4 = *((gptr<const f32x4>)(p_bias + n4));
#pragma unroll
          for (int it = 0; it < ITS; ++it) {
            const long long off = off0 + (long long)it * 8 * l_ld;
            f32x4 z = *reinterpret_cast<f32x4*>(lp + it * 32);
            f32x4 v;
            if (MODE == 0) {
              f32x4 rb = {0.f, 0.f, 0.f, 0.f};
              if (p_rowbias) rb = *((gptr<const f32x4>)(p_rowbias + ((row0 + mbase + 8 * it) / rb_div) * (long long)l_N + n4));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float t = z[q] + bias4[q] + rb[q];
                float u = b2 * t;
                float sp = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(u)) * ib2;
                v[q] = (u > 20.f * LOG2E ? t : sp) * sc;
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
            } else {
              f32x4 ex = {0.f, 0.f, 0.f, 0.f}, x2;
              if (MODE == 1 && p_side_add) ex = *((gptr<const f32x4>)(p_side_add + off));
              if (MODE == 2 && p_side_in2) ex = *((gptr<const f32x4>)(p_side_in2 + off));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float e = __builtin_amdgcn_exp2f(nb2 * hsv[it][q]);   // exp(-beta h)
                float sp = (1.f - e) * sc;                            // softplus'(z) [* skip scale]
                if (MODE == 1) v[q] = z[q] * sp + ex[q];
                else { v[q] = z[q] * sp; x2[q] = beta * z[q] * ex[q] * e; }   // ex = s of the sdf chain
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
              if (MODE == 2 && p_side_out2) *((gptr<f32x4>)(p_side_out2 + off)) = x2;
              colsum += v;
            }
            *reinterpret_cast<f32x4*>(lp + it * 32) = v;
          }
        } else {
          // general path: ragged column block, partial tile, unaligned rows, or the output layer
          const bool vec_ok = (n4 + 3 < nlim);
          const bool vec_side = vec_ok && (l_ld & 3) == 0;
          f32x4 cm, bias4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 4; ++q) cm[q] = (n4 + q < nlim) ? 1.f : 0.f;
          if (MODE == 0 && p_bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n4 + q < l_N) bias4[q] = p_bias[n4 + q];
          }
#pragma unroll 1
          for (int it = 0; it < ITS; ++it, lp += 32) {
            const int m = mbase + 8 * it;
            const bool mrow = m < rows;
            const float rm = mrow ? 1.f : 0.f;
            const long long grow = row0 + m;
            const long long off = off0 + (long long)it * 8 * l_ld;
            f32x4 z = *reinterpret_cast<f32x4*>(lp);
            f32x4 v;
            if (last) {
              // output layer (forward: + bias; backward: plain) -> Y
              if (mrow) {
                float* y = a.Y + grow * a.ldy + n4;
                if (vec_ok && (a.ldy & 3) == 0 && !a.accum_y) {
                  f32x4 t = z;
                  if (MODE == 0) t += bias4;
                  *reinterpret_cast<f32x4*>(y) = t;
                } else {
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    if (n4 + q < l_N) {
                      float t = z[q] + (MODE == 0 ? bias4[q] : 0.f);
                      y[q] = a.accum_y ? y[q] + t : t;
                    }
                  }
                }
              }
              continue;
            }
            if (MODE == 0) {
              // hidden forward layer: softplus_beta(z + b) [* skip scale]
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float t = z[q] + bias4[q];
                if (p_rowbias && mrow && n4 + q < l_N) t += p_rowbias[(grow / rb_div) * (long long)l_N + n4 + q];
                float u = b2 * t;
                float sp = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(u)) * ib2;
                v[q] = (u > 20.f * LOG2E ? t : sp) * (sc * cm[q] * rm);
              }
              if (mrow && p_side_out) {
                if (vec_side) *((gptr<f32x4>)(p_side_out + off)) = v;
                else {
#pragma unroll
                  for (int q = 0; q < 4; ++q) if (n4 + q < l_N) p_side_out[off + q] = v[q];
                }
              }
            } else {
              // MODE 1: this GEMM produced dL/dh of the layer below; MODE 2: the tangent s-bar of this layer.
              // The stored activation h gives softplus'(z) = 1 - exp(-beta h).
              f32x4 hs = {0.f, 0.f, 0.f, 0.f}, ex = {0.f, 0.f, 0.f, 0.f}, x2 = {0.f, 0.f, 0.f, 0.f};
              if (mrow) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (n4 + q < nlim) {
                  hs[q] = p_side_in[off + q];
                  if (MODE == 1 && p_side_add) ex[q] = p_side_add[off + q];
                  if (MODE == 2 && p_side_in2) ex[q] = p_side_in2[off + q];
                }
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float e = __builtin_amdgcn_exp2f(nb2 * hs[q]);            // exp(-beta h)
                float sp = (1.f - e) * sc;
                float mk = cm[q] * rm;
                if (MODE == 1) v[q] = (z[q] * sp + ex[q]) * mk;
                else { v[q] = z[q] * sp * mk; x2[q] = beta * z[q] * ex[q] * e * mk; }   // ex = s of the sdf chain
              }
              if (MODE == 1 && is_skip && mrow && a.Xskip) {
                // gradient of the concatenated chain input: stash (scaled) for the final dL/dX
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int n = n4 + q;
                  if (n >= a.skip_split && n < l_N) a.Xskip[grow * a.ld_xskip + (n - a.skip_split)] = z[q] * sc;
                }
              }
              if (mrow) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (n4 + q < nlim) {
                  if (p_side_out) p_side_out[off + q] = v[q];
                  if (MODE == 2 && p_side_out2) p_side_out2[off + q] = x2[q];
                }
              }
              colsum += v;
            }
            *reinterpret_cast<f32x4*>(lp) = v;
          }
        }
        if (MODE != 0 && !last && p_bgrad) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float c = colsum[q];
            c += __shfl_xor(c, 8);
            c += __shfl_xor(c, 16);
            c += __shfl_xor(c, 32);
            if (lane < 8 && n4 + q < nlim) atomicAdd(p_bgrad + n4 + q, c);   // LDS (row-split units share columns)
          }
        }
      };

      {
        const int rem = (RB == 1) ? 0 : (NB % NWAVES);
        const int full = NB - rem;
        for (int nb = wave; nb < full; nb += NWAVES) process(std::integral_constant<int, RB>{}, nb, 0);
        for (int u = wave; u < rem * RB; u += NWAVES) process(std::integral_constant<int, 1>{}, full + u / RB, u % RB);
        stamp(li, 3);
      }

      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output ----
      if (MODE != 1 && li == a.skip_layer) {
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
      stamp(li, 4);
      float* t = cur; cur = nxt; nxt = t;
    }
  }
  if (MODE != 0 && a.bg_total > 0) {
    __syncthreads();
    float* part = a.bg_partial + (long long)blockIdx.x * a.bg_total;
    for (int i = tid; i < a.bg_total; i += NTHREADS) part[i] = bsum[i];
  }
}

// bias gradients: out_l[n] = sum over workgroups of partial[g][off_l + n]
struct BgOut {
  float* ptr[MAX_CHAIN_LAYERS + 1];
  int off[MAX_CHAIN_LAYERS + 2];
  int n;
  int accum;       // destinations += sums
};
__global__ void __launch_bounds__(256) k_bgrad_reduce(const float* __restrict__ partial, int S, int total, BgOut o) {
  __shared__ float red[256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (i < total) {
#pragma unroll 8
    for (int sidx = ty; sidx < S; sidx += 8) acc += partial[(long long)sidx * total + i];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (ty == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q * 32 + tx];
    int l = 0;
    while (l + 1 < o.n && i >= o.off[l + 1]) ++l;
    float* dst = o.ptr[l] + (i - o.off[l]);
    *dst = o.accum ? *dst + t : t;
  }
}

int launch_bgrad_reduce(const float* partial, int S, int total, float* const* ptr, const int* off, int n, int accum, hipStream_t stream) {
  BgOut o{};
  o.n = n;
  o.accum = accum;
  for (int i = 0; i < n; ++i) { o.ptr[i] = ptr[i]; o.off[i] = off[i]; }
  o.off[n] = total;
  hipLaunchKernelGGL(k_bgrad_reduce, dim3((total + 31) / 32), dim3(256), 0, stream, partial, S, total, o);
  return ndjir_check_launch();
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

int launch_chain(const ChainArgs& a, int mode, hipStream_t stream) {
  const bool bwd = (mode == 1);
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
  const int TM = a.tile_rows == 32 ? 32 : 64;
  const int GP = TM * 4 + 4;
  size_t szA = (size_t)(wa / 4) * GP * 4;
  size_t szB = (size_t)(wmax / 4) * GP * 4;
  const size_t partials = (size_t)8 * 32 * 32 * 4;                     // narrow-layer partial sums
  if (szA < partials) szA = partials;
  if (szB < partials) szB = partials;
  ChainArgs b = a;
  b.lds_split = (int)(szA / 4);
  b.n_tiles = (a.P + TM - 1) / TM;   // TM defined above
  size_t lds_bytes = szA + szB;
  BgOut bg{};
  bg.accum = a.bg_accum;
  int bg_total = 0;
  if (mode != 0) {
    for (int i = 0; i < a.L; ++i) if (a.layers[i].bgrad && !(a.has_output && i == a.L - 1)) {
      b.layers[i].bg_off = bg_total;
      bg.ptr[bg.n] = a.layers[i].bgrad;
      bg.off[bg.n] = bg_total;
      ++bg.n;
      bg_total += a.layers[i].N;
    } else b.layers[i].bgrad = nullptr;
    bg.off[bg.n] = bg_total;
  }
  if (mode != 0 && a.in_bgrad) {
    b.in_bg_off = bg_total;
    bg.ptr[bg.n] = a.in_bgrad;
    bg.off[bg.n] = bg_total;
    ++bg.n;
    bg_total += a.K0;
    bg.off[bg.n] = bg_total;
  }
  b.bg_total = bg_total;
  b.bg_lds = (int)(lds_bytes / 4);
  lds_bytes += (size_t)bg_total * 4;
  if (bg_total > 0 && !a.bg_partial) return NDJIR_ERR_ARG;
  if (lds_bytes > 160 * 1024) return NDJIR_ERR_UNSUPPORTED;
  long long blocks = b.n_tiles;
  if (blocks > 256LL * 8) blocks = 256LL * 8;
  if (bg_total > 0 && blocks > CHAIN_MAX_GRID_BG) blocks = CHAIN_MAX_GRID_BG;
  static bool attr_set = false;
  if (!attr_set) {
#define NDJIR_SET(M, T) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain<M, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    NDJIR_SET(0, 64); NDJIR_SET(1, 64); NDJIR_SET(2, 64); NDJIR_SET(0, 32); NDJIR_SET(1, 32); NDJIR_SET(2, 32);
#undef NDJIR_SET
    attr_set = true;
  }
  if (a.dry) { snprintf(a.dry->name, 64, "ndjir::k_mlp_chain<%d, %d>", mode, TM); a.dry->blocks = (int)blocks; a.dry->bg_total = -1; return NDJIR_OK; }
#define NDJIR_GO(M, T) hipLaunchKernelGGL((k_mlp_chain<M, T>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, b)
  if (TM == 64) { if (mode == 0) NDJIR_GO(0, 64); else if (mode == 1) NDJIR_GO(1, 64); else NDJIR_GO(2, 64); }
  else { if (mode == 0) NDJIR_GO(0, 32); else if (mode == 1) NDJIR_GO(1, 32); else NDJIR_GO(2, 32); }
#undef NDJIR_GO
  if (bg_total > 0)
    hipLaunchKernelGGL(k_bgrad_reduce, dim3((bg_total + 31) / 32), dim3(256), 0, stream, a.bg_partial, (int)blocks, bg_total, bg);
  return ndjir_check_launch();
}

}  // namespace ndjir
