// mlp3.hip -- the fused MLP chain on the f16 matrix cores: two-way split operands, three partial products.
//
// gfx950 runs fp32-input MFMA at the fp32 VECTOR rate; 16-bit MFMA is 16x faster.  mlp6.hip splits every
// fp32 operand exactly into three bf16 planes (8 + 8 + 8 bits) and pays six partial products.  Here each
// operand is scaled by a power of two (exact) into the fp16 range and split into TWO fp16 planes
//     x s = hi + lo 2^-11,   hi = fp16(x s),   lo = fp16((x s - hi) 2^11)          (11 + 11 significant bits)
// and a product is formed from THREE partial products in two fp32 accumulators
//     acc0 += hi hi'            acc1 += hi lo' + lo hi'            z = (acc0 + acc1 2^-11) / (s s')
// (lo lo' ~ 2^-22 of the product is dropped; each partial product is an exact fp16 x fp16 product accumulated
// in fp32 by v_mfma_f32_32x32x16_f16).  Keeping lo pre-scaled by 2^11 puts it in hi's exponent range, so every
// element within 2^-28 of the largest one of its scaling group keeps all 22 bits; smaller ones degrade
// gracefully (absolute error <= 2^-39 of the group maximum).  Scaling groups: a 32-column block of a packed
// weight matrix (scale fixed at pack time, stored behind the planes), and ONE ROW (point) of the activations: the row
// maxima are reduced in LDS between the activation math and the split.  Power-of-two scaling is exact, so a point's
// result does not depend on which other points share its tile (batch invariance, as with the other engines).
// Accuracy (tests/test_gpu_mlp.py, vs fp64): below a plain fp32 FMA chain's, at 3/16 of the fp32 MFMA time
// and half of mlp6's -- operand rounding 2^-23 sits under the fp32 accumulation error that every fp32 GEMM has.
//
// Structure per tile of TM points, 8 waves (as mlp6.hip, with the epilogue cut in two at the tile maximum):
//   for every layer:  k-loops of all column blocks (accumulator pairs stay in registers)
//                     phase A per 32x32 block: acc0 + acc1 2^-11 -> per-wave fp32 staging tile in LDS -> row-major
//                       pass (bias + softplus / softplus' product, coalesced float4 side stores and loads,
//                       bias-gradient column sums); results written back to the staging tile; running maximum
//                     row maxima -> LDS; barrier (every wave has also finished reading the planes)
//                     phase B: per-row scale from the maximum, staging tile -> 2-way split, planes updated in place
//                     barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "mlp.h"
#include "mlp3_util.h"

namespace ndjir {
namespace x3 {

using namespace x3u;

constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
constexpr int MAXNB = 16;        // widest layer: 512 columns (2 column blocks per wave)
constexpr int GPS = 32 * 4 + 4;  // staging tile: dwords per group of 4 columns (32 rows + pad)
constexpr int STG = 8 * GPS;     // staging dwords per wave (32 x 32 tile)
constexpr int IN_CACHE = 10;     // float4 groups of the chain input a thread keeps between the max pass and the split
#ifndef NDJIR_BWD_KLOOP_BARRIER
#define NDJIR_BWD_KLOOP_BARRIER 0
#endif
constexpr bool BWD_KLOOP_BARRIER = NDJIR_BWD_KLOOP_BARRIER;   // (A/B switch; measured +-0 on the backward chain, off)

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- weight packing -----------------------------------------------------------------------------
// dst (16-byte units): [Np/32][Kp/16][plane 0..1][lane 0..63], lane (c = lane & 31, h = lane >> 5) holds
// W[16 ks + 8 h + j][32 nb + c] * s_nb, j = 0..7, of plane p (0 = hi, 1 = lo * 2^11); behind the planes, at float
// offset Kp * Np: 1 / s_nb for every column block.  One workgroup per column block: slab maximum, then the split.
// transpose packs W^T.
constexpr int PACK_T = 1024;
// one column block nb of one matrix: W (K, N) with row stride ldw (a column slice of a wider matrix packs in place)
__device__ __forceinline__ void pack3_block(const float* __restrict__ W, int ldw, _Float16* __restrict__ dst, float* __restrict__ inv_out,
                                            int K, int N, int transpose, int Kp, int Np, int nb, unsigned* red) {
  const int KS = Kp >> 4;
  const int total = Kp * 32;
  auto value = [&](int t) -> float {
    int k, c;
    if (!transpose) { c = t & 31; k = t >> 5; } else { k = t % Kp; c = t / Kp; }     // contiguous axis fastest
    const int n = nb * 32 + c;
    float v = 0.f;
    if (!transpose) { if (k < K && n < N) v = W[(long long)k * ldw + n]; }
    else { if (k < N && n < K) v = W[(long long)n * ldw + k]; }
    return v;
  };
  unsigned m = 0;
  for (int t = threadIdx.x; t < total; t += PACK_T) { const unsigned b = finite_abs_bits(value(t)); m = b > m ? b : m; }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = PACK_T / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { const unsigned o = red[threadIdx.x + s]; if (o > red[threadIdx.x]) red[threadIdx.x] = o; }
    __syncthreads();
  }
  float sc, inv;
  scale_from_max(red[0], sc, inv);
  if (threadIdx.x == 0) {
    inv_out[nb] = inv;
    // (the scale array is padded to a multiple of 4 floats: a packed buffer is a deterministic function of its matrix)
    if (nb == (Np >> 5) - 1) for (int i = nb + 1; i < (((Np >> 5) + 3) & ~3); ++i) inv_out[i] = 0.f;
  }
  for (int t = threadIdx.x; t < total; t += PACK_T) {
    int k, c;
    if (!transpose) { c = t & 31; k = t >> 5; } else { k = t % Kp; c = t / Kp; }
    const float xs = value(t) * sc;
    const _Float16 hi = (_Float16)xs;
    const _Float16 lo = (_Float16)((xs - (float)hi) * LO_SCALE);
    const int ks = k >> 4, lane = c + 32 * ((k & 15) >> 3), j = k & 7;
    const long long base = (((long long)nb * KS + ks) * 2) * 64 * 8 + lane * 8 + j;
    dst[base] = hi;
    dst[base + 64 * 8] = lo;
  }
}

__global__ void __launch_bounds__(PACK_T) k_pack3(const float* __restrict__ W, int ldw, _Float16* __restrict__ dst,
                                               float* __restrict__ inv_out, int K, int N, int transpose, int Kp, int Np) {
  __shared__ unsigned red[PACK_T];
  pack3_block(W, ldw, dst, inv_out, K, N, transpose, Kp, Np, blockIdx.x, red);
}

// Many matrices in ONE launch: a device table of (matrix, orientation) entries, one workgroup per column block of every
// entry.  What the optimizer step launches after its update instead of one pack per weight and orientation (~50 launches
// per training iteration): the packed copies are persistent buffers, rewritten in place.
__global__ void __launch_bounds__(PACK_T) k_pack3_table(const PackEntry* __restrict__ tab, int n) {
  __shared__ unsigned red[PACK_T];
  int lo = 0, hi = n - 1;                       // last entry whose first_block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackEntry e = tab[lo];
  pack3_block(e.W, e.ldw, reinterpret_cast<_Float16*>(e.dst), e.dst + (long long)e.Kp * e.Np, e.K, e.N, e.transpose, e.Kp, e.Np,
              (int)blockIdx.x - e.first_block, red);
}

// ---- the chain kernel ---------------------------------------------------------------------------
template <int MODE, int TM>
__global__ void __launch_bounds__(NTHREADS, 2) k_chain3(ChainArgs a) {
  constexpr bool BWD = (MODE == 1);
  constexpr bool BATCH_EPILOGUE = (MODE == 0);
  constexpr int RB = TM / 32;
  constexpr int TMP = TM + 4;          // rows per k-group incl. pad: (TMP * 16) % 256 == 64 -> conflict-free plane writes
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned s_rmax[2][64];   // per row: largest finite |output| of the layer (bit pattern; ping-pong by layer)
  __shared__ unsigned s_xmax[2][64];   // per row: largest finite |x| of the chain input tile (ping-pong by tile)
  __shared__ float s_ainv[64];         // per row: 1 / scale of the planes the next k-loop reads
  const int PLANE = a.lds_split;       // 16-byte units per plane ( = k-groups * TMP )
  f16x8* act = reinterpret_cast<f16x8*>(lds);
  char* actb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // an opaque copy of the lane id: addresses derived from it are formed where they are used instead of being hoisted out of
  // the tile / layer loops into registers that stay occupied (or spilled) through the k-loops
  auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  float* stage_all = lds + (size_t)2 * PLANE * 4;
  float* stage = stage_all + wave * 2 * STG;      // two 32 x 32 tiles per wave: one per output block of a layer
  float* bsum = lds + a.bg_lds;
  const float beta = a.beta;
  auto stamp = [&](int li, int phase) {
    if (a.timeline && blockIdx.x == 0 && lane == 0) a.timeline[(li * 5 + phase) * NWAVES + wave] = (long long)__builtin_amdgcn_s_memtime();
  };
  // write 4 consecutive features k..k+3 (k % 4 == 0) of row m, scaled by s, into the two planes
  auto put4 = [&](int k, int m, f32x4 v, float s) {
    f16x4 ph, pl;
    split4(v, s, ph, pl);
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<f16x4*>(p) = ph;
    *reinterpret_cast<f16x4*>(p + (size_t)PLANE * 16) = pl;
  };
  auto put1 = [&](int k, int m, float v, float s) {
    _Float16 ph, pl;
    split1(v, s, ph, pl);
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<_Float16*>(p) = ph;
    *reinterpret_cast<_Float16*>(p + (size_t)PLANE * 16) = pl;
  };

  stamp(MAX_CHAIN_LAYERS - 1, 0);                  // (diagnostics: kernel start / first input stage done / first tile done)
  auto stamp_rt = [&](int phase) {                 // constant-rate (100 MHz) clock beside the shader clock: effective frequency
    if (a.timeline && blockIdx.x == 0 && lane == 0) a.timeline[((MAX_CHAIN_LAYERS - 2) * 5 + phase) * NWAVES + wave] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  stamp_rt(0);
  // diagnostics: slot [9][4][7] of the timeline buffer = number of per-workgroup records behind it (start / end on the
  // 100 MHz clock, hardware id): lets tools/chain_timeline.py see how the grid packs onto the CUs
  const bool rec = a.timeline && tid == 0 && (long long)blockIdx.x < a.timeline[399];
  if (rec) {
    a.timeline[400 + 3 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
    a.timeline[400 + 3 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  }
  if (MODE != 0) for (int i = tid; i < a.bg_total; i += NTHREADS) bsum[i] = 0.f;
  if (tid < 128) { (&s_rmax[0][0])[tid] = 0u; (&s_xmax[0][0])[tid] = 0u; }
  __syncthreads();
  int xpar = 0;                        // ping-pong slot of the input row maxima

  for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = tile * TM;
    const int rows = (int)((a.P - row0) < TM ? (a.P - row0) : TM);

    // ---- chain input tile -> planes (zero padded to a multiple of 16 features) ----
    {
      const int K0p = a.K0p, K0 = a.K0;
      const float* X = a.X + row0 * a.ldx;
      int groups = K0p >> 2;
      // (opaque to the optimiser: the per-thread addresses below are tile-invariant, and hoisted out of the tile loop
      // they would sit in -- spilled -- registers for the whole kernel)
      asm volatile("" : "+s"(groups));
      const int total = groups * TM;
      const bool vec = (a.ldx & 3) == 0 && ((uintptr_t)a.X & 15) == 0;
      auto load = [&](int t) -> f32x4 {
        const int g = t % groups, m = t / groups;
        const int k = g * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < rows) {
          if (vec && k + 3 < K0) v = *reinterpret_cast<const f32x4*>(X + (long long)m * a.ldx + k);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (k + q < K0) v[q] = X[(long long)m * a.ldx + k + q];
          }
        }
        return v;
      };
      f32x4 cache[IN_CACHE];
      auto rowmax = [&](int t, f32x4 v) {
        unsigned mb = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const unsigned b = finite_abs_bits(v[q]); mb = b > mb ? b : mb; }
        if (mb) atomicMax(&s_xmax[xpar][t / groups], mb);
      };
#pragma unroll
      for (int i = 0; i < IN_CACHE; ++i) {
        const int t = tid + i * NTHREADS;
        cache[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < total) { cache[i] = load(t); rowmax(t, cache[i]); }
      }
      for (int t = tid + IN_CACHE * NTHREADS; t < total; t += NTHREADS) rowmax(t, load(t));
      __syncthreads();
      if (tid < TM) {
        float s_row, inv_row;
        scale_from_max(s_xmax[xpar][tid], s_row, inv_row);
        s_ainv[tid] = inv_row;
        s_xmax[xpar ^ 1][tid] = 0u;                    // the other slot: next tile's input stage, many barriers away
      }
      if (a.x_amax && tid < 64) {
        const float wm = wave_max(__uint_as_float(tid < TM ? s_xmax[xpar][tid] : 0u));
        if (tid == 0) atomicMax(a.x_amax, __float_as_uint(wm));
      }
      auto emit = [&](int t, f32x4 v) {
        const int g = t % groups, m = t / groups;
        const int k = g * 4;
        float s_in, inv_in;
        scale_from_max(s_xmax[xpar][m], s_in, inv_in);
        put4(k, m, v, s_in);
        if (MODE != 0 && a.in_bgrad && m < rows) {      // bias gradient of the output layer: column sums of the input
#pragma unroll
          for (int q = 0; q < 4; ++q) if (k + q < K0) atomicAdd(bsum + a.in_bg_off + k + q, v[q]);
        }
      };
#pragma unroll
      for (int i = 0; i < IN_CACHE; ++i) {
        const int t = tid + i * NTHREADS;
        if (t < total) emit(t, cache[i]);
      }
      for (int t = tid + IN_CACHE * NTHREADS; t < total; t += NTHREADS) emit(t, load(t));
    }
    __syncthreads();
    if (tile == blockIdx.x) stamp(MAX_CHAIN_LAYERS - 1, 1);

    int cur = 0;                         // ping-pong slot of the layer's tile maximum
    for (int li = 0; li < a.L; ++li) {
      const ChainLayer& ly = a.layers[li];
      const int KS = (ly.Kp + 15) >> 4;          // k-steps of 16 (planes are zero beyond Kp)
      const int NB = ly.Np >> 5;
      const bool last = a.has_output && (li == a.L - 1);
      stamp(li, 0);
      const gptr<const f16x8> p_wp = (gptr<const f16x8>)pin(ly.Wp);
      const gptr<const float> p_winv = pin(ly.Wp + (long long)KS * 16 * ly.Np);   // [NB] behind the planes
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
      const int nlim = pin((BWD && is_skip) ? a.skip_split : l_N);

      // one k-loop: RBU row blocks of column block nb, accumulator pairs acc0/acc1[SLOT .. SLOT + RBU)
      // Two accumulator-pair slots: a wave works on at most two 32 x 32 output blocks at a time.
      f32x16 acc0[2], acc1[2];   // static indices only (an accumulator array indexed under run-time branches goes to scratch)
      auto kloop = [&](auto rbu_tag, auto slot_tag, const int nb, const int rb0, const int ks0, const int ks1) {
        constexpr int RBU = decltype(rbu_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
        for (int q = 0; q < RBU; ++q) { acc0[SLOT + q] = f32x16{0}; acc1[SLOT + q] = f32x16{0}; }
        const int lane_k = fresh_lane();
        const gptr<const f16x8> Bp = p_wp + ((long long)nb * KS) * 2 * 64 + lane_k;
        const f16x8* A0 = act + (lane_k >> 5) * TMP + rb0 * 32 + (lane_k & 31);
        // Software pipeline (static register slots, unrolled by 3): weight fragments 3 steps ahead, activation
        // fragments one step ahead (double buffered).
        f16x8 b[3][2];                 // [slot][plane]
        f16x8 af[2][2][RBU];           // [buffer][plane][row block]
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int p = 0; p < 2; ++p) b[s][p] = Bp[(long long)((ks0 + s < ks1 ? ks0 + s : ks0) * 2 + p) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const f16x8* An = A0 + 2 * ks0 * TMP;
#pragma unroll
          for (int q = 0; q < RBU; ++q) { af[0][0][q] = An[q * 32]; af[0][1][q] = An[PLANE + q * 32]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto stag, auto btag, auto gtag, const int ks) {
          constexpr int S = decltype(stag)::value;       // weight slot
          constexpr int C = decltype(btag)::value;       // activation buffer
          constexpr bool GUARD = decltype(gtag)::value;
          const bool nxt = !GUARD || ks + 1 < ks1;
          const f16x8* An = A0 + 2 * (ks + 1) * TMP;
          if (nxt) {
#pragma unroll
            for (int q = 0; q < RBU; ++q) { af[C ^ 1][0][q] = An[q * 32]; af[C ^ 1][1][q] = An[PLANE + q * 32]; }
          }
          // three partial products: lo*hi', hi*lo' -> acc1; hi*hi' -> acc0 (dependent MFMAs kept apart)
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc1[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[C][1][q], b[S][0], acc1[SLOT + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc0[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[C][0][q], b[S][0], acc0[SLOT + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc1[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[C][0][q], b[S][1], acc1[SLOT + q], 0, 0, 0);
          if (!GUARD || ks + 3 < ks1) {
#pragma unroll
            for (int p = 0; p < 2; ++p) b[S][p] = Bp[(long long)((ks + 3) * 2 + p) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        {
          using T = std::true_type;
          using F = std::false_type;
          using S0 = std::integral_constant<int, 0>;
          using S1 = std::integral_constant<int, 1>;
          using S2 = std::integral_constant<int, 2>;
          int ks = ks0;
          // the weight slots rotate with period 3, the activation buffers with period 2: unroll by 6
          for (; ks + 9 <= ks1; ks += 6) {
            kstep(S0{}, S0{}, F{}, ks); kstep(S1{}, S1{}, F{}, ks + 1); kstep(S2{}, S0{}, F{}, ks + 2);
            kstep(S0{}, S1{}, F{}, ks + 3); kstep(S1{}, S0{}, F{}, ks + 4); kstep(S2{}, S1{}, F{}, ks + 5);
          }
          for (; ks < ks1; ks += 6) {
            kstep(S0{}, S0{}, T{}, ks);
            if (ks + 1 < ks1) kstep(S1{}, S1{}, T{}, ks + 1);
            if (ks + 2 < ks1) kstep(S2{}, S0{}, T{}, ks + 2);
            if (ks + 3 < ks1) kstep(S0{}, S1{}, T{}, ks + 3);
            if (ks + 4 < ks1) kstep(S1{}, S0{}, T{}, ks + 4);
            if (ks + 5 < ks1) kstep(S2{}, S1{}, T{}, ks + 5);
          }
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      using IRB = std::integral_constant<int, RB>;

      if (NB == 1) {
        // ---- narrow output (N <= 32): K split over 4 wave groups, partial sums through the staging area ----
        constexpr int KSPLIT = 4;
        const int rb = wave % RB, kq = wave / RB;
        const int k0 = kq < KSPLIT ? (KS * kq) / KSPLIT : 0, k1 = kq < KSPLIT ? (KS * (kq + 1)) / KSPLIT : 0;
        kloop(I1{}, I0{}, 0, rb, k0, k1);
        float* part = stage_all;                 // [kq][m][n]: KSPLIT x TM x 32 floats
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kq < KSPLIT) part[(kq * TM + rb * 32 + acc_row(i, h)) * 32 + r] = acc_sum(acc0[0][i], acc1[0][i]);
        __syncthreads();
        const float winv = p_winv[0];
        for (int t = tid; t < TM * 32; t += NTHREADS) {
          const int n = t & 31, m = t >> 5;
          float z = 0.f;
#pragma unroll
          for (int q = 0; q < KSPLIT; ++q) z += part[q * TM * 32 + t];
          z = out_z(z, s_ainv[m], winv);
          if (n < l_N && m < rows) {
            if (MODE == 0) z = out_add(z, p_bias ? p_bias[n] : 0.f);
            if (last) {      // (a narrow layer is always an output layer: chain_impl refuses it elsewhere)
              float* y = a.Y + (row0 + m) * a.ldy + n;
              *y = a.accum_y ? out_add(z, *y) : z;
            }
          }
        }
        __syncthreads();
        continue;
      }

      // ---- general layer: this wave's 32 x 32 output blocks ("jobs"), two per round ----
      // TM = 64 and a multiple of 8 column blocks: a column block goes to one wave with both row blocks (the weight
      // fragments are fetched once per tile); otherwise the NB * RB blocks are dealt round-robin.  Hidden layers fit one
      // round (the launcher picks 32-point tiles for hidden layers wider than 256: their results are parked in the
      // wave's two staging tiles); an output layer, which parks nothing, may take several.
      const bool pair_mode = (RB == 2) && (NB % NWAVES == 0);
      const int nblk = NB * RB;
      const int nrounds = pair_mode ? NB / NWAVES : (nblk + 2 * NWAVES - 1) / (2 * NWAVES);
      constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
      const float b2 = beta * LOG2E, ib2sc = LN2 / beta * sc;
      const float hsc = (MODE != 0 && is_skip) ? 1.f / sc : 1.f;
      const float nb2 = -b2 * hsc;
      const int g = lane & 7;
      unsigned jobs = 0;    // job j in bits 8j..8j+7: column block | row block << 5
      int njobs = 0;
      // running maximum of row `row` (8 lanes hold 4 columns each of it): max over the lane's 4 values, xor-butterfly over
      // the 8 lanes, one LDS atomic.  v_max ignores NaN; an Inf sends the group through the bit-pattern filter.
      auto row_max = [&](int row, f32x4 v) {
        float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        if (!(m4 < 3.0e38f)) {
          unsigned mb = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) { const unsigned b = finite_abs_bits(v[q]); mb = b > mb ? b : mb; }
          m4 = __uint_as_float(mb);
        }
        m4 = fmaxf(m4, __shfl_xor(m4, 1));
        m4 = fmaxf(m4, __shfl_xor(m4, 2));
        m4 = fmaxf(m4, __shfl_xor(m4, 4));
        if (g == 0) atomicMax(&s_rmax[cur][row], __float_as_uint(m4));
      };
      // ---- batched epilogue of a hidden layer (every job of the wave is a full 32 x 32 block of full rows) ----
      // Straight-line code over the wave's NJ blocks x 4 row steps: all side loads are requested at once, ahead of the
      // staging round trip; the results stay in registers across the row-maximum barrier (phase B splits them from
      // there: no parking in the staging tile); the 8-lane row maxima and the bias-gradient column sums cross lanes by
      // DPP instead of ds_bpermute.  (The per-job path below keeps the ragged / output-layer / multi-round cases.)
      f32x4 pv[2][4];          // phase A results of the batched path: [job][row step]
      auto phase_a_batched = [&](auto nj_tag, auto rb_tag) {
        constexpr int NJ = decltype(nj_tag)::value;
        constexpr bool HAS_RB = decltype(rb_tag)::value;
        // (lane-derived addresses are formed here, from an opaque copy of the lane id: hoisted out of the layer / tile loops
        // they would sit in -- spilled -- registers through the k-loops)
        const int lane_o = fresh_lane();
        const int g = lane_o & 7, l8 = lane_o >> 3, r = lane_o & 31, h = lane_o >> 5;
        int nbj[NJ], rbj[NJ];
        unsigned lo[NJ];          // element offset of (row lane / 8 of row block, column group g of column block) in the tile
        float winv[NJ];
        // uniform base (SGPR pair) + 32-bit lane offset: one address register per access instead of a 64-bit pair
        const long long tile_off = row0 * l_ld;
        const unsigned step = 8u * (unsigned)l_ld;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          nbj[j] = (jobs >> (8 * j)) & 31;
          rbj[j] = (jobs >> (8 * j + 5)) & 7;
          lo[j] = (unsigned)(rbj[j] * 32 + l8) * (unsigned)l_ld + (unsigned)(nbj[j] * 32 + g * 4);
          winv[j] = p_winv[nbj[j]];
        }
        // side loads first (independent of the staging tile)
        f32x4 hs[NJ][4], ex[NJ][4], bb[NJ];
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            bb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p_bias) bb[j] = *((gptr<const f32x4>)(p_bias + (unsigned)(nbj[j] * 32 + g * 4))) * b2;
          }
          if (HAS_RB) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
              // row group of (row0 + 32 rb + lane / 8 + 8 it): the tile-level part is wave-uniform
              const unsigned base = (unsigned)(row0 + rbj[j] * 32);
              const unsigned d = (unsigned)rb_div;
              const unsigned q0 = pin((int)(base / d)), r0 = pin((int)(base % d));
#pragma unroll
              for (int it = 0; it < 4; ++it) {
                const unsigned t = r0 + (unsigned)l8 + 8u * it;
                const unsigned grp = q0 + (d >= 32u ? (t >= d ? 1u : 0u) : t / d);
                hs[j][it] = *((gptr<const f32x4>)(p_rowbias + (long long)grp * l_N + nbj[j] * 32 + g * 4));
              }
            }
          }
        }
#ifndef HS_LATE
        if (MODE != 0) {
          const gptr<const float> b_in = p_side_in + tile_off;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) hs[j][it] = *((gptr<const f32x4>)(b_in + (lo[j] + it * step)));
        }
#endif
        // pass 1: acc0 + acc1 2^-11 -> staging tiles (conflict-free: column groups GPS apart, rows 4 dwords apart)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float* dst = stage + j * STG + (r >> 2) * GPS + (r & 3);
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[acc_row(i, h) * 4] = fmaf(acc1[j][i], LO_INV, acc0[j][i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // second set of side loads (backward: extra adjoints; tangent: s) once the accumulators are out of the registers
        if (MODE != 0) {
#ifdef HS_LATE
          const gptr<const float> b_in = p_side_in + tile_off;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) hs[j][it] = *((gptr<const f32x4>)(b_in + (lo[j] + it * step)));
#endif
          const gptr<const float> p_ex = MODE == 1 ? p_side_add : p_side_in2;
          if (p_ex) {
            const gptr<const float> b_ex = p_ex + tile_off;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int it = 0; it < 4; ++it) ex[j][it] = *((gptr<const f32x4>)(b_ex + (lo[j] + it * step)));
          } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
              for (int it = 0; it < 4; ++it) ex[j][it] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        __builtin_amdgcn_wave_barrier();
        // pass 2: lane = (column group g of 4 columns, row lane / 8 of every 8-row step)
        float sa[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float* lp = stage + j * STG + g * GPS + l8 * 4;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            pv[j][it] = *reinterpret_cast<const f32x4*>(lp + it * 32);
            sa[j][it] = s_ainv[rbj[j] * 32 + l8 + 8 * it];
          }
        }
        const gptr<float> b_out2 = (MODE == 2 && p_side_out2) ? p_side_out2 + tile_off : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float wk = winv[j];
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            f32x4 z, v;
            if (MODE == 0) {
              const float kk = fwd_kk(sa[j][it], winv[j], b2);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float u = fwd_u(pv[j][it][q], kk, bb[j][q]);          // b2 * (pre-activation)
                if (HAS_RB) u = fwd_u_rowbias(u, hs[j][it][q], b2);
                v[q] = softplus_u(u, ib2sc);
              }
            } else {
              z = pv[j][it] * (sa[j][it] * wk);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float e = __builtin_amdgcn_exp2f(nb2 * hs[j][it][q]);
                const float sp = (1.f - e) * sc;
                if (MODE == 1) v[q] = fmaf(z[q], sp, ex[j][it][q]);
                else { v[q] = z[q] * sp; ex[j][it][q] = beta * z[q] * ex[j][it][q] * e; }     // (the extra adjoint, in place)
              }
              if (MODE == 2 && b_out2) *((gptr<f32x4>)(b_out2 + (lo[j] + it * step))) = ex[j][it];
            }
            pv[j][it] = v;
          }
        }
        if (p_side_out) {
          const gptr<float> b_out = p_side_out + tile_off;
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) *((gptr<f32x4>)(b_out + (lo[j] + it * step))) = pv[j][it];
        }
        // row maxima: the lane's 4 values, the row's 8 lanes by DPP, one LDS atomic per row and block.  v_max ignores NaN;
        // a group holding an Inf (or only NaN) goes through the bit-pattern filter.
        float m4[NJ][4];
        bool odd = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const f32x4 v = pv[j][it];
            m4[j][it] = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
            odd |= !(m4[j][it] < 3.0e38f);
          }
        if (__builtin_expect(odd, 0)) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              unsigned mb = 0;
#pragma unroll
              for (int q = 0; q < 4; ++q) { const unsigned b = finite_abs_bits(pv[j][it][q]); mb = b > mb ? b : mb; }
              m4[j][it] = __uint_as_float(mb);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            float m = m4[j][it];
            m = fmaxf(m, dpp<DPP_XOR1>(m));
            m = fmaxf(m, dpp<DPP_XOR2>(m));
            m = fmaxf(m, dpp<DPP_HALF_MIRROR>(m));
            m4[j][it] = m;
          }
        if (g == 0) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int it = 0; it < 4; ++it) atomicMax(&s_rmax[cur][rbj[j] * 32 + l8 + 8 * it], __float_as_uint(m4[j][it]));
        }
        // bias gradient: column sums of the deltas (rows 8 apart share a lane; xor 8 by DPP, the rest by LDS atomics)
        if (MODE != 0 && p_bgrad) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            f32x4 c = pv[j][0] + pv[j][1] + pv[j][2] + pv[j][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] += dpp<DPP_ROR8>(c[q]);
            if ((lane_o & 8) == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) atomicAdd(p_bgrad + nbj[j] * 32 + g * 4 + q, c[q]);
            }
          }
        }
      };
      bool batched = false;
#pragma unroll 1
      for (int round = 0; round < nrounds; ++round) {
      jobs = 0; njobs = 0;
      auto job = [&](int j, int nb, int rb) { jobs |= (unsigned)(nb | (rb << 5)) << (8 * j); };
      int kl_nb0 = 0, kl_rb0 = 0, kl_nb1 = 0, kl_rb1 = 0;
      if (pair_mode) {
        kl_nb0 = wave + NWAVES * round;
        job(0, kl_nb0, 0); job(1, kl_nb0, 1); njobs = 2;
      } else {
        const int b0 = wave + 2 * NWAVES * round, b1 = b0 + NWAVES;
        if (b0 < nblk) { kl_nb0 = b0 / RB; kl_rb0 = b0 % RB; job(0, kl_nb0, kl_rb0); njobs = 1; }
        if (b1 < nblk) { kl_nb1 = b1 / RB; kl_rb1 = b1 % RB; job(1, kl_nb1, kl_rb1); njobs = 2; }
      }
      if (pair_mode) kloop(IRB{}, I0{}, kl_nb0, 0, 0, KS);
      else {
        if (njobs > 0) kloop(I1{}, I0{}, kl_nb0, kl_rb0, 0, KS);
        if (njobs > 1) kloop(I1{}, I1{}, kl_nb1, kl_rb1, 0, KS);
      }
      if (round == 0) stamp(li, 1);
      __builtin_amdgcn_wave_barrier();     // (a later round reuses the staging tiles)
      batched = BATCH_EPILOGUE && !a.side_blocked && !last && nrounds == 1 && njobs > 0 && (l_ld & 3) == 0 && rows == TM && (!p_rowbias || (l_N & 3) == 0) &&
                (row0 + TM) < (1LL << 31);
#pragma unroll
      for (int j = 0; j < 2; ++j) if (j < njobs && !((((jobs >> (8 * j)) & 31) * 32 + 31) < nlim)) batched = false;
      if (!batched)        // (batched: nrounds == 1, the epilogue follows the loop -- its results must not be live in here)
#pragma unroll 1
      for (int j = 0; j < njobs; ++j) {
        const int nb = (jobs >> (8 * j)) & 31, rb0 = (jobs >> (8 * j + 5)) & 7;
        const float winv = p_winv[nb];
        float* const stj = stage + j * STG;
        const int lane = fresh_lane();
        const int r = lane & 31, h = lane >> 5, g = lane & 7;
        // pass 1: acc0 + acc1 2^-11 -> staging tile (conflict-free: column groups GPS apart, rows 4 dwords apart)
        {
          float* dst = stj + (r >> 2) * GPS + (r & 3);
          if (j == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[acc_row(i, h) * 4] = fmaf(acc1[0][i], LO_INV, acc0[0][i]);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[acc_row(i, h) * 4] = fmaf(acc1[1][i], LO_INV, acc0[1][i]);
          }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // pass 2: 4 steps of 8 rows; lane = (column group g of 4 columns, row)
        const int n4 = nb * 32 + g * 4;
        const int mbase = rb0 * 32 + (lane >> 3);
        const long long off0 = (row0 + mbase) * l_ld + n4;
        // (forward: full blocks of full tiles take the batched path after the round loop; what arrives here is ragged, an
        // output layer, or a wave whose other block is ragged -- the barrier waits for that wave anyway.  Backward and
        // tangent keep the block-by-block fast path: their side loads come from HBM at the burst rate of the chip,
        // ~11 B / clk / CU, and requesting all of a wave's blocks at once measured 6 % slower than this spread-out order)
        const bool fast = !BATCH_EPILOGUE && !a.side_blocked && !last && (nb * 32 + 31 < nlim) && (l_ld & 3) == 0 && rows == TM && (!p_rowbias || (l_N & 3) == 0);
        f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
        float* lp = stj + g * GPS + (lane >> 3) * 4;
        if (fast) {
          f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
          if (MODE == 0 && p_bias) bias4 = *((gptr<const f32x4>)(p_bias + n4)) * b2;
          const float wb2 = winv * b2;
          // (requesting these before the k-loops does not help: vmcnt retires in order, the k-loop's weight fragments
          // would wait behind the HBM latency; requesting both jobs' at once measured the same)
          f32x4 hs[4];
          if (MODE != 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it) hs[it] = *((gptr<const f32x4>)(p_side_in + off0 + (long long)it * 8 * l_ld));
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long off = off0 + (long long)it * 8 * l_ld;
            const int row = mbase + 8 * it;
            const f32x4 z = *reinterpret_cast<const f32x4*>(lp + it * 32) * s_ainv[row];
            f32x4 v;
            if (MODE == 0) {
              f32x4 rb = {0.f, 0.f, 0.f, 0.f};
              if (p_rowbias) rb = *((gptr<const f32x4>)(p_rowbias + ((row0 + mbase + 8 * it) / rb_div) * (long long)l_N + n4));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float u = fmaf(z[q], wb2, bias4[q]);                 // b2 * (pre-activation)
                if (p_rowbias) u = fmaf(rb[q], b2, u);
                // softplus_beta(t) = (max(u, 0) + log2(1 + 2^-|u|)) ln2 / beta,  u = beta log2(e) t
                const float l2 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(u)));
                v[q] = (fmaxf(u, 0.f) + l2) * ib2sc;
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
            } else {
              f32x4 ex = {0.f, 0.f, 0.f, 0.f}, x2;
              if (MODE == 1 && p_side_add) ex = *((gptr<const f32x4>)(p_side_add + off));
              if (MODE == 2 && p_side_in2) ex = *((gptr<const f32x4>)(p_side_in2 + off));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float zz = z[q] * winv;
                const float e = __builtin_amdgcn_exp2f(nb2 * hs[it][q]);
                const float sp = (1.f - e) * sc;
                if (MODE == 1) v[q] = zz * sp + ex[q];
                else { v[q] = zz * sp; x2[q] = beta * zz * ex[q] * e; }
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
              if (MODE == 2 && p_side_out2) *((gptr<f32x4>)(p_side_out2 + off)) = x2;
              colsum += v;
            }
            row_max(row, v);
            *reinterpret_cast<f32x4*>(lp + it * 32) = v;       // parked for phase B (same lane reads it back)
          }
        } else {
        const bool vec_ok = (n4 + 3 < nlim);
        const bool vec_side = vec_ok && (l_ld & 3) == 0 && !a.side_blocked;
        // element (row, column n4 + q) of a side tensor: row-major, or point-blocked (ChainArgs::side_blocked; this kernel only
        // keeps such launches correct -- the layout is meant for the 128-point-tile kernel, mlp3w.hip)
        const bool blk = a.side_blocked != 0;
        auto sidx = [&](long long grow, long long off_rm, int q) -> long long {
          return blk ? ((grow >> 5) * l_ld + (n4 + q)) * 32 + (grow & 31) : off_rm + q;
        };
        f32x4 cm, bias4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) cm[q] = (n4 + q < nlim) ? 1.f : 0.f;
        if (MODE == 0 && p_bias) {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (n4 + q < l_N) bias4[q] = p_bias[n4 + q];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int m = mbase + 8 * it;
          const bool mrow = m < rows;
          const float rm = mrow ? 1.f : 0.f;
          const long long grow = row0 + m;
          const long long off = off0 + (long long)it * 8 * l_ld;
          const f32x4 pvs = *reinterpret_cast<const f32x4*>(lp + it * 32);      // acc0 + acc1 2^-11
          f32x4 z;
#pragma unroll
          for (int q = 0; q < 4; ++q) z[q] = out_z(pvs[q], s_ainv[m], winv);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (last) {
            if (mrow) {
              float* y = a.Y + grow * a.ldy + n4;
              f32x4 t = z;
              if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = out_add(z[q], bias4[q]);
              }
              if (vec_ok && (a.ldy & 3) == 0 && ((uintptr_t)a.Y & 15) == 0) {
                if (a.accum_y) {
                  const f32x4 y0 = *reinterpret_cast<const f32x4*>(y);
#pragma unroll
                  for (int q = 0; q < 4; ++q) t[q] = out_add(t[q], y0[q]);
                }
                *reinterpret_cast<f32x4*>(y) = t;
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (n4 + q < l_N) y[q] = a.accum_y ? out_add(t[q], y[q]) : t[q];
                }
              }
            }
            continue;
          }
          if (MODE == 0) {
            const float kk = fwd_kk(s_ainv[m], winv, b2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float u = fwd_u(pvs[q], kk, bias4[q] * b2);
              if (p_rowbias && mrow && n4 + q < l_N) u = fwd_u_rowbias(u, p_rowbias[(grow / rb_div) * (long long)l_N + n4 + q], b2);
              v[q] = (cm[q] != 0.f && mrow) ? softplus_u(u, ib2sc) : 0.f;
            }
            if (mrow && p_side_out) {
              if (vec_side) *((gptr<f32x4>)(p_side_out + off)) = v;
              else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (n4 + q < l_N) p_side_out[sidx(grow, off, q)] = v[q];
              }
            }
          } else {
            f32x4 hs = {0.f, 0.f, 0.f, 0.f}, ex = {0.f, 0.f, 0.f, 0.f}, x2 = {0.f, 0.f, 0.f, 0.f};
            if (mrow) {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (n4 + q < nlim) {
                hs[q] = p_side_in[sidx(grow, off, q)];
                if (MODE == 1 && p_side_add) ex[q] = p_side_add[sidx(grow, off, q)];
                if (MODE == 2 && p_side_in2) ex[q] = p_side_in2[sidx(grow, off, q)];
              }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float e = __builtin_amdgcn_exp2f(nb2 * hs[q]);
              const float sp = (1.f - e) * sc;
              const float mk = cm[q] * rm;
              if (MODE == 1) v[q] = (z[q] * sp + ex[q]) * mk;
              else { v[q] = z[q] * sp * mk; x2[q] = beta * z[q] * ex[q] * e * mk; }
            }
            if (MODE == 1 && is_skip && mrow && a.Xskip) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int n = n4 + q;
                if (n >= a.skip_split && n < l_N) a.Xskip[grow * a.ld_xskip + (n - a.skip_split)] = z[q] * sc;
              }
            }
            if (mrow) {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (n4 + q < nlim) {
                if (p_side_out) p_side_out[sidx(grow, off, q)] = v[q];
                if (MODE == 2 && p_side_out2) p_side_out2[sidx(grow, off, q)] = x2[q];
              }
            }
            colsum += v;
          }
          row_max(m, v);
          *reinterpret_cast<f32x4*>(lp + it * 32) = v;
        }
        }
        if (MODE != 0 && !last && p_bgrad) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float c = colsum[q];
            c += __shfl_xor(c, 8);
            c += __shfl_xor(c, 16);
            c += __shfl_xor(c, 32);
            if (lane < 8 && n4 + q < nlim) atomicAdd(p_bgrad + n4 + q, c);
          }
        }
      }
      }   // rounds
      // Backward / tangent: the epilogue's side loads come from HBM.  The CU's vector memory path returns in order, so a
      // wave that leaves its k-loop early and requests them stalls the weight fragments (L2 hits) of the waves still in
      // theirs: the loads are held back until every wave is through its k-loop (measured: k-loop of the slowest wave
      // 13.7 k -> ~10 k cycles per 256-wide layer).
      if (MODE != 0 && !last && BWD_KLOOP_BARRIER) __syncthreads();
      if (batched) {
        using TT = std::true_type;
        using FF = std::false_type;
        if (njobs == 2) { if (MODE == 0 && p_rowbias) phase_a_batched(I2{}, TT{}); else phase_a_batched(I2{}, FF{}); }
#ifndef NO_NJ1
        else { if (MODE == 0 && p_rowbias) phase_a_batched(I1{}, TT{}); else phase_a_batched(I1{}, FF{}); }
#endif
      }
      stamp(li, 2);

      if (last) {              // the output layer leaves nothing in the planes: no maximum, no split
        __syncthreads();
        stamp(li, 4);
        continue;
      }

      // ---- row maxima of this layer's outputs are in s_rmax[cur] (incl. the skip concatenation's input part) ----
      if (MODE != 1 && is_skip && tid < TM) {
        const float xm = __uint_as_float(s_xmax[xpar][tid]) * fabsf(a.skip_scale);
        atomicMax(&s_rmax[cur][tid], __float_as_uint(xm));
      }
      __syncthreads();          // every wave has read the planes and contributed its maxima
      stamp(li, 3);

      // ================= phase B: per-row scale, 2-way split, planes updated in place =================
      if (batched) {
        const int lane_o = fresh_lane();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j < njobs) {
            const int nb = (jobs >> (8 * j)) & 31, rb0 = (jobs >> (8 * j + 5)) & 7;
            const int n4 = nb * 32 + (lane_o & 7) * 4;
            const int mbase = rb0 * 32 + (lane_o >> 3);
            unsigned mx[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) mx[it] = s_rmax[cur][mbase + 8 * it];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              float s_row, inv_row;
              scale_from_max(mx[it], s_row, inv_row);
              put4(n4, mbase + 8 * it, pv[j][it], s_row);
            }
          }
        }
      } else
#pragma unroll 1
      for (int j = 0; j < njobs; ++j) {
        const int nb = (jobs >> (8 * j)) & 31, rb0 = (jobs >> (8 * j + 5)) & 7;
        const int lane = fresh_lane();
        const int g = lane & 7;
        const int n4 = nb * 32 + g * 4;
        const int mbase = rb0 * 32 + (lane >> 3);
        const float* lp = stage + j * STG + g * GPS + (lane >> 3) * 4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float s_row, inv_row;
          scale_from_max(s_rmax[cur][mbase + 8 * it], s_row, inv_row);
          put4(n4, mbase + 8 * it, *reinterpret_cast<const f32x4*>(lp + it * 32), s_row);
        }
      }
      if (tid < TM) {
        float s_row, inv_row;
        scale_from_max(s_rmax[cur][tid], s_row, inv_row);
        s_ainv[tid] = inv_row;           // read by the next layer's phase A, two barriers on
      }
      if (ly.side_amax && tid < 64) {
        const float wm = wave_max(__uint_as_float(tid < TM ? s_rmax[cur][tid] : 0u));
        if (tid == 0) atomicMax(ly.side_amax, __float_as_uint(wm));
      }

      // the planes beyond this layer's padded width must read as zero for the next layer's k-loop:
      // column blocks are written whole (32 columns), the k-loop reads multiples of 16 <= Np
      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output ----
      if (MODE != 1 && is_skip) {
        __syncthreads();
        const int K0 = a.K0, base = l_N;
        const float* X = a.X + row0 * a.ldx;
        for (int t = tid; t < K0 * TM; t += NTHREADS) {
          const int k = t % K0, m = t / K0;
          const float v = (m < rows) ? X[(long long)m * a.ldx + k] * a.skip_scale : 0.f;
          const int kk = base + k;
          float s_row, inv_row;
          scale_from_max(s_rmax[cur][m], s_row, inv_row);
          put1(kk, m, v, s_row);
          if (m < rows && ly.side_out)
            ly.side_out[a.side_blocked ? (((row0 + m) >> 5) * ly.ld_side + kk) * 32 + ((row0 + m) & 31) : (row0 + m) * ly.ld_side + kk] = v;
        }
        // zero the tail up to the next multiple of 16
        const int wcat = base + K0, wpad = (wcat + 15) & ~15;
        for (int t = tid; t < (wpad - wcat) * TM; t += NTHREADS) put1(wcat + t % (wpad - wcat), t / (wpad - wcat), 0.f, 1.f);
      }
      __syncthreads();
      if (tid < TM) s_rmax[cur][tid] = 0u;      // next use: two layers on, two barriers away
      cur ^= 1;
      stamp(li, 4);
    }
    if (tile == blockIdx.x) stamp(MAX_CHAIN_LAYERS - 1, 2);
    xpar ^= 1;
  }
  stamp(MAX_CHAIN_LAYERS - 1, 3);
  stamp_rt(1);
  if (rec) a.timeline[400 + 3 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  if (MODE != 0 && a.bg_total > 0) {
    __syncthreads();
    float* part = a.bg_partial + (long long)blockIdx.x * a.bg_total;
    for (int i = tid; i < a.bg_total; i += NTHREADS) part[i] = bsum[i];
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace x3

long long packed_size3(int K, int N, int transpose) {
  const int Kp = x3::round_up(transpose ? N : K, 16), Np = x3::round_up(transpose ? K : N, 32);
  return (long long)Kp * Np + x3::round_up(Np / 32, 4);       // two f16 planes + 1/scale per column block, in floats
}

int launch_pack3(const float* W, int ldw, float* dst, int K, int N, int transpose, hipStream_t stream) {
  const int Kp = x3::round_up(transpose ? N : K, 16), Np = x3::round_up(transpose ? K : N, 32);
  hipLaunchKernelGGL(x3::k_pack3, dim3(Np / 32), dim3(x3::PACK_T), 0, stream, W, ldw, reinterpret_cast<_Float16*>(dst),
                     dst + (long long)Kp * Np, K, N, transpose, Kp, Np);
  return ndjir_check_launch();
}

int launch_pack3_table(const PackEntry* table, int n, int total_blocks, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return NDJIR_OK;
  hipLaunchKernelGGL(x3::k_pack3_table, dim3(total_blocks), dim3(x3::PACK_T), 0, stream, table, n);
  return ndjir_check_launch();
}

int launch_chain3(const ChainArgs& a, int mode, hipStream_t stream) {
  using namespace x3;
  constexpr int LDS_DYN_MAX = 160 * 1024 - 2048;      // the kernel also holds 1.3 KB of static LDS (row maxima / scales)
  if (a.P <= 0) return NDJIR_OK;
  if (a.tile_rows == 128 || (a.tile_rows == 64 && a.forced_tile == 0)) {      // large launches: 128-point tiles where supported
    const int rc = launch_chainw(a, mode, stream);
    if (rc != NDJIR_ERR_UNSUPPORTED) return rc;
  }
  int TM = a.tile_rows == 32 ? 32 : 64;
  // widest activation the planes have to hold: chain input, every hidden output (+ skip concat)
  int wmax = round_up(a.K0, 16);
  for (int i = 0; i < a.L; ++i) {
    const bool last = a.has_output && i == a.L - 1;
    if (a.layers[i].Np > MAXNB * 32) return NDJIR_ERR_UNSUPPORTED;
    if (!last && a.layers[i].Np > NWAVES * 32) TM = 32;   // a hidden layer's results are parked in two staging tiles per wave
    if (!last && a.layers[i].Np > 32) { if (a.layers[i].Np > wmax) wmax = a.layers[i].Np; }
  }
  const int TMP = TM + 4;
  if (a.skip_layer >= 0 && mode != 1) { int w = round_up(a.layers[a.skip_layer].N + a.K0, 16); if (w > wmax) wmax = w; }
  ChainArgs b = a;
  b.K0p = round_up(a.K0, 16);
  b.lds_split = (wmax / 8) * TMP;                          // 16-byte units per plane
  size_t lds_bytes = (size_t)2 * b.lds_split * 16;
  size_t stage_bytes = (size_t)NWAVES * 2 * STG * 4;
  const size_t partials = (size_t)4 * TM * 32 * 4;
  if (stage_bytes < partials) stage_bytes = partials;
  lds_bytes += stage_bytes;
  b.n_tiles = (a.P + TM - 1) / TM;
  float* bg_ptr[MAX_CHAIN_LAYERS + 1];
  int bg_off[MAX_CHAIN_LAYERS + 1];
  int bg_n = 0, bg_total = 0;
  if (mode != 0) {
    for (int i = 0; i < a.L; ++i) if (a.layers[i].bgrad && !(a.has_output && i == a.L - 1)) {
      b.layers[i].bg_off = bg_total;
      bg_ptr[bg_n] = a.layers[i].bgrad;
      bg_off[bg_n] = bg_total;
      ++bg_n;
      bg_total += a.layers[i].N;
    } else b.layers[i].bgrad = nullptr;
  }
  if (mode != 0 && a.in_bgrad) {
    b.in_bg_off = bg_total;
    bg_ptr[bg_n] = a.in_bgrad;
    bg_off[bg_n] = bg_total;
    ++bg_n;
    bg_total += a.K0;
  }
  bg_total = (bg_total + 3) & ~3;       // (partial rows a multiple of 16 bytes long: the step's reduction reads them with 16-byte loads)
  b.bg_total = bg_total;
  b.bg_lds = (int)(lds_bytes / 4);
  lds_bytes += (size_t)bg_total * 4;
  if (bg_total > 0 && !a.bg_partial) return NDJIR_ERR_ARG;
  if (lds_bytes > LDS_DYN_MAX) return NDJIR_ERR_UNSUPPORTED;
  long long blocks = b.n_tiles;
  if (blocks > 256LL * 8) blocks = 256LL * 8;
  if (bg_total > 0 && blocks > CHAIN_MAX_GRID_BG) blocks = CHAIN_MAX_GRID_BG;
  static bool attr_set = false;
  if (!attr_set) {
#define NDJIR_SET(M, T) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain3<M, T>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX)
    NDJIR_SET(0, 64); NDJIR_SET(1, 64); NDJIR_SET(2, 64); NDJIR_SET(0, 32); NDJIR_SET(1, 32); NDJIR_SET(2, 32);
#undef NDJIR_SET
    attr_set = true;
  }
  if (a.dry) { snprintf(a.dry->name, 64, "ndjir::x3::k_chain3<%d, %d>", mode, TM); a.dry->blocks = (int)blocks; a.dry->bg_total = bg_total; return NDJIR_OK; }
#define NDJIR_GO(M, T) hipLaunchKernelGGL((k_chain3<M, T>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, b)
  if (TM == 64) { if (mode == 0) NDJIR_GO(0, 64); else if (mode == 1) NDJIR_GO(1, 64); else NDJIR_GO(2, 64); }
  else { if (mode == 0) NDJIR_GO(0, 32); else if (mode == 1) NDJIR_GO(1, 32); else NDJIR_GO(2, 32); }
#undef NDJIR_GO
  int rc = ndjir_check_launch();
  if (rc != NDJIR_OK) return rc;
  if (bg_total > 0 && !a.defer_bg_reduce) return launch_bgrad_reduce(a.bg_partial, (int)blocks, bg_total, bg_ptr, bg_off, bg_n, a.bg_accum, stream);
  return NDJIR_OK;
}

}  // namespace ndjir
