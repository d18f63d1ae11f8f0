// mlp3p.hip -- the f16x3 chain on 128-point tiles (mlp3w.hip's arithmetic, planes and side-tensor layout), software-PIPELINED
// across the two 64-point halves of the tile: a wave's k-loop MFMAs for one half carry the epilogue instructions of the other.
//
// Why (DESIGN.md 3.1, round 5's measurements): in mlp3w.hip a 256-wide layer of a 128-point tile takes 32 k cycles forward /
// 38 - 43 k backward for 12.4 k cycles of matrix work -- k-loop, activation epilogue and split run one after the other, and on
// gfx950 nothing overlaps them ACROSS waves: a wave stalled on the saturated matrix pipe of its SIMD blocks that SIMD's vector
// issue (tools/ubench/coissue.hip), so neither a second workgroup per CU (T64) nor two teams of one workgroup (TEAMS) helped.
// What does overlap is ONE wave issuing both: ~4 - 6 vector instructions ride in the shadow of each 32-cycle MFMA
// (tools/ubench/shadow.hip; 8 waves x (MFMA + 6 FMAs) = +10 % over the bare MFMA stream).
//
// Schedule.  A wave owns ONE column block of a hidden layer and, per half H0 = row blocks {0, 1}, H1 = {2, 3}, two 32 x 32
// accumulator pairs.  Unit u = (layer l, half h); phase p runs  K(unit p)  ||  E(unit p - 1):
//     phase (l, 0):  k-loop of layer l on H0 (reads the planes' rows of H0: layer l - 1's output, written one phase earlier)
//                    + epilogue of layer l - 1 on H1 (activation math -> row maxima -> [barrier] -> split -> planes' rows of H1)
//     phase (l, 1):  k-loop of layer l on H1  +  epilogue of layer l on H0
// One phase = 16 static slots of [one k-step: 6 MFMAs (3 partial products x 2 row blocks), the next k-step's activation
// fragments from LDS, the weight fragments three k-steps ahead from L2] + [one epilogue item: 4 consecutive features of one row
// block -- 8 activation items, the row-maximum barrier, 8 split items]; k-steps beyond 16 (skip-concatenated inputs) follow
// bare.  The planes are updated in place: a phase writes rows of one half while its MFMAs read the other; phases are separated
// by one barrier.  Per layer and tile a weight fragment now feeds 2 row blocks instead of 4: the L2 -> CU weight stream
// doubles (42 B / clk / CU at the matrix pipe's rate against the 52 measured by tools/ubench/wstream.hip).
//
// Arithmetic: the same three f16 partial products per fp32 product under the same power-of-two scales (per point row, per
// 32-column weight block) and the same forward expressions (mlp3_util.h) as mlp3.hip / mlp3w.hip -- but ONE fp32 accumulator per
// 32 x 32 block instead of two (the representation csrc/wgrad.hip uses since round 5): lo = f16(x s - hi) is kept UNSCALED in
// the planes, the packed weights' lo plane (stored x 2^11) is brought back by one v_pk_mul_f16 per fragment register, and
// hi hi' + hi lo' + lo hi' accumulate together; the matrix cores keep f16 denormals (tools/ubench/mfma_denorm.hip), so x s is
// held to 2^-22 relative or 2^-25 absolute with max |x s| in [2^14, 2^15).  That frees 64 of the wave's registers -- what lets
// the two roles of a phase live side by side without scratch (a scratch reload waits on `vmcnt(0)`, i.e. on every side store
// in flight: the first build of this kernel, with two accumulators, was bit-identical to mlp3w.hip and ran 4 x SLOWER for its
// 2 800 spilled registers) -- and the epilogue's 64 accumulator sums.  Results agree with the other two kernels to round-off
// (2e-6 relative), not bit for bit: this kernel therefore only takes the launches of a TRAINING pass (every hidden layer
// stores its side tensor); forward passes without side tensors -- the sampler's SDF rounds, whose values of different
// launches are merged bit for bit, the SDF volume, render_image -- stay on mlp3.hip / mlp3w.hip.  The output layer (and a
// narrow split-K output layer) runs un-pipelined after the last hidden epilogue has drained.
// Launched by launch_chainw_group (mlp3w.hip) for the nets it gives 4 row blocks per wave (hidden layers wider than 128
// columns), point-blocked or row-major side tensors; NDJIR_CHAINP=0 keeps mlp3w.hip's kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mlp.h"
#include "mlp3_util.h"

namespace ndjir {
namespace x3p {

using namespace x3u;

typedef ChainGroup __attribute__((address_space(4))) KGroup;
typedef ChainArgs __attribute__((address_space(4))) KArgs;

constexpr int TM = 128, TMP = TM + 4, NWAVES = 8, NTHREADS = NWAVES * 64, HALF = 64;
constexpr int IN_CACHE = 10;

__device__ __forceinline__ int acc_feat(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

struct OneNet {
  const ChainArgs& a;
  __device__ __forceinline__ const ChainArgs& get(int) const { return a; }
  __device__ __forceinline__ int n() const { return 1; }
};
struct ManyNets {
  const KGroup* g;
  __device__ __forceinline__ const KArgs& get(int i) const { return g->net[i]; }
  __device__ __forceinline__ int n() const { return g->n; }
};

template <int V>
using IC = std::integral_constant<int, V>;
using TT = std::true_type;
using FF = std::false_type;

// LDS traffic is ordered by hand around the mid-phase / end-of-phase barriers: only the LDS counter is drained -- the weight
// fragments in flight from L2 (and the side loads / stores) stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// pointers out of the argument blocks: forced into scalar registers (in the group kernel the compiler cannot always prove an
// argument block's address uniform -- `pin` alone then fails with "illegal VGPR to SGPR copy")
template <class T>
__device__ __forceinline__ gptr<T> upin(T* p) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (gptr<T>)reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// single operations that must not be contracted into fused multiply-adds (the forward expressions of mlp3_util.h, taken apart
// so that their pieces can be placed between the MFMAs of a slot)
__device__ __forceinline__ float add_nc(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float mul_nc(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float sub_nc(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// the two-way split with an UNSCALED low part:  x s = hi + lo,  lo = f16(x s - hi)  (x s - hi is exact in fp32)
__device__ __forceinline__ void split4u(f32x4 v, float s, f16x4& ph, f16x4& pl) {
#pragma clang fp contract(off)
  const f32x4 xs = v * s;
  ph = __builtin_convertvector(xs, f16x4);
  pl = __builtin_convertvector(xs - __builtin_convertvector(ph, f32x4), f16x4);
}
__device__ __forceinline__ void split1u(float v, float s, _Float16& ph, _Float16& pl) {
#pragma clang fp contract(off)
  const float xs = v * s;
  ph = (_Float16)xs;
  pl = (_Float16)(xs - (float)ph);
}

template <int MODE, class NETS>
__device__ __forceinline__ void chainp_body(const NETS nets) {
  constexpr bool BWD = (MODE == 1);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned s_rmax[TM];       // per row: largest finite |output| of the layer in its epilogue (bit pattern)
  __shared__ unsigned s_xmax[2][TM];    // per row: largest finite |x| of the chain input tile (ping-pong by tile)
  __shared__ float s_ainv[TM];          // per row: 1 / scale of the planes' current content
  __shared__ __attribute__((aligned(16))) float s_bias[2][256];   // forward: bias * beta log2(e) of the layer in its epilogue (ping-pong
                                        // by layer; zero beyond the layer's width) -- 16 registers per lane the allocator does not have
  const int n_nets = nets.n();
  const auto& a0 = nets.get(0);
  const int PLANE = a0.lds_split;       // 16-byte units per plane
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  f16x8* act = reinterpret_cast<f16x8*>(lds);
  char* actb = reinterpret_cast<char*>(act);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto stamp = [&](int li, int phase) {
    if (a0.timeline && blockIdx.x == 0 && lane == 0) a0.timeline[(li * 5 + phase) * 8 + wave] = (long long)__builtin_amdgcn_s_memtime();
  };
  auto stamp_rt = [&](int phase) {
    if (a0.timeline && blockIdx.x == 0 && lane == 0) a0.timeline[((MAX_CHAIN_LAYERS - 2) * 5 + phase) * 8 + wave] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  auto put4 = [&](int k, int m, f32x4 v, float s) {
    f16x4 ph, pl;
    split4u(v, s, ph, pl);
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<f16x4*>(p) = ph;
    *reinterpret_cast<f16x4*>(p + (size_t)PLANE * 16) = pl;
  };
  auto put1 = [&](int k, int m, float v, float s) {
    _Float16 ph, pl;
    split1u(v, s, ph, pl);
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<_Float16*>(p) = ph;
    *reinterpret_cast<_Float16*>(p + (size_t)PLANE * 16) = pl;
  };

  stamp(MAX_CHAIN_LAYERS - 1, 0);
  stamp_rt(0);
#ifndef NDJIR_NO_L2_WARMUP
  // (mlp3w.hip: the first round's workgroups of an XCD touch every layer's packed weights once, all layers' misses overlap)
  unsigned warm = 0;
  if (blockIdx.x < 256) {
    const int part = (blockIdx.x >> 3) & 31;
    for (int ni = 0; ni < n_nets; ++ni) {
      const auto& a = nets.get(ni);
      for (int li = 0; li < a.L; ++li) {
        const auto& ly = a.layers[li];
        const long long lines = (((long long)((ly.Kp + 15) >> 4) * 16 * ly.Np + (ly.Np >> 5)) * 4 + 127) >> 7;
        const long long per = (lines + 31) >> 5;
        const unsigned* base = reinterpret_cast<const unsigned*>(ly.Wp);
        for (long long l = part * per + tid; l < (part + 1) * per && l < lines; l += NTHREADS) warm ^= base[l * 32];
      }
    }
  }
#endif
  if (MODE != 0)
    for (int ni = 0; ni < n_nets; ++ni) {
      float* bs = lds + nets.get(ni).bg_lds;
      for (int i = tid; i < nets.get(ni).bg_total; i += NTHREADS) bs[i] = 0.f;
    }
  if (tid < TM) s_rmax[tid] = 0u;
  if (tid < 2 * TM) (&s_xmax[0][0])[tid] = 0u;
  __syncthreads();
  int xpar = 0;
  bool warm_pending = true;

  for (long long tile = blockIdx.x; tile < a0.n_tiles; tile += gridDim.x) {
    const long long row0 = tile * TM;
   for (int ni = 0; ni < n_nets; ++ni) {
    const auto& a = nets.get(ni);
    float* const bsum = lds + a.bg_lds;
    const float beta = a.beta;

    // ---- chain input tile -> planes (zero padded to a multiple of 16 features): as mlp3w.hip ----
    {
      const int K0p = a.K0p, K0 = a.K0;
      const float* X = a.X + row0 * a.ldx;
      int groups = K0p >> 2;
      asm volatile("" : "+s"(groups));
      const int total = groups * TM;
      auto load = [&](int t) -> f32x4 {
        const int g = t % groups, m = t / groups;
        const int k = g * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < K0) v = *reinterpret_cast<const f32x4u*>(X + (long long)m * a.ldx + k);
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (k + q < K0) v[q] = X[(long long)m * a.ldx + k + q];
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
        s_xmax[xpar ^ 1][tid] = 0u;
      }
      if (a.x_amax && tid < TM) {
        const float wm = wave_max(__uint_as_float(s_xmax[xpar][tid]));
        if (lane == 0) atomicMax(a.x_amax, __float_as_uint(wm));
      }
      auto emit = [&](int t, f32x4 v) {
        const int g = t % groups, m = t / groups;
        const int k = g * 4;
        float s_in, inv_in;
        scale_from_max(s_xmax[xpar][m], s_in, inv_in);
        put4(k, m, v, s_in);
        if (MODE != 0 && a.in_bgrad) {
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
#ifndef NDJIR_NO_L2_WARMUP
    if (warm_pending) { asm volatile("" :: "v"(warm)); warm_pending = false; }
#endif
    __syncthreads();
    if (tile == blockIdx.x) stamp(MAX_CHAIN_LAYERS - 1, 1);

    // Accumulators of the wave's four 32 x 32 blocks: H0 = blocks 0, 1; H1 = blocks 2, 3 (static indices only).
    f32x16 acc0[4];
    const _Float16 LOU = (_Float16)LO_INV;          // 2^-11: the packed weights' lo plane back to its unscaled value

    // =====================================================================================================================
    // One phase:  K(layer liK, half HK)  ||  E(layer liE, half 1 - HK).   doK / doE: uniform.
    // =====================================================================================================================
    auto phase = [&](auto hk_tag, const int liK, const bool doK, const int liE, const bool doE, const int stamp_li, const int stamp_ph) {
      constexpr int HK = decltype(hk_tag)::value, HE = 1 - HK;
      constexpr int KB = 2 * HK, EB = 2 * HE;            // first accumulator block of the K / E role
      const int nb = wave;                                // the wave's column block (hidden layers: at most 8)
      stamp(stamp_li, stamp_ph);
      // rows of the K half: their maxima were consumed by the previous phase's split -- re-arm them for the next phase's epilogue
      if (tid >= TM && tid < TM + HALF) s_rmax[HK * HALF + (tid - TM)] = 0u;

      // forward: the bias table of layer liK, read by its two epilogues in the next two phases
      if (MODE == 0 && HK == 0 && doK && tid < 256) {
        const auto& lw = a.layers[liK];
        s_bias[liK & 1][tid] = (lw.bias && tid < lw.N) ? lw.bias[tid] * (beta * LOG2E) : 0.f;
      }
      // ---------------- K role: set-up ----------------
      const auto& lyK = a.layers[doK ? liK : 0];
      const int KS = doK ? (lyK.Kp + 15) >> 4 : 0;
      const bool activeK = doK && nb < (lyK.Np >> 5);
      const gptr<const f16x8> p_wp = (gptr<const f16x8>)upin(lyK.Wp);
      f16x8 b[3][2];            // weight fragments [slot][plane], three k-steps ahead
      f16x8 af[2][2][2];        // activation fragments [buffer][plane][row block of the half]
      gptr<const f16x8> Bp = nullptr;
      const f16x8* A0 = nullptr;
      // One k-step of the half: slot S of the weight ring, buffer C of the activation fragments; per block the MFMA order is
      // w_hi x_lo, w_hi x_hi, w_lo x_hi into the block's one accumulator.  No branches: the prefetches past the last
      // k-step re-read it (clamped indices) -- a branch would end the basic block, and with it the scheduling region in
      // which the epilogue item's vector instructions are placed between these MFMAs.
      auto kstep = [&](auto stag, auto ctag, const int ks, auto first_tag) {
        constexpr int S = decltype(stag)::value, C = decltype(ctag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const int ksn = ks + 1 < KS ? ks + 1 : KS - 1;
        const f16x8* An = A0 + 2 * ksn * TMP;
#pragma unroll
        for (int q = 0; q < 2; ++q) { af[C ^ 1][0][q] = An[q * 32]; af[C ^ 1][1][q] = An[PLANE + q * 32]; }
        const f32x16 zero = f32x16{0};
#pragma unroll
        for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][1][q], FIRST ? zero : acc0[KB + q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][0][q], acc0[KB + q], 0, 0, 0);
        const f16x8 blo = b[S][1] * LOU;
#pragma unroll
        for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, af[C][0][q], acc0[KB + q], 0, 0, 0);
        const int ksw = ks + 3 < KS ? ks + 3 : KS - 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) b[S][p] = Bp[(long long)(ksw * 2 + p) * 64];
      };

      // ---------------- E role: set-up ----------------
      const auto& lyE = a.layers[doE ? liE : 0];
      const int NBE = lyE.Np >> 5;
      const bool activeE = doE && nb < NBE;
      const gptr<const float> p_winv = upin(lyE.Wp + (long long)((lyE.Kp + 15) >> 4) * 16 * lyE.Np);
      const gptr<const float> p_rowbias = upin((MODE == 0 && liE == 0) ? a.row_bias : nullptr);
      const int rb_div = pin(a.row_bias_div > 0 ? a.row_bias_div : 1);
      const gptr<const float> p_side_in = upin(lyE.side_in);
      const gptr<const float> p_side_ex = upin(MODE == 1 ? lyE.side_add : lyE.side_in2);
      const gptr<float> p_side_out = upin(lyE.side_out);
      const gptr<float> p_side_out2 = upin(lyE.side_out2);
      float* const p_bgrad = (MODE != 0 && lyE.bgrad) ? bsum + pin(lyE.bg_off) : nullptr;
      const int l_N = pin(lyE.N);
      const int l_ld = pin(lyE.ld_side);
      const bool is_skip = doE && (liE == a.skip_layer);
      const float sc = pin(is_skip ? a.skip_scale : 1.f);
      const int nlim = pin((BWD && is_skip) ? a.skip_split : l_N);
      const float b2 = beta * LOG2E, ib2sc = LN2 / beta * sc;
      const float hsc = (MODE != 0 && is_skip) ? 1.f / sc : 1.f;
      const float nb2 = -b2 * hsc;
      const bool blk = a.side_blocked != 0;
      const bool has_ex = MODE == 2 || (MODE == 1 && p_side_ex != nullptr);
      const long long tile_off = row0 * l_ld;
      // forward: the skip concatenation's input part enters the row maxima (before the mid-phase barrier)
      if (MODE != 1 && is_skip && tid < HALF) {
        const float xm = __uint_as_float(s_xmax[xpar][HE * HALF + tid]) * fabsf(a.skip_scale);
        atomicMax(&s_rmax[HE * HALF + tid], __float_as_uint(xm));
      }
      int lane_o = 0, r_o = 0, hh = 0, fb = 0;
      float winvc = 0.f, sa[2] = {0.f, 0.f}, mrow[2] = {0.f, 0.f};
      f32x4 hs[3], ex[3];                 // backward / tangent: side loads of the activation items, three items ahead
      float csum[16];                     // fast path, backward: column sums of the half's deltas (bias gradient)
      bool full = false;
      // An item comes in a FAST form -- straight-line code: full column block, point-blocked side tensors, no per-row-group
      // term, every side output present (backward: OPT = an extra adjoint is added) -- and in the general form (FULL: the
      // block lies inside the layer's columns), which may branch.
      // side loads of activation item IT (block J = IT / 4 of the half, feature group g = IT % 4) into ring slot IT % 3
      auto side_load = [&](auto it_tag, auto fast_tag, auto opt_tag, auto full_tag) {
        constexpr int IT = decltype(it_tag)::value, J = IT / 4, g = IT % 4, SL = IT % 3;
        constexpr bool FAST = decltype(fast_tag)::value, OPT = decltype(opt_tag)::value, FULL = FAST || decltype(full_tag)::value;
        if constexpr (MODE != 0) {
          const bool hx = MODE == 2 ? true : (FAST ? OPT : has_ex);
          const int rbJ = EB + J;
          if (FAST || blk) {
            const unsigned boff = ((unsigned)rbJ * (unsigned)l_ld + (unsigned)fb) * 32u + (unsigned)r_o;
            const gptr<const float> b_in = p_side_in + tile_off + boff;
#pragma unroll
            for (int q = 0; q < 4; ++q) hs[SL][q] = (FULL || fb + 8 * g + q < nlim) ? b_in[(8 * g + q) * 32] : 0.f;
            if (hx) {
              const gptr<const float> b_ex = p_side_ex + tile_off + boff;
#pragma unroll
              for (int q = 0; q < 4; ++q) ex[SL][q] = (FULL || fb + 8 * g + q < nlim) ? b_ex[(8 * g + q) * 32] : 0.f;
            } else ex[SL] = f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
            // row-major side tensors: the lane's own point, 4 consecutive features
            const unsigned rowoff = (unsigned)(rbJ * 32 + r_o) * (unsigned)l_ld + (unsigned)fb + 8 * g;
            const gptr<const float> b_in = p_side_in + tile_off + rowoff;
            if (FULL && (l_ld & 3) == 0) hs[SL] = *((gptr<const f32x4>)b_in);
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) hs[SL][q] = (fb + 8 * g + q < nlim) ? b_in[q] : 0.f;
            }
            if (hx) {
              const gptr<const float> b_ex = p_side_ex + tile_off + rowoff;
              if (FULL && (l_ld & 3) == 0) ex[SL] = *((gptr<const f32x4>)b_ex);
              else {
#pragma unroll
                for (int q = 0; q < 4; ++q) ex[SL][q] = (fb + 8 * g + q < nlim) ? b_ex[q] : 0.f;
              }
            } else ex[SL] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      };
      // activations (forward) / deltas (backward, tangent) of group g of block J -> side_out
      auto side_store = [&](auto j_tag, auto g_tag, auto fast_tag, auto opt_tag, auto full_tag) {
        constexpr int J = decltype(j_tag)::value, g = decltype(g_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value, OPT = decltype(opt_tag)::value, FULL = FAST || decltype(full_tag)::value;
        const bool so = FAST || p_side_out != nullptr;
        if (!so) return;
#ifdef NDJIR_CHAINP_X_NOSTORE      // (timing experiment, WRONG results: the fast form without its side stores)
        if (FAST) return;
#endif
        const int rbJ = EB + J;
        const int lim = MODE == 0 ? l_N : nlim;
        const gptr<float> b_out = p_side_out + tile_off;
        if (FAST || blk) {
          const gptr<float> b_blk = b_out + (((unsigned)rbJ * (unsigned)l_ld + (unsigned)fb) * 32u + (unsigned)r_o);
#pragma unroll
          for (int q = 0; q < 4; ++q) if (FULL || fb + 8 * g + q < lim) b_blk[(8 * g + q) * 32] = acc0[EB + J][4 * g + q];
        } else {
          const unsigned rowoff = (unsigned)(rbJ * 32 + r_o) * (unsigned)l_ld + (unsigned)fb + 8 * g;
          if (FULL && (l_ld & 3) == 0)
            *((gptr<f32x4>)(b_out + rowoff)) = f32x4{acc0[EB + J][4 * g], acc0[EB + J][4 * g + 1], acc0[EB + J][4 * g + 2], acc0[EB + J][4 * g + 3]};
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (fb + 8 * g + q < lim) b_out[rowoff + q] = acc0[EB + J][4 * g + q];
          }
        }
      };
      // activation item IT: the math of 4 consecutive features of one row block, on the accumulator registers (mlp3w.hip's
      // hidden_block, one feature group at a time)
      auto act_item = [&](auto it_tag, auto fast_tag, auto opt_tag, auto full_tag) {
        constexpr int IT = decltype(it_tag)::value, J = IT / 4, g = IT % 4, SL = IT % 3;
        constexpr bool FAST = decltype(fast_tag)::value, FULL = FAST || decltype(full_tag)::value;
        const int rbJ = EB + J;
        const int R = rbJ * 32 + r_o;
        if constexpr (MODE == 0) {
          const float kk = fwd_kk(sa[J], winvc, b2);
          const f32x4 bb = *reinterpret_cast<const f32x4*>(&s_bias[liE & 1][fb + 8 * g]);      // bias * beta log2(e)
          const bool rbz = !FAST && p_rowbias != nullptr;
          f32x4 rbv = {0.f, 0.f, 0.f, 0.f};
          if (rbz) {
            const gptr<const float> rbp = p_rowbias + (long long)((unsigned)(row0 + R) / (unsigned)rb_div) * l_N + fb;
            if (FULL) rbv = *((gptr<const f32x4>)(rbp + 8 * g));
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (fb + 8 * g + q < l_N) rbv[q] = rbp[8 * g + q];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 4 * g + q;
            float u = fwd_u(acc0[EB + J][i], kk, bb[q]);
            if (rbz) u = fwd_u_rowbias(u, rbv[q], b2);
            float v = softplus_u(u, ib2sc);
            if (!FULL) v = (fb + 8 * g + q < nlim) ? v : 0.f;
            acc0[EB + J][i] = v;
          }
        } else {
          const float saw = sa[J] * winvc;
          f32x4 x2;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 4 * g + q;
            const float zz = acc0[EB + J][i] * saw;
            const float e = __builtin_amdgcn_exp2f(nb2 * hs[SL][q]);
            const float sp = __builtin_fmaf(-e, sc, sc);
            float v;
            // (without an extra adjoint ex holds zeros: fma(zz, sp, 0) is the rounded product, the value mlp3w.hip computes)
            if (MODE == 1) v = zz * sp + ex[SL][q];
            else { v = zz * sp; x2[q] = beta * zz * ex[SL][q] * e; }
            if (!FULL) {
              const int f = fb + 8 * g + q;
              if (MODE == 1 && is_skip && a.Xskip && f >= a.skip_split && f < l_N)
                a.Xskip[(row0 + R) * a.ld_xskip + (f - a.skip_split)] = zz * sc;
              if (f >= nlim) { v = 0.f; x2[q] = 0.f; }
            }
            acc0[EB + J][i] = v;
          }
          if (MODE == 2 && (FAST || p_side_out2)) {
            const gptr<float> b_out2 = p_side_out2 + tile_off;
            if (FAST || blk) {
              const gptr<float> b_blk = b_out2 + (((unsigned)rbJ * (unsigned)l_ld + (unsigned)fb) * 32u + (unsigned)r_o);
#pragma unroll
              for (int q = 0; q < 4; ++q) if (FULL || fb + 8 * g + q < nlim) b_blk[(8 * g + q) * 32] = x2[q];
            } else {
              const unsigned rowoff = (unsigned)R * (unsigned)l_ld + (unsigned)fb + 8 * g;
              if (FULL && (l_ld & 3) == 0) *((gptr<f32x4>)(b_out2 + rowoff)) = x2;
              else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (fb + 8 * g + q < nlim) b_out2[rowoff + q] = x2[q];
              }
            }
          }
        }
        // (the backward's stores wait for the split items: every side load of the phase is consumed by then; in the fast form
        //  the forward's / tangent's are spread over the phase's 16 slots, one group every other slot -- `slot` below: with all
        //  eight groups stored from the eight activation slots every CU of the chip writes its 64 KB at the same time and the
        //  activation half of the phase runs at the HBM's pace, 10 - 13 k cycles for 3 k cycles of matrix work)
        if (MODE != 1 && !FAST) side_store(IC<J>{}, IC<g>{}, fast_tag, opt_tag, full_tag);
        float m = g == 0 ? 0.f : mrow[J];
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(acc0[EB + J][4 * g + q]));
        mrow[J] = m;
        // refill the ring slot just consumed with the loads of the item three ahead
        if constexpr (MODE != 0 && IT + 3 < 8) side_load(IC<IT + 3>{}, fast_tag, opt_tag, full_tag);
      };
      // row maxima of the two blocks (after the 8 activation items): v_max ignores NaN; a set holding an Inf (or only NaN)
      // goes through the bit-pattern filter.  One LDS atomic per lane and block.
      auto rowmax_finish = [&]() {
#pragma unroll
        for (int J = 0; J < 2; ++J) {
          float m = mrow[J];
          if (!(m < 3.0e38f)) {
            unsigned mb = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const unsigned bb = finite_abs_bits(acc0[EB + J][i]); mb = bb > mb ? bb : mb; }
            m = __uint_as_float(mb);
          }
          atomicMax(&s_rmax[(EB + J) * 32 + r_o], __float_as_uint(m));
        }
      };
      float s_row[2] = {1.f, 1.f};
      // split item IT (block J, feature group g): scale by the row's power of two, two f16 planes, in place; backward: the
      // group's delta store; two features of the bias gradient's column sums
      auto split_item = [&](auto it_tag, auto fast_tag, auto opt_tag, auto full_tag) {
        constexpr int IT = decltype(it_tag)::value, J = IT / 4, g = IT % 4;
        constexpr bool FAST = decltype(fast_tag)::value;
        const int rbJ = EB + J;
        const int R = rbJ * 32 + r_o;
        if (g == 0) {
          float inv_row;
          scale_from_max(s_rmax[R], s_row[J], inv_row);
        }
        put4(nb * 32 + 4 * hh + 8 * g, R, f32x4{acc0[EB + J][4 * g], acc0[EB + J][4 * g + 1], acc0[EB + J][4 * g + 2], acc0[EB + J][4 * g + 3]}, s_row[J]);
        if (MODE == 1) side_store(IC<J>{}, IC<g>{}, fast_tag, opt_tag, full_tag);
        if (MODE != 0 && (FAST ? MODE == 1 : p_bgrad != nullptr)) {
          // bias gradient: column sums of the deltas over the wave's 64 points of the half -- 16-lane rows by DPP; the LDS
          // atomics (first lane of every row) follow the last slot in the fast form
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int i = 2 * IT + t;
            float c = acc0[EB][i] + acc0[EB + 1][i];
            c += dpp<DPP_XOR1>(c);
            c += dpp<DPP_XOR2>(c);
            c += dpp<DPP_HALF_MIRROR>(c);
            c += dpp<DPP_MIRROR>(c);
            if (FAST) csum[i] = c;
            else {
              const int f = fb + acc_feat(i, 0);
              if ((lane_o & 15) == 0 && f < nlim) atomicAdd(p_bgrad + f, c);
            }
          }
        }
      };
      if (activeE) {
        lane_o = fresh_lane(); r_o = lane_o & 31; hh = lane_o >> 5;
        fb = nb * 32 + 4 * hh;
        const bool full_cols = nb * 32 + 31 < nlim;
        full = blk ? (full_cols && (!p_rowbias || (l_N & 3) == 0))
                   : ((l_ld & 3) == 0 && (!p_rowbias || (l_N & 3) == 0) && full_cols);
        const float winv_raw = p_winv[nb];
        if (MODE != 0) {
          if (full) { side_load(IC<0>{}, FF{}, FF{}, TT{}); side_load(IC<1>{}, FF{}, FF{}, TT{}); side_load(IC<2>{}, FF{}, FF{}, TT{}); }
          else { side_load(IC<0>{}, FF{}, FF{}, FF{}); side_load(IC<1>{}, FF{}, FF{}, FF{}); side_load(IC<2>{}, FF{}, FF{}, FF{}); }
        }
        sa[0] = s_ainv[EB * 32 + r_o];
        sa[1] = s_ainv[(EB + 1) * 32 + r_o];
        winvc = pin(winv_raw);
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---------------- the 16 slots ----------------
      // ---------------- K role: first fragments (after the E role's set-up: its loads and the fragments would otherwise be live
      // together with it, and the allocator spills them) ----------------
      auto k_preload = [&]() {
       if (activeK) {
        // (no zero-initialisation: the first k-step's MFMAs take the constant 0 as their addend, so the K half's accumulators
        //  come to life inside slot 0 -- after the E half's second accumulators have died in its set-up)
        const int lane_k = fresh_lane();
        Bp = p_wp + ((long long)nb * KS) * 2 * 64 + lane_k;
        A0 = act + (lane_k >> 5) * TMP + KB * 32 + (lane_k & 31);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int p = 0; p < 2; ++p) b[s][p] = Bp[(long long)((s < KS ? s : KS - 1) * 2 + p) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) { af[0][0][q] = A0[q * 32]; af[0][1][q] = A0[PLANE + q * 32]; }
        __builtin_amdgcn_sched_barrier(0);
       }
      };
      // ---------------- fast path: the pieces of a slot ----------------
      f16x8 blo;                          // this k-step's weight lo fragment, unscaled
      f32x4 fu, fa, fm;                   // item temporaries: forward u, exp / log, max(u, 0); backward zz, e, softplus'
      f32x4 bbn = {0.f, 0.f, 0.f, 0.f};   // forward: bias * beta log2(e) of the NEXT item (read one slot ahead)
      f32x4 xs4, bk4;                     // split temporaries
      f16x4 ph4;
      // MFMA j of slot I's k-step: j = 2 * product + row block of the half; activation fragments of buffer I % 2
      // (-DNDJIR_CHAINP_X_AF1: one buffer, each fragment re-read right behind the last MFMA that uses it -- 16 registers less, but
      //  the hi planes then arrive two MFMAs before their first use and every slot waits for the LDS)
#ifdef NDJIR_CHAINP_X_AF1
#define NDJIR_AFC(I) 0
#else
#define NDJIR_AFC(I) ((I) % 2)
#endif
      auto mfma_j = [&](auto i_tag, auto j_tag) {
        constexpr int I = decltype(i_tag)::value, Jv = decltype(j_tag)::value, S = I % 3, q = Jv & 1, pr = Jv >> 1, C = NDJIR_AFC(I);
#ifdef NDJIR_CHAINP_X_NOMFMA      // (timing experiment, WRONG results: the fast form without its matrix work)
        acc0[KB + q][0] += (float)b[S][0][0] + (float)af[C][pr == 0][q][0] + (float)blo[0];
        return;
#endif
        const f32x16 zero = f32x16{0};
        if constexpr (pr == 0) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][1][q], I == 0 ? zero : acc0[KB + q], 0, 0, 0);
        else if constexpr (pr == 1) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][0][q], acc0[KB + q], 0, 0, 0);
        else acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, af[C][0][q], acc0[KB + q], 0, 0, 0);
      };
      // the k-step's own loads, each behind the last MFMA that reads the registers it overwrites
      auto k_side = [&](auto i_tag, auto j_tag) {
        constexpr int I = decltype(i_tag)::value, Jv = decltype(j_tag)::value, S = I % 3;
        const int ksn = I + 1 < KS ? I + 1 : KS - 1;
        const f16x8* An = A0 + 2 * ksn * TMP;
#ifdef NDJIR_CHAINP_X_NOPKMUL      // (timing experiment, WRONG results: the scaled lo plane as it is)
        if constexpr (Jv == 0) blo = b[S][1];
#else
        if constexpr (Jv == 0) blo = b[S][1] * LOU;
#endif
#ifdef NDJIR_CHAINP_X_NOAF         // (timing experiment, WRONG results: the activation fragments are never re-read)
        if constexpr (Jv == 3) {
          const int ksw = I + 3 < KS ? I + 3 : KS - 1;
#pragma unroll
          for (int p = 0; p < 2; ++p) b[S][p] = Bp[(long long)(ksw * 2 + p) * 64];
        }
        return;
#endif
#ifdef NDJIR_CHAINP_X_AF1
        if constexpr (Jv == 1) { af[0][1][0] = An[PLANE]; af[0][1][1] = An[PLANE + 32]; }      // lo planes of the next k-step
        if constexpr (Jv == 5) { af[0][0][0] = An[0]; af[0][0][1] = An[32]; }                   // hi planes of the next k-step
#else
        constexpr int Cn = (I % 2) ^ 1;
        if constexpr (Jv == 0) { af[Cn][1][0] = An[PLANE]; af[Cn][1][1] = An[PLANE + 32]; }    // the next k-step's fragments: a slot ahead
        if constexpr (Jv == 1) { af[Cn][0][0] = An[0]; af[Cn][0][1] = An[32]; }
#endif
        if constexpr (Jv == 3) {
#ifdef NDJIR_CHAINP_X_WSAME      // (timing experiment, WRONG results: every k-step re-reads the first weight fragment -- an L1 hit)
          const int ksw = 0;
#else
          const int ksw = I + 3 < KS ? I + 3 : KS - 1;
#endif
#pragma unroll
          for (int p = 0; p < 2; ++p) b[S][p] = Bp[(long long)(ksw * 2 + p) * 64];
        }
      };
      auto load_bias_next = [&](const int it) {      // forward: item `it`'s bias * beta log2(e) out of the LDS table
        const int g = it & 3;
        bbn = *reinterpret_cast<const f32x4*>(&s_bias[liE & 1][fb + 8 * g]);
      };
      // slice Jv of activation item IT (fast form: full block, point-blocked side tensors, every output present)
      auto act_slice = [&](auto it_tag, auto j_tag, auto opt_tag) {
        constexpr int IT = decltype(it_tag)::value, Jv = decltype(j_tag)::value, J = IT / 4, g = IT % 4, SL = IT % 3;
        if constexpr (MODE == 0) {
          if constexpr (Jv == 0) {
            const float kk = fwd_kk(sa[J], winvc, b2);
#pragma unroll
            for (int q = 0; q < 4; ++q) { fu[q] = fwd_u(acc0[EB + J][4 * g + q], kk, bbn[q]); fa[q] = __builtin_amdgcn_exp2f(-__builtin_fabsf(fu[q])); }
          } else if constexpr (Jv == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa[q] = add_nc(1.f, fa[q]); fm[q] = __builtin_fmaxf(fu[q], 0.f); }
          } else if constexpr (Jv == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fa[q] = __builtin_amdgcn_logf(fa[q]);
          } else if constexpr (Jv == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc0[EB + J][4 * g + q] = mul_nc(add_nc(fm[q], fa[q]), ib2sc);
          }
        } else {
          if constexpr (Jv == 0) {
            const float saw = sa[J] * winvc;
#pragma unroll
            for (int q = 0; q < 4; ++q) { fu[q] = acc0[EB + J][4 * g + q] * saw; fa[q] = nb2 * hs[SL][q]; }
          } else if constexpr (Jv == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fa[q] = __builtin_amdgcn_exp2f(fa[q]);
          } else if constexpr (Jv == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fm[q] = __builtin_fmaf(-fa[q], sc, sc);
          } else if constexpr (Jv == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (MODE == 1) acc0[EB + J][4 * g + q] = fu[q] * fm[q] + ex[SL][q];
              else { acc0[EB + J][4 * g + q] = fu[q] * fm[q]; fm[q] = beta * fu[q] * ex[SL][q] * fa[q]; }
            }
          }
        }
        if constexpr (Jv == 4) {
          float m = g == 0 ? 0.f : mrow[J];
#pragma unroll
          for (int q = 0; q < 4; ++q) m = fmaxf(m, fabsf(acc0[EB + J][4 * g + q]));
          mrow[J] = m;
          if constexpr (MODE == 0 && IT + 1 < 8) load_bias_next(IT + 1);
          if constexpr (MODE == 2) {        // the tangent's second output: stored at once (its values are not kept)
            const gptr<float> b_blk = p_side_out2 + tile_off + (((unsigned)(EB + J) * (unsigned)l_ld + (unsigned)fb) * 32u + (unsigned)r_o);
#pragma unroll
            for (int q = 0; q < 4; ++q) b_blk[(8 * g + q) * 32] = fm[q];
          }
        }
        if constexpr (Jv == 5) {
          if constexpr (MODE != 0 && IT + 3 < 8) side_load(IC<IT + 3>{}, TT{}, opt_tag, TT{});
          // forward / tangent: store group (IT - 1) / 2, computed in slot (IT - 1) / 2 -- one group every other slot
          if constexpr (MODE != 1 && (IT & 1) == 1) side_store(IC<((IT - 1) / 2) / 4>{}, IC<((IT - 1) / 2) % 4>{}, TT{}, opt_tag, TT{});
        }
      };
      // slice Jv of split item IT: x s -> hi = f16(x s), lo = f16(x s - hi), both planes in place
      auto split_slice = [&](auto it_tag, auto j_tag, auto opt_tag) {
        constexpr int IT = decltype(it_tag)::value, Jv = decltype(j_tag)::value, J = IT / 4, g = IT % 4;
        if constexpr (Jv == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xs4[q] = mul_nc(acc0[EB + J][4 * g + q], s_row[J]);
        } else if constexpr (Jv == 1) {
          ph4 = __builtin_convertvector(xs4, f16x4);
        } else if constexpr (Jv == 2) {
          bk4 = __builtin_convertvector(ph4, f32x4);
        } else if constexpr (Jv == 3) {
#pragma unroll
          for (int q = 0; q < 4; ++q) xs4[q] = sub_nc(xs4[q], bk4[q]);
        } else if constexpr (Jv == 4) {
          const f16x4 pl4 = __builtin_convertvector(xs4, f16x4);
          const int k = nb * 32 + 4 * hh + 8 * g, R = (EB + J) * 32 + r_o;
          char* pp = actb + ((size_t)((k >> 3) * TMP + R) * 16 + (k & 7) * 2);
          *reinterpret_cast<f16x4*>(pp) = ph4;
          *reinterpret_cast<f16x4*>(pp + (size_t)PLANE * 16) = pl4;
        } else {
          if constexpr (MODE == 1) side_store(IC<J>{}, IC<g>{}, TT{}, opt_tag, TT{});
          else if constexpr ((IT & 1) == 1) side_store(IC<((IT + 8 - 1) / 2) / 4>{}, IC<((IT + 8 - 1) / 2) % 4>{}, TT{}, opt_tag, TT{});
        }
        if constexpr (MODE == 1 && (Jv == 1 || Jv == 2)) {
          // bias gradient: column sums of the half's deltas, one feature per slice (the LDS atomics follow the last slot)
          const int i = 2 * IT + (Jv - 1);
          float c = acc0[EB][i] + acc0[EB + 1][i];
          c += dpp<DPP_XOR1>(c);
          c += dpp<DPP_XOR2>(c);
          c += dpp<DPP_HALF_MIRROR>(c);
          c += dpp<DPP_MIRROR>(c);
          csum[i] = c;
        }
      };
      // FAST: straight-line epilogue items (this wave's block is full, side tensors point-blocked and present, no per-row-group
      // term).  KCOND: the k-step of a slot sits behind a uniform branch (fewer than 16 k-steps -- a first layer --, no K role in
      // the phase, or a wave without a column block in layer liK); otherwise the slot is one basic block and the scheduler places
      // the item's instructions between the MFMAs.
      const bool fast_e = activeE && full && blk && !p_rowbias && p_side_out != nullptr &&
                          (MODE != 2 || p_side_out2 != nullptr) && (MODE == 0 || ((p_bgrad != nullptr) == (MODE == 1)));
      const bool fast_k = activeK && KS >= 16;
      const bool opt = MODE == 1 && p_side_ex != nullptr;
      auto slots = [&](auto fast_tag, auto opt_tag, auto kcond_tag) {
        constexpr bool FAST = decltype(fast_tag)::value, KCOND = decltype(kcond_tag)::value;
        auto slot = [&](auto i_tag) {
          constexpr int I = decltype(i_tag)::value;
          if constexpr (FAST) {
            if constexpr (KCOND) { if (activeK && I < KS) kstep(IC<I % 3>{}, IC<I % 2>{}, I, std::integral_constant<bool, I == 0>{}); }
            // six steps: [MFMA j of the k-step] [the k-step's own loads that may follow it] [slice j of the epilogue item], fenced --
            // the order the matrix pipe wants (one MFMA, ~6 vector instructions, the next MFMA ...) written down instead of
            // asked for: left to `sched_group_barrier` the MFMAs of a slot clustered at its head and the item's dependent
            // chain (fma -> exp -> add -> log -> add -> mul per value) followed them, and a wave that waits on the saturated
            // matrix pipe blocks its SIMD's vector issue (tools/ubench/coissue.hip): matrix and vector time added up again
#ifdef NDJIR_CHAINP_X_NOITEM      // (timing experiment, WRONG results: the fast form without its epilogue items)
#define NDJIR_ITEM(Jv)
#else
#define NDJIR_ITEM(Jv) if constexpr (I < 8) act_slice(IC<I>{}, IC<Jv>{}, opt_tag); else split_slice(IC<I - 8>{}, IC<Jv>{}, opt_tag);
#endif
#define NDJIR_STEP(Jv)                                                                                   \
            if constexpr (!KCOND) { mfma_j(i_tag, IC<Jv>{}); __builtin_amdgcn_sched_barrier(0); k_side(i_tag, IC<Jv>{}); } \
            NDJIR_ITEM(Jv) \
            __builtin_amdgcn_sched_barrier(0);
            NDJIR_STEP(0) NDJIR_STEP(1) NDJIR_STEP(2) NDJIR_STEP(3) NDJIR_STEP(4) NDJIR_STEP(5)
#undef NDJIR_STEP
#undef NDJIR_ITEM
          } else {
            if (activeK && I < KS) kstep(IC<I % 3>{}, IC<I % 2>{}, I, std::integral_constant<bool, I == 0>{});
            if (activeE) {
              if constexpr (I < 8) { if (full) act_item(IC<I>{}, FF{}, FF{}, TT{}); else act_item(IC<I>{}, FF{}, FF{}, FF{}); }
              else { if (full) split_item(IC<I - 8>{}, FF{}, FF{}, TT{}); else split_item(IC<I - 8>{}, FF{}, FF{}, FF{}); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#ifdef NDJIR_CHAINP_SUBSTAMP      // (diagnostic build: per-slot stamps of layer 2, waves 0 and 4 of workgroup 0, behind the regular timeline)
          if (a0.timeline && blockIdx.x == 0 && lane == 0 && stamp_li == 2 && (wave & 3) == 0 && a0.timeline[399] >= 64)
            a0.timeline[400 + (HK * 2 + (wave >> 2)) * 18 + I + 1] = (long long)__builtin_amdgcn_s_memtime();
#endif
        };
#ifdef NDJIR_CHAINP_SUBSTAMP
        if (a0.timeline && blockIdx.x == 0 && lane == 0 && stamp_li == 2 && (wave & 3) == 0 && a0.timeline[399] >= 64)
          a0.timeline[400 + (HK * 2 + (wave >> 2)) * 18] = (long long)__builtin_amdgcn_s_memtime();
#endif
        k_preload();        // (inside the variant: the fragments are born in the block that consumes them)
        if constexpr (FAST && MODE == 0) load_bias_next(0);
        slot(IC<0>{}); slot(IC<1>{}); slot(IC<2>{}); slot(IC<3>{});
        slot(IC<4>{}); slot(IC<5>{}); slot(IC<6>{}); slot(IC<7>{});
        if (activeE) rowmax_finish();
        if (doE) {
          // every wave has contributed its row maxima of the E half
#ifndef NDJIR_CHAINP_X_NOMIDBAR      // (timing experiment, WRONG results: no barrier in front of the split)
          lds_barrier();
#endif
          if (tid < HALF) {
            const int row = HE * HALF + tid;
            float s_r, inv_r;
            scale_from_max(s_rmax[row], s_r, inv_r);
            s_ainv[row] = inv_r;            // read by the epilogue of the next layer on this half, two barriers on
            if (lyE.side_amax) {
              const float wm = wave_max(__uint_as_float(s_rmax[row]));
              if (lane == 0) atomicMax(lyE.side_amax, __float_as_uint(wm));
            }
          }
        }
        stamp(stamp_li, stamp_ph + 1);
        if constexpr (FAST) {
          float inv_row;
          scale_from_max(s_rmax[EB * 32 + r_o], s_row[0], inv_row);
          scale_from_max(s_rmax[(EB + 1) * 32 + r_o], s_row[1], inv_row);
        }
        slot(IC<8>{}); slot(IC<9>{}); slot(IC<10>{}); slot(IC<11>{});
        slot(IC<12>{}); slot(IC<13>{}); slot(IC<14>{}); slot(IC<15>{});
        if (FAST && MODE == 1) {
          if ((lane_o & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(p_bgrad + fb + acc_feat(i, 0), csum[i]);
          }
        }
      };
      auto slots_k = [&](auto fast_tag, auto opt_tag) {
        if (fast_k) slots(fast_tag, opt_tag, FF{}); else slots(fast_tag, opt_tag, TT{});
      };
      if (fast_e) {
        if constexpr (MODE == 1) { if (opt) slots_k(TT{}, TT{}); else slots_k(TT{}, FF{}); }
        else slots_k(TT{}, FF{});
      } else slots(FF{}, FF{}, TT{});
      // k-steps beyond the 16 slots (inputs wider than 256 columns): bare
      if (activeK) {
        for (int ks = 16; ks < KS; ks += 6) {
          kstep(IC<1>{}, IC<0>{}, ks, FF{});
          if (ks + 1 < KS) kstep(IC<2>{}, IC<1>{}, ks + 1, FF{});
          if (ks + 2 < KS) kstep(IC<0>{}, IC<0>{}, ks + 2, FF{});
          if (ks + 3 < KS) kstep(IC<1>{}, IC<1>{}, ks + 3, FF{});
          if (ks + 4 < KS) kstep(IC<2>{}, IC<0>{}, ks + 4, FF{});
          if (ks + 5 < KS) kstep(IC<0>{}, IC<1>{}, ks + 5, FF{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output, rows of the E half ----
      if (MODE != 1 && is_skip) {
        lds_barrier();                   // (the owner of the last, partial column block wrote zeros where the input part begins)
        const int K0 = a.K0, base = l_N;
        const float* X = a.X + (row0 + HE * HALF) * a.ldx;
        for (int t = tid; t < K0 * HALF; t += NTHREADS) {
          const int k = t % K0, m = t / K0;
          const float v = X[(long long)m * a.ldx + k] * a.skip_scale;
          const int kk = base + k;
          float s_r, inv_r;
          scale_from_max(s_rmax[HE * HALF + m], s_r, inv_r);
          put1(kk, HE * HALF + m, v, s_r);
          if (lyE.side_out && !blk) lyE.side_out[(row0 + HE * HALF + m) * lyE.ld_side + kk] = v;
        }
        if (lyE.side_out && blk) {
          float* so = lyE.side_out + row0 * lyE.ld_side;
          for (int t = tid; t < K0 * HALF; t += NTHREADS) {
            const int m = HE * HALF + t % HALF, k = t / HALF;
            so[((unsigned)(m >> 5) * (unsigned)lyE.ld_side + (unsigned)(base + k)) * 32u + (unsigned)(m & 31)] =
                X[(long long)(m - HE * HALF) * a.ldx + k] * a.skip_scale;
          }
        }
        const int wcat = base + K0, wpad = (wcat + 15) & ~15;
        for (int t = tid; t < (wpad - wcat) * HALF; t += NTHREADS)
          put1(wcat + t % (wpad - wcat), HE * HALF + t / (wpad - wcat), 0.f, 1.f);
      }
      lds_barrier();                     // end of the phase: the E half's planes are written, the K half's are read
    };

    int pend = -1;                       // hidden layer whose H1 epilogue is still to run
    for (int li = 0; li < a.L; ++li) {
      const bool last = a.has_output && (li == a.L - 1);
      if (!last) {
        phase(IC<0>{}, li, true, pend, pend >= 0, li, 0);
        phase(IC<1>{}, li, true, li, true, li, 2);
        pend = li;
        stamp(li, 4);
        continue;
      }
      if (pend >= 0) { phase(IC<0>{}, 0, false, pend, true, li, 0); pend = -1; }
      // ================= output layer: un-pipelined (mlp3w.hip's), both halves one after the other =================
      const auto& ly = a.layers[li];
      const int KS = (ly.Kp + 15) >> 4;
      const int NB = ly.Np >> 5;
      const gptr<const f16x8> p_wp = (gptr<const f16x8>)upin(ly.Wp);
      const gptr<const float> p_winv = upin(ly.Wp + (long long)KS * 16 * ly.Np);
      const gptr<const float> p_bias = upin(ly.bias);
      const int l_N = pin(ly.N);
      // plain k-loop of one half: blocks 2 H, 2 H + 1 of column block nbk, k-steps [ks0, ks1)
      auto kloop_half = [&](auto h_tag, const int nbk, const int ks0, const int ks1) {
        constexpr int H = decltype(h_tag)::value, KB = 2 * H;
#pragma unroll
        for (int q = 0; q < 2; ++q) acc0[KB + q] = f32x16{0};
        const int lane_k = fresh_lane();
        const gptr<const f16x8> Bp = p_wp + ((long long)nbk * KS) * 2 * 64 + lane_k;
        const f16x8* A0 = act + (lane_k >> 5) * TMP + KB * 32 + (lane_k & 31);
        f16x8 b[3][2], af[2][2][2];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int p = 0; p < 2; ++p) b[s][p] = Bp[(long long)((ks0 + s < ks1 ? ks0 + s : ks0) * 2 + p) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const f16x8* An = A0 + 2 * ks0 * TMP;
#pragma unroll
          for (int q = 0; q < 2; ++q) { af[0][0][q] = An[q * 32]; af[0][1][q] = An[PLANE + q * 32]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto stag, auto ctag, const int ks) {
          constexpr int S = decltype(stag)::value, C = decltype(ctag)::value;
          const f16x8* An = A0 + 2 * (ks + 1) * TMP;
          if (ks + 1 < ks1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { af[C ^ 1][0][q] = An[q * 32]; af[C ^ 1][1][q] = An[PLANE + q * 32]; }
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][1][q], acc0[KB + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0], af[C][0][q], acc0[KB + q], 0, 0, 0);
          const f16x8 blo = b[S][1] * LOU;
#pragma unroll
          for (int q = 0; q < 2; ++q) acc0[KB + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, af[C][0][q], acc0[KB + q], 0, 0, 0);
          if (ks + 3 < ks1) {
#pragma unroll
            for (int p = 0; p < 2; ++p) b[S][p] = Bp[(long long)((ks + 3) * 2 + p) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        for (int ks = ks0; ks < ks1; ks += 6) {
          kstep(IC<0>{}, IC<0>{}, ks);
          if (ks + 1 < ks1) kstep(IC<1>{}, IC<1>{}, ks + 1);
          if (ks + 2 < ks1) kstep(IC<2>{}, IC<0>{}, ks + 2);
          if (ks + 3 < ks1) kstep(IC<0>{}, IC<1>{}, ks + 3);
          if (ks + 4 < ks1) kstep(IC<1>{}, IC<0>{}, ks + 4);
          if (ks + 5 < ks1) kstep(IC<2>{}, IC<1>{}, ks + 5);
        }
      };
      stamp(li, 1);
      const bool narrow = NB == 1;
      if (!narrow) {
        const int nrounds = (NB + NWAVES - 1) / NWAVES;
#pragma unroll 1
        for (int round = 0; round < nrounds; ++round) {
          const int nb = NWAVES * round + wave;
          if (nb >= NB) continue;
          kloop_half(IC<0>{}, nb, 0, KS);
          kloop_half(IC<1>{}, nb, 0, KS);
          // ---- z = acc / scales (+ bias) -> Y ----
          const int lane_o = fresh_lane();
          const int r_o = lane_o & 31, hh = lane_o >> 5;
          const int fb = nb * 32 + 4 * hh;
          const float winv = p_winv[nb];
          const bool vec_y = (nb * 32 + 31 < l_N);
          f32x4 bias4[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) bias4[g][q] = (MODE == 0 && p_bias && fb + 8 * g + q < l_N) ? p_bias[fb + 8 * g + q] : 0.f;
#pragma unroll
          for (int J = 0; J < 4; ++J) {
            const int R = J * 32 + r_o;
            const float sa = s_ainv[R];
            float* y = a.Y + (row0 + R) * a.ldy + fb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 t;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float z = out_z(acc0[J][4 * g + q], sa, winv);
                t[q] = MODE == 0 ? out_add(z, bias4[g][q]) : z;
              }
              if (vec_y) {
                if (a.accum_y) {
                  const f32x4 y0 = *reinterpret_cast<const f32x4u*>(y + 8 * g);
#pragma unroll
                  for (int q = 0; q < 4; ++q) t[q] = out_add(t[q], y0[q]);
                }
                *reinterpret_cast<f32x4u*>(y + 8 * g) = t;
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (fb + 8 * g + q < l_N) y[8 * g + q] = a.accum_y ? out_add(t[q], y[8 * g + q]) : t[q];
              }
            }
          }
        }
        stamp(li, 2);
        __syncthreads();
        stamp(li, 4);
      } else {
        // narrow output layer (N <= 32): K in four quarters, one per wave 0 .. 3, summed in mlp3.hip's fixed order
        const int kq = wave & 3;
        const bool active = wave < 4;
        const int ks0 = (KS * kq) / 4, ks1 = (KS * (kq + 1)) / 4;
        if (active) { kloop_half(IC<0>{}, 0, ks0, ks1); kloop_half(IC<1>{}, 0, ks0, ks1); }
        __syncthreads();
        float* part = reinterpret_cast<float*>(actb);     // [kq][m][n]: 4 x TM x 32 floats <= the planes
        if (active) {
          const int lane_o = fresh_lane();
          const int r_o = lane_o & 31, hh = lane_o >> 5;
#pragma unroll
          for (int J = 0; J < 4; ++J) {
            float* dst = part + ((size_t)(kq * TM + J * 32 + r_o)) * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v;
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = acc0[J][4 * g + q];
              *reinterpret_cast<f32x4*>(dst + 8 * g) = v;
            }
          }
        }
        __syncthreads();
        const float winv = p_winv[0];
        for (int t = tid; t < TM * 32; t += NTHREADS) {
          const int n = t & 31, m = t >> 5;
          float z = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) z += part[q * TM * 32 + t];
          z = out_z(z, s_ainv[m], winv);
          if (n < l_N) {
            if (MODE == 0) z = out_add(z, p_bias ? p_bias[n] : 0.f);
            float* y = a.Y + (row0 + m) * a.ldy + n;
            *y = a.accum_y ? out_add(z, *y) : z;
          }
        }
        __syncthreads();
        stamp(li, 4);
      }
    }
    if (pend >= 0) phase(IC<0>{}, 0, false, pend, true, a.L - 1, 0);      // (a chain that ends with a hidden layer)
    if (tile == blockIdx.x) stamp(MAX_CHAIN_LAYERS - 1, 2);
    xpar ^= 1;
   }   // nets
  }
  stamp(MAX_CHAIN_LAYERS - 1, 3);
  stamp_rt(1);
  if (MODE != 0) {
    __syncthreads();
    for (int ni = 0; ni < n_nets; ++ni) {
      const auto& a = nets.get(ni);
      const float* bs = lds + a.bg_lds;
      float* part = a.bg_partial + (long long)blockIdx.x * a.bg_total;
      for (int i = tid; i < a.bg_total; i += NTHREADS) part[i] = bs[i];
    }
  }
}

template <int MODE>
__global__ void __launch_bounds__(NTHREADS, 2) k_chainp(ChainArgs a) {
  chainp_body<MODE>(OneNet{a});
}
template <int MODE>
__global__ void __launch_bounds__(NTHREADS, 2) k_chainp_nets(ChainGroup /* read in the kernel-argument segment */) {
  chainp_body<MODE>(ManyNets{(const KGroup*)__builtin_amdgcn_kernarg_segment_ptr()});
}

}  // namespace x3p

#ifndef NDJIR_NO_LAUNCHER
// Launched with the plan launch_chainw_group (mlp3w.hip) made for its <mode, 4, 8, 128> kernel: same planes, same bias sums,
// same grid (the deferred bias partials' layout), same argument blocks.
int launch_chainp_group(const ChainGroup& grp, int mode, int blocks, size_t lds_bytes, hipStream_t stream) {
  using namespace x3p;
  static bool attr_set = false;
  if (!attr_set) {
    constexpr int LDS_DYN_MAX = 160 * 1024 - 4096;
#define NDJIR_SETP(M) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainp<M>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX); \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainp_nets<M>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX)
    NDJIR_SETP(0); NDJIR_SETP(1); NDJIR_SETP(2);
#undef NDJIR_SETP
    attr_set = true;
  }
#define NDJIR_GOP(M)                                                                                                      \
  do {                                                                                                                    \
    if (grp.n == 1) hipLaunchKernelGGL((k_chainp<M>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, grp.net[0]); \
    else hipLaunchKernelGGL((k_chainp_nets<M>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, grp);              \
  } while (0)
  if (mode == 0) NDJIR_GOP(0); else if (mode == 1) NDJIR_GOP(1); else NDJIR_GOP(2);
#undef NDJIR_GOP
  return NDJIR_OK;
}
#endif

}  // namespace ndjir
