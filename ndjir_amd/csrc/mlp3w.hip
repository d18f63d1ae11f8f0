// mlp3w.hip -- the f16x3 chain (see mlp3.hip for the arithmetic) on 128-point tiles, epilogue in the accumulator registers.
//
// Why a second kernel.  In mlp3.hip a weight fragment fetched from L2 feeds the two row blocks of a 64-point tile: a
// 256-wide layer streams 256 KB of packed weights per tile, 27 B / clk / CU achieved against the ~42 B / clk the matrix pipe
// could consume -- the k-loops wait for the L2 -> CU stream, not for the MFMAs (measured: k-loop of the slower wave of a SIMD
// 9.7 k cycles per layer against a 6.1 k matrix floor, with or without the activation stores).  Twice the rows per weight
// fetch need a 128-point tile, whose two f16 planes (135 KB at 256 columns) leave no room for the 67 KB of fp32 staging
// tiles through which mlp3.hip transposes its accumulators.  So this kernel does not transpose:
//   * the MFMA operands are swapped -- A = weight fragment (rows = output features), B = activation fragment (columns =
//     points) -- so that a lane's 16 accumulator values are 4 x 4 consecutive FEATURES of ONE point.  Bias, activation,
//     softplus' products, the row maximum (a lane-local max + one LDS atomic), the 2-way split and the plane writes
//     (ds_write_b64, conflict-free) all happen on the accumulator registers; side loads / stores are 16 bytes per lane
//     (32 contiguous bytes per point and instruction);
//   * a wave owns ONE column block and up to FOUR row blocks of it (128 accumulator registers): 12 MFMAs per pair of
//     weight fragments instead of 6;
//   * the results wait in the accumulator registers for the row-maximum barrier; no staging memory exists.
// The packed weights, the plane layout, the scaling groups and every forward expression (mlp3_util.h) are those of
// mlp3.hip: a point's forward result is bit-identical whichever kernel evaluates it (tests/test_gpu_mlp.py).
// Launched for P % 128 == 0, P >= 32768, hidden layers of at most 8 column blocks; everything else stays with mlp3.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mlp.h"
#include "mlp3_util.h"

namespace ndjir {
namespace x3w {

using namespace x3u;

typedef ChainGroup __attribute__((address_space(4))) KGroup;     // (the kernel-argument segment is constant address space)
typedef ChainArgs __attribute__((address_space(4))) KArgs;
typedef ChainLayer __attribute__((address_space(4))) KLayer;

constexpr int TM_MAX = 128;      // points per tile: 128 (one 8- or two 4-wave workgroups per CU) or 64 (TWO workgroups per CU, see below)
constexpr int tile_pad(int tm) { return tm + 4; }      // rows per k-group incl. pad: (TMP * 16) % 256 == 64 -> conflict-free plane writes
constexpr int IN_CACHE = 10;     // float4 groups of the chain input a thread keeps between the max pass and the split

// accumulator register i of lane (r = lane & 31, hh = lane >> 5): point r of the row block, feature acc_feat(i, hh) of the
// column block -- four groups (i >> 2) of four consecutive features
__device__ __forceinline__ int acc_feat(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// RPW: row blocks per wave -- 4 (hidden layers of up to 8 column blocks: one wave per column block) or 2 (up to 4 column
// blocks: two waves per column block).  One k-loop instantiation per kernel: the register allocator sees one hot loop.
// NWAVES: 8 (one workgroup per CU) or 4 (RPW = 4, hidden layers of up to 4 column blocks: TWO workgroups per CU, one wave
// of each per SIMD -- the epilogue of one runs beside the k-loop of the other).
// The launch carries up to MAX_GROUP_NETS nets on the same points (ChainGroup): a workgroup takes its tile through net 0, net 1, ...
// in turn -- each net stages the tile's input again (L2-resident after the first) and, in the backward, the first net assigns
// the input gradient's tile and the others add to it while it is still in L2.  One net = the plain launch.
// NETS: where the argument blocks are -- OneNet: the kernel's by-value ChainArgs (the plain launch: the net loop folds away and
// the compiler reads the block into scalar registers once, as it always did); ManyNets: a ChainGroup, indexed by the net counter
// where it lies in the kernel-argument segment (uniform scalar loads; indexing a by-value copy would put all 3.4 KB in scratch).
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
// TM = 64, CPW = 2 (round 5): a 4-wave workgroup on a 64-point tile, a wave owns TWO column blocks (nb, nb + 4) x both row
// blocks -- the same 128 accumulator registers, 12 MFMAs per k-step.  Its planes (<= 74 KB) fit the LDS twice, so a CU holds TWO
// independent workgroups, one wave of each per SIMD: while one is in its k-loop (matrix pipe) the other is in its epilogue
// (vector pipe, memory) -- inside ONE workgroup all waves share a barrier schedule and the two phases can only add
// (a 256-wide layer of a 128-point tile took 32 k cycles forward / 38 k backward against 12.3 k of matrix time,
// on an idle chip as on a full one).  The price: a weight fragment feeds 2 row blocks instead of 4 (the L2 -> CU stream doubles).
// TEAMS = 2 (round 5, TM = 64, CPW = 2, NWAVES = 4 waves PER TEAM): ONE 8-wave workgroup holds two independent 64-point tiles, one
// per team of four waves (one wave of each team on every SIMD; each team has its own planes, row maxima and scales).  Both
// teams run the same program -- everything below is written per team: `tid`, `wave`, NTHREADS, NWAVES are team-local -- but team
// 1 runs it ONE BARRIER BEHIND team 0 (it passes one extra barrier before its first tile, team 0 one after its last), and a hidden
// layer is cut into three barrier intervals: k-loop | activation epilogue | split.  A workgroup barrier releases when every wave
// has arrived at A barrier, so the n-th barrier of team 0 pairs with the (n - 1)-th of team 1 and the intervals pair up as
// (k-loop, split), (epilogue, k-loop), (split, epilogue): on every SIMD one wave multiplies while the other runs vector
// instructions -- the overlap two free-running workgroups per CU did not find (they fell into step).  Weight fragments feed
// 2 row blocks instead of 4, as in the 64-point-tile kernel above.
template <int MODE, int RPW, int NWAVES, int TM, int CPW, int TEAMS, class NETS>
__device__ __forceinline__ void chainw_body(const NETS nets) {
  constexpr int TMP = tile_pad(TM);
  constexpr int NTHREADS = NWAVES * 64;      // (per team)
  constexpr bool BWD = (MODE == 1);
  constexpr int G = (TM / 32) / RPW;   // wave groups sharing a column block's rows
  constexpr int CB = NWAVES / G;       // column blocks per round (x CPW for a hidden layer)
  constexpr int NBLK = RPW * CPW;      // 32 x 32 output blocks (accumulator pairs) of a wave
  static_assert(NBLK <= 4 && (CPW == 1 || (RPW == 2 && G == 1)), "accumulator budget");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned s_rmax_t[TEAMS][2][TM];   // per row: largest finite |output| of the layer (bit pattern; ping-pong by layer)
  __shared__ unsigned s_xmax_t[TEAMS][2][TM];   // per row: largest finite |x| of the chain input tile (ping-pong by tile)
  __shared__ float s_ainv_t[TEAMS][TM];         // per row: 1 / scale of the planes the next k-loop reads
  const int n_nets = nets.n();
  const auto& a0 = nets.get(0);
  const int PLANE = a0.lds_split;      // 16-byte units per plane ( = k-groups * TMP ); one value for the whole group
  const int gwave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int team = TEAMS == 1 ? 0 : gwave / NWAVES;
  unsigned (&s_rmax)[2][TM] = s_rmax_t[team];
  unsigned (&s_xmax)[2][TM] = s_xmax_t[team];
  float (&s_ainv)[TM] = s_ainv_t[team];
  f16x8* act = reinterpret_cast<f16x8*>(lds) + (size_t)team * 2 * PLANE;      // (a team's two planes)
  char* actb = reinterpret_cast<char*>(act);
  const int tid = (int)threadIdx.x - team * NTHREADS;
  const int lane = tid & 63;
  const int wave = gwave - team * NWAVES;
  // an opaque copy of the lane id: addresses derived from it are formed where they are used instead of being hoisted out of
  // the tile / layer loops into registers that stay occupied (or spilled) through the k-loops
  auto fresh_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
  auto stamp = [&](int li, int phase) {
    if (a0.timeline && blockIdx.x == 0 && lane == 0) a0.timeline[(li * 5 + phase) * 8 + gwave] = (long long)__builtin_amdgcn_s_memtime();
  };
  auto stamp_rt = [&](int phase) {
    if (a0.timeline && blockIdx.x == 0 && lane == 0) a0.timeline[((MAX_CHAIN_LAYERS - 2) * 5 + phase) * 8 + gwave] = (long long)__builtin_amdgcn_s_memrealtime();
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

  stamp(MAX_CHAIN_LAYERS - 1, 0);
  stamp_rt(0);
  if constexpr (TM == 64 && TEAMS == 1) {
    // Two workgroups share a CU (a grid of 512 fills the chip exactly, 256 at a time: workgroups b and b + 256 are the pair of
    // a CU, the launcher caps the grid there).  With equal priority the two fall into step -- both in their k-loops at half
    // speed, then both in their epilogues -- and nothing overlaps.  One of them always wins the arbitration instead: it runs
    // as if alone, the other takes the pipe the first is not using (matrix while it is in its epilogue, vector while it
    // multiplies), which puts them in anti-phase by itself.
    if (a0.t64_prio && (blockIdx.x & 256) == 0) __builtin_amdgcn_s_setprio(3);
  }
#ifndef NDJIR_NO_L2_WARMUP
  // L2 warm-up.  A launch finds its net's packed weights cold (the previous launch streamed another net's weights and hundreds
  // of MB of activations through the 4 MB L2 of each XCD), and a layer's first k-steps on every CU of an XCD then miss together:
  // one memory latency per LAYER, 8 in a row for the geometric net (the fixed-cost fit: 25 - 80 us of intercept per backward /
  // tangent launch).  Here the workgroups of an XCD's first round (blockIdx & 7 = XCD, blockIdx >> 3 = its index there) each
  // touch 1 / 32 of EVERY layer's weights, one dword per 128-byte line, before the first tile's input stage: all layers'
  // misses overlap once.  The values are consumed (an empty asm) before the first barrier.
  unsigned warm = 0;
  if (blockIdx.x < 256) {
    const int part = (blockIdx.x >> 3) & 31;
    for (int ni = 0; ni < n_nets; ++ni) {
      const auto& a = nets.get(ni);
      for (int li = 0; li < a.L; ++li) {
        const auto& ly = a.layers[li];
        const long long lines = (((long long)((ly.Kp + 15) >> 4) * 16 * ly.Np + (ly.Np >> 5)) * 4 + 127) >> 7;      // 128-byte lines of the packed matrix
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
  if (tid < 2 * TM) { (&s_rmax[0][0])[tid] = 0u; (&s_xmax[0][0])[tid] = 0u; }
  __syncthreads();
  if (TEAMS == 2 && team == 1) __syncthreads();      // team 1 runs one barrier behind team 0 from here on
  int xpar = 0;                        // ping-pong slot of the input row maxima
  bool warm_pending = true;

  for (long long tile = blockIdx.x; tile < a0.n_tiles; tile += gridDim.x) {
    const long long row0 = (tile * TEAMS + team) * TM;  // (the launcher guarantees P % (TM * TEAMS) == 0: every tile is full)
   for (int ni = 0; ni < n_nets; ++ni) {
    const auto& a = nets.get(ni);
    float* const bsum = lds + a.bg_lds;
    const float beta = a.beta;

    // ---- chain input tile -> planes (zero padded to a multiple of 16 features) ----
    {
      const int K0p = a.K0p, K0 = a.K0;
      const float* X = a.X + row0 * a.ldx;
      int groups = K0p >> 2;
      asm volatile("" : "+s"(groups));      // (keeps the per-thread addresses below out of the tile loop's preheader)
      const int total = groups * TM;
      auto load = [&](int t) -> f32x4 {
        const int g = t % groups, m = t / groups;
        const int k = g * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k + 3 < K0) v = *reinterpret_cast<const f32x4u*>(X + (long long)m * a.ldx + k);     // (any row stride / base offset)
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
        s_xmax[xpar ^ 1][tid] = 0u;                    // the other slot: next tile's input stage, many barriers away
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
        if (MODE != 0 && a.in_bgrad) {      // bias gradient of the output layer: column sums of the input
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

    int cur = 0;                         // ping-pong slot of the layer's row maxima
    for (int li = 0; li < a.L; ++li) {
      const auto& ly = a.layers[li];
      const int KS = (ly.Kp + 15) >> 4;          // k-steps of 16 (planes are zero beyond Kp)
      const int NB = ly.Np >> 5;
      const bool last = a.has_output && (li == a.L - 1);
      stamp(li, 0);
      const gptr<const f16x8> p_wp = (gptr<const f16x8>)pin(ly.Wp);
      const gptr<const float> p_winv = pin(ly.Wp + (long long)KS * 16 * ly.Np);   // [NB] behind the planes
      const gptr<const float> p_bias = pin(ly.bias);
      const gptr<const float> p_rowbias = pin((MODE == 0 && li == 0) ? a.row_bias : nullptr);
      const int rb_div = pin(a.row_bias_div > 0 ? a.row_bias_div : 1);
      const gptr<const float> p_side_in = pin(ly.side_in);
      const gptr<const float> p_side_ex = pin(MODE == 1 ? ly.side_add : ly.side_in2);
      const gptr<float> p_side_out = pin(ly.side_out);
      const gptr<float> p_side_out2 = pin(ly.side_out2);
      float* const p_bgrad = (MODE != 0 && ly.bgrad) ? bsum + pin(ly.bg_off) : nullptr;
      const int l_N = pin(ly.N);
      const int l_ld = pin(ly.ld_side);
      const bool is_skip = (li == a.skip_layer);
      const float sc = pin(is_skip ? a.skip_scale : 1.f);
      const int nlim = pin((BWD && is_skip) ? a.skip_split : l_N);
      const float b2 = beta * LOG2E, ib2sc = LN2 / beta * sc;
      const float hsc = (MODE != 0 && is_skip) ? 1.f / sc : 1.f;
      const float nb2 = -b2 * hsc;

      // Accumulator pairs of up to four 32 x 32 blocks (static indices only).
      f32x16 acc0[4], acc1[4];
      // One k-loop: RPW row blocks rb0.. of the wave's CPW column blocks nbk[0 .. CPW), k-steps [ks0, ks1), into accumulator
      // pairs c * RPW + q.  Weight fragments (the MFMA's A operand here) 3 steps ahead in rotating static slots; activation
      // fragments (B operand) per unit of two row blocks: with two units (RPW = 4) a unit's registers are reloaded for the
      // next k-step as soon as its six MFMAs have issued, i.e. one unit = 6 MFMAs ahead of their use; with one unit they are
      // double buffered (and, CPW = 2, feed both column blocks: 12 MFMAs per pair of activation fragments).
      auto kloop = [&](const int (&nbk)[CPW], const int rb0, const int ks0, const int ks1) {
        constexpr int U = 2;                       // row blocks per unit
        constexpr int NU = RPW / U;                // units per k-step
        static_assert(NU == 1 || CPW == 1, "two units only with one column block");
#pragma unroll
        for (int q = 0; q < NBLK; ++q) { acc0[q] = f32x16{0}; acc1[q] = f32x16{0}; }
        // A wave in its k-loop outranks a wave in its epilogue (round 5): the two waves of a SIMD leave their k-loops 5 k cycles
        // apart -- the older one wins the matrix pipe while both multiply -- and the younger one's last 40 MFMAs then took
        // those 5 k cycles: the older wave's epilogue, a dense vector stream, won every issue slot.  An MFMA costs its
        // neighbour one slot in eight.
#ifndef NDJIR_NO_KLOOP_PRIO
        __builtin_amdgcn_s_setprio(2);
#endif
        const int lane_k = fresh_lane();
        gptr<const f16x8> Bp[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) Bp[c] = p_wp + ((long long)nbk[c] * KS) * 2 * 64 + lane_k;
        const f16x8* A0 = act + (lane_k >> 5) * TMP + rb0 * 32 + (lane_k & 31);
        f16x8 b[3][CPW][2];            // [slot][column block][plane]
        f16x8 af[2][2][U];             // NU == 2: [unit][plane][row block of the unit]; NU == 1: [buffer][plane][row block]
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[s][c][p] = Bp[c][(long long)((ks0 + s < ks1 ? ks0 + s : ks0) * 2 + p) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const f16x8* An = A0 + 2 * ks0 * TMP;
#pragma unroll
          for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int q = 0; q < U; ++q) { af[u][0][q] = An[(u * U + q) * 32]; af[u][1][q] = An[PLANE + (u * U + q) * 32]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto kstep = [&](auto stag, auto btag, auto gtag, const int ks) {
          constexpr int S = decltype(stag)::value;       // weight slot
          constexpr int C = decltype(btag)::value;       // activation buffer (NU == 1)
          constexpr bool GUARD = decltype(gtag)::value;
          const bool nxt = !GUARD || ks + 1 < ks1;
          const f16x8* An = A0 + 2 * (ks + 1) * TMP;
          if (NU == 1) {
            if (nxt) {
#pragma unroll
              for (int q = 0; q < U; ++q) { af[C ^ 1][0][q] = An[q * 32]; af[C ^ 1][1][q] = An[PLANE + q * 32]; }
            }
            // three partial products (operands swapped: A = weights): w_hi x_lo, w_lo x_hi -> acc1; w_hi x_hi -> acc0
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
#pragma unroll
              for (int q = 0; q < U; ++q) acc1[c * RPW + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][c][0], af[C][1][q], acc1[c * RPW + q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < U; ++q) acc0[c * RPW + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][c][0], af[C][0][q], acc0[c * RPW + q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < U; ++q) acc1[c * RPW + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][c][1], af[C][0][q], acc1[c * RPW + q], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#pragma unroll
              for (int q = 0; q < U; ++q) acc1[u * U + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0][0], af[u][1][q], acc1[u * U + q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < U; ++q) acc0[u * U + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0][0], af[u][0][q], acc0[u * U + q], 0, 0, 0);
#pragma unroll
              for (int q = 0; q < U; ++q) acc1[u * U + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[S][0][1], af[u][0][q], acc1[u * U + q], 0, 0, 0);
#ifdef NDJIR_KLOOP_HALF_LDS      // (timing experiment, WRONG results: only the first unit's fragments are re-read -- half of the k-loop's LDS reads)
              if (nxt && u == 0) {
#else
              if (nxt) {
#endif
#pragma unroll
                for (int q = 0; q < U; ++q) { af[u][0][q] = An[(u * U + q) * 32]; af[u][1][q] = An[PLANE + (u * U + q) * 32]; }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (!GUARD || ks + 3 < ks1) {
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
              for (int p = 0; p < 2; ++p) b[S][c][p] = Bp[c][(long long)((ks + 3) * 2 + p) * 64];
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
          // the weight slots rotate with period 3, the activation buffers (NU == 1) with period 2: unroll by 6
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
#ifndef NDJIR_NO_KLOOP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      using TT = std::true_type;
      using FF = std::false_type;

      // ---- general layer: wave -> (column block nb, RPW consecutive row blocks from rb0) ----
      // RPW = 4: one wave per column block with all 4 row blocks (hidden layers of up to 8 column blocks); RPW = 2: two waves
      // per column block with 2 row blocks each (up to 4 column blocks); an output layer wider than that takes rounds.
      // A narrow output layer (N <= 32, NB = 1) splits K instead: four quarters, one per wave (and half of the rows when
      // RPW = 2), each accumulated from zero; the quarters meet in the plane memory -- free once every wave is through its
      // k-loop -- and are summed in a fixed order: the order of mlp3.hip's split-K path, whatever the tile height.
      {
        const bool narrow = NB == 1;
        const int nrounds = (NB + CB * CPW - 1) / (CB * CPW);       // (> 1 only for an output layer)
        const int wc = narrow ? 0 : wave % CB;        // column of the round
        const int wg = narrow ? wave >> 2 : wave / CB;   // row group
        const int kq = wave & 3;
        const int ks0 = narrow ? (KS * kq) / 4 : 0, ks1 = narrow ? (KS * (kq + 1)) / 4 : KS;
        int nbc[CPW];                                 // the wave's column blocks of the round: wc, wc + CB
        bool actc[CPW];                               // ... that exist (a missing one is computed as a copy of the first, and ignored)
        const int rb0 = wg * RPW;
        bool active = false;
#pragma unroll 1
        for (int round = 0; round < nrounds; ++round) {
          int nbk[CPW];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            nbc[c] = CB * CPW * round + wc + c * CB;
            actc[c] = nbc[c] < NB && wg < G && !(narrow && c > 0);
            nbk[c] = actc[c] ? nbc[c] : nbc[0];
          }
          active = actc[0];
          if (active) kloop(nbk, rb0, ks0, ks1);        // (the kernel's only k-loop instantiation)
          if (round == 0) stamp(li, 1);
          if (TEAMS == 2 && !last && !narrow) __syncthreads();      // a hidden layer's first interval ends here (see TEAMS above)
          if (narrow) break;
          if (!last || !active) continue;
          // ---- output layer: z = acc / scales (+ bias) -> Y ----
          const int lane_o = fresh_lane();
          const int r_o = lane_o & 31, hh = lane_o >> 5;
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            if (!actc[c]) continue;
            const int nb = nbc[c];
            const int fb = nb * 32 + 4 * hh;
            const float winv = p_winv[nb];
            // (16-byte stores at whatever alignment the rows have: the geometric net's output lives at Z + 2 floats, the packed
            // first-order pass's at Z + 3, ndjir_amd/geometric.py / mlp.py)
            const bool vec_y = (nb * 32 + 31 < l_N);
            f32x4 bias4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int q = 0; q < 4; ++q) bias4[g][q] = (MODE == 0 && p_bias && fb + 8 * g + q < l_N) ? p_bias[fb + 8 * g + q] : 0.f;
#pragma unroll
            for (int J = 0; J < RPW; ++J) {
              const int R = (rb0 + J) * 32 + r_o;
              const float sa = s_ainv[R];
              float* y = a.Y + (row0 + R) * a.ldy + fb;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float z = out_z(acc_sum(acc0[c * RPW + J][4 * g + q], acc1[c * RPW + J][4 * g + q]), sa, winv);
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
        }   // rounds
        if (narrow) {
          __syncthreads();
          float* part = reinterpret_cast<float*>(actb);     // [kq][m][n]: 4 x TM x 32 floats <= the (team's) planes
          if (active) {
            const int lane_o = fresh_lane();
            const int r_o = lane_o & 31, hh = lane_o >> 5;
#pragma unroll
            for (int J = 0; J < RPW; ++J) {
              float* dst = part + ((size_t)(kq * TM + (rb0 + J) * 32 + r_o)) * 32 + 4 * hh;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc_sum(acc0[J][4 * g + q], acc1[J][4 * g + q]);
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
              if (last) {      // (a narrow layer is always an output layer: chain_impl refuses it elsewhere)
                float* y = a.Y + (row0 + m) * a.ldy + n;
                *y = a.accum_y ? out_add(z, *y) : z;
              }
            }
          }
          __syncthreads();
          continue;
        }
        if (last) {              // the output layer leaves nothing in the planes: no maximum, no split
          stamp(li, 2);
          __syncthreads();
          stamp(li, 4);
          continue;
        }

        // ================= phase A (hidden layer): activation math on the accumulator registers =================
        if (active) {
          // lane-derived values: (re)assigned from a fresh opaque lane id at the top of every variant of `run` below, so that
          // each variant's copies live inside its own code region (a copy shared by all variants is spilled once ANY of them
          // runs out of registers -- and a scratch reload waits for `vmcnt(0)`, i.e. for every side load in flight)
          // Block B = c * RPW + q of the wave: column block nbc[c], row block rb0 + q.
          int lane_o, r_o, hh, fbc[CPW];                    // fbc: first feature of register group 0 of the column block
          auto lane_values = [&]() {
            lane_o = fresh_lane(); r_o = lane_o & 31; hh = lane_o >> 5;
#pragma unroll
            for (int c = 0; c < CPW; ++c) fbc[c] = nbc[c] * 32 + 4 * hh;
          };
          // Prologue loads (column-block scales, biases, the first blocks' side tensors) are ISSUED TOGETHER at the top of
          // `run` and waited for once, behind the 64 accumulator sums: one round trip instead of five in a row (round 5: a
          // wave spent 5 - 7 k cycles between the end of its k-loop and its first block -- `winv` was pinned to a scalar
          // register by a load + vmcnt(0) + readfirstlane, then each of the four bias vectors was loaded, waited for with
          // vmcnt(0) and multiplied, with a scratch reload of the multiplier in between)
          float winv_raw[CPW], winvc[CPW];                  // winvc: scalar registers (uniform)
          bool full = (l_ld & 3) == 0 && (!p_rowbias || (l_N & 3) == 0), full_cols = true;
#pragma unroll
          for (int c = 0; c < CPW; ++c)
            if (actc[c] && !(nbc[c] * 32 + 31 < nlim)) full_cols = false;
          auto load_winv = [&]() {
#pragma unroll
            for (int c = 0; c < CPW; ++c) winv_raw[c] = p_winv[actc[c] ? nbc[c] : nbc[0]];
          };
          auto pin_winv = [&]() {
#pragma unroll
            for (int c = 0; c < CPW; ++c) winvc[c] = pin(winv_raw[c]);
          };
          full = full && full_cols;
          const bool two = CPW == 1 || actc[CPW - 1];       // (the second column block exists: blocks RPW .. are live)
          const long long tile_off = row0 * l_ld;           // uniform base + 32-bit lane offset: one address register per access
          // the accumulator pairs become single values at once: acc1 is dead from then on.  Issued AFTER the first two blocks'
          // side loads (their latency covers the 64 multiply-adds) and fenced: left to itself the scheduler sinks every sum
          // to its use, acc1 stays live through the whole epilogue, and the lane's addresses go to scratch -- whose reloads
          // (`s_waitcnt vmcnt(0)`: scratch and global loads share one in-order counter) then drain every side load in flight
          auto sum_accs = [&]() {
#pragma unroll
            for (int J = 0; J < NBLK; ++J)
#pragma unroll
              for (int i = 0; i < 16; ++i) acc0[J][i] = acc_sum(acc0[J][i], acc1[J][i]);
            __builtin_amdgcn_sched_barrier(0);
          };
          f32x4 bbv[CPW][4];                                // forward: bias * beta log2(e)
          auto load_bias = [&](auto ft) {                   // (the loads; `scale_bias` multiplies once they are in)
            constexpr bool FULL = decltype(ft)::value;
            if (MODE == 0) {
#pragma unroll
              for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  const int fb = fbc[c];
                  bbv[c][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                  if (FULL && (l_ld & 3) == 0) { if (p_bias) bbv[c][g] = *((gptr<const f32x4>)(p_bias + (unsigned)(fb + 8 * g))); }
                  else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bbv[c][g][q] = (p_bias && fb + 8 * g + q < l_N) ? p_bias[fb + 8 * g + q] : 0.f;
                  }
                }
            }
          };
          auto scale_bias = [&]() {
            if (MODE == 0) {
#pragma unroll
              for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                  for (int q = 0; q < 4; ++q) bbv[c][g][q] = bbv[c][g][q] * b2;
            }
          };
          f32x4 hs[2][4], ex[2][4];                         // backward / tangent: side loads, one block ahead
          // side loads of block J (backward: stored activation + extra adjoint; tangent: stored activation + s)
          auto side_loads = [&](auto jt, auto ft, auto et, auto lt) {
            constexpr int J = decltype(jt)::value;
            constexpr bool FULL = decltype(ft)::value;
            constexpr bool HAS_EX = decltype(et)::value;
            constexpr bool BLK = decltype(lt)::value;
            if (BLK) {
              // point-blocked tensors: register (g, q) of the 32 lanes of a half-wave = 32 consecutive points of feature
              // fbc[J / RPW] + 8 g + q = one 128-byte line; one address register, the feature in the instruction's offset field
              const unsigned boff = ((unsigned)(rb0 + J % RPW) * (unsigned)l_ld + (unsigned)fbc[J / RPW]) * 32u + (unsigned)r_o;
              const gptr<const float> b_in = p_side_in + tile_off + boff;
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) hs[J & 1][g][q] = (FULL || fbc[J / RPW] + 8 * g + q < nlim) ? b_in[(8 * g + q) * 32] : 0.f;
              if (HAS_EX) {
                const gptr<const float> b_ex = p_side_ex + tile_off + boff;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                  for (int q = 0; q < 4; ++q) ex[J & 1][g][q] = (FULL || fbc[J / RPW] + 8 * g + q < nlim) ? b_ex[(8 * g + q) * 32] : 0.f;
              }
              return;
            }
            const unsigned rowoff = (unsigned)((rb0 + J % RPW) * 32 + r_o) * (unsigned)l_ld + (unsigned)fbc[J / RPW];
            // full rows: instruction g reads chunk (lane & 3) of point (r & ~3) + g; hidden_block transposes the quads back
            const unsigned rowoff_t = (unsigned)((rb0 + J % RPW) * 32 + (r_o & ~3)) * (unsigned)l_ld + (unsigned)(nbc[J / RPW] * 32 + 8 * (r_o & 3) + 4 * hh);
            const gptr<const float> b_in = p_side_in + tile_off;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (FULL) hs[J & 1][g] = *((gptr<const f32x4>)(b_in + (rowoff_t + (unsigned)g * (unsigned)l_ld)));
              else {
#pragma unroll
                for (int q = 0; q < 4; ++q) hs[J & 1][g][q] = fbc[J / RPW] + 8 * g + q < nlim ? b_in[rowoff + 8 * g + q] : 0.f;
              }
            }
            if (HAS_EX) {
              const gptr<const float> b_ex = p_side_ex + tile_off;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                if (FULL) ex[J & 1][g] = *((gptr<const f32x4>)(b_ex + (rowoff_t + (unsigned)g * (unsigned)l_ld)));
                else {
#pragma unroll
                  for (int q = 0; q < 4; ++q) ex[J & 1][g][q] = fbc[J / RPW] + 8 * g + q < nlim ? b_ex[rowoff + 8 * g + q] : 0.f;
                }
              }
            }
          };
          // activations (forward) / deltas (backward, tangent) of block J -> side_out.  The values stay in acc0 until phase B,
          // so the backward issues these stores AFTER the last block's side loads: loads and stores share one in-order
          // counter (vmcnt), and a block's loads queued behind its predecessor's stores cost 24 k instead of 10 k cycles per layer
          auto side_store = [&](auto jt, auto ft, auto lt) {
            constexpr int J = decltype(jt)::value;
            constexpr bool FULL = decltype(ft)::value;
            constexpr bool BLK = decltype(lt)::value;
            if (!p_side_out) return;
            const unsigned rowoff = (unsigned)((rb0 + J % RPW) * 32 + r_o) * (unsigned)l_ld + (unsigned)fbc[J / RPW];
            const gptr<float> b_out = p_side_out + tile_off;
            const int lim = MODE == 0 ? l_N : nlim;
            if (BLK) {
              const gptr<float> b_blk = b_out + (((unsigned)(rb0 + J % RPW) * (unsigned)l_ld + (unsigned)fbc[J / RPW]) * 32u + (unsigned)r_o);
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) if (FULL || fbc[J / RPW] + 8 * g + q < lim) b_blk[(8 * g + q) * 32] = acc0[J][4 * g + q];
              return;
            }
            if (FULL && MODE == 0) {
              // forward: 16 bytes per lane at the lane's own point (its epilogue is VALU-bound: the transposition below costs more
              // issue slots than the wider stores give back -- measured +2 %)
#pragma unroll
              for (int g = 0; g < 4; ++g)
                *((gptr<f32x4>)(b_out + (rowoff + 8 * g))) = f32x4{acc0[J][4 * g], acc0[J][4 * g + 1], acc0[J][4 * g + 2], acc0[J][4 * g + 3]};
            } else if (FULL) {
              // full rows: a transposed copy of the block (quad_transpose), instruction g writes chunk (lane & 3) of point (r & ~3) + g
              f32x4 t[4];
#pragma unroll
              for (int g = 0; g < 4; ++g) t[g] = f32x4{acc0[J][4 * g], acc0[J][4 * g + 1], acc0[J][4 * g + 2], acc0[J][4 * g + 3]};
              quad_transpose(t, lane_o);
              const unsigned rowoff_t = (unsigned)((rb0 + J % RPW) * 32 + (r_o & ~3)) * (unsigned)l_ld + (unsigned)(nbc[J / RPW] * 32 + 8 * (r_o & 3) + 4 * hh);
#pragma unroll
              for (int g = 0; g < 4; ++g) *((gptr<f32x4>)(b_out + (rowoff_t + (unsigned)g * (unsigned)l_ld))) = t[g];
            } else {
#pragma unroll
              for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) if (fbc[J / RPW] + 8 * g + q < lim) b_out[rowoff + 8 * g + q] = acc0[J][4 * g + q];
            }
          };
          auto hidden_block = [&](auto jt, auto ft, auto et, auto lt) {
            constexpr int J = decltype(jt)::value;
            constexpr bool FULL = decltype(ft)::value;
            constexpr bool HAS_EX = decltype(et)::value;      // backward: an extra adjoint is added (tangent: always has s)
            constexpr bool BLK = decltype(lt)::value;         // point-blocked side tensors: nothing to transpose
            const int R = (rb0 + J % RPW) * 32 + r_o;             // row of the tile
            const float sa = s_ainv[R];
            const unsigned rowoff = (unsigned)R * (unsigned)l_ld + (unsigned)fbc[J / RPW];
            if (MODE == 0) {
              const float kk = fwd_kk(sa, winvc[J / RPW], b2);
              gptr<const float> rbp = nullptr;
              if (p_rowbias) rbp = p_rowbias + (long long)((unsigned)(row0 + R) / (unsigned)rb_div) * l_N + fbc[J / RPW];
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 rbv = {0.f, 0.f, 0.f, 0.f};
                if (p_rowbias) {
                  if (FULL) rbv = *((gptr<const f32x4>)(rbp + 8 * g));
                  else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (fbc[J / RPW] + 8 * g + q < l_N) rbv[q] = rbp[8 * g + q];
                  }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int i = 4 * g + q;
                  float u = fwd_u(acc0[J][i], kk, bbv[J / RPW][g][q]);       // b2 * (pre-activation)
                  if (p_rowbias) u = fwd_u_rowbias(u, rbv[q], b2);   // (uniform branch; without a row term fma(0, b2, u) = u exactly)
                  float v = softplus_u(u, ib2sc);
                  if (!FULL) v = (fbc[J / RPW] + 8 * g + q < nlim) ? v : 0.f;
                  acc0[J][i] = v;
                }
              }
            } else {
              const float saw = sa * winvc[J / RPW];
              if (FULL && !BLK) {        // the side loads came in by full rows: back to "four chunks of my point"
                quad_transpose(hs[J & 1], lane_o);
                if (HAS_EX) quad_transpose(ex[J & 1], lane_o);
              }
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 x2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int i = 4 * g + q;
                  const float zz = acc0[J][i] * saw;          // (sa, winvc[J / RPW]: powers of two -- one exact product)
                  const float e = __builtin_amdgcn_exp2f(nb2 * hs[J & 1][g][q]);
                  const float sp = __builtin_fmaf(-e, sc, sc);   // softplus' (1 - e) * (skip scale)
                  float v;
                  if (MODE == 1) v = HAS_EX ? zz * sp + ex[J & 1][g][q] : zz * sp;
                  else { v = zz * sp; x2[q] = beta * zz * ex[J & 1][g][q] * e; }
                  if (!FULL) {
                    const int f = fbc[J / RPW] + 8 * g + q;
                    if (MODE == 1 && is_skip && a.Xskip && f >= a.skip_split && f < l_N)
                      a.Xskip[(row0 + R) * a.ld_xskip + (f - a.skip_split)] = zz * sc;
                    if (f >= nlim) { v = 0.f; x2[q] = 0.f; }
                  }
                  acc0[J][i] = v;
                }
                if (MODE == 2 && p_side_out2) {
                  const gptr<float> b_out2 = p_side_out2 + tile_off;
                  if (BLK) {
                    const gptr<float> b_blk = b_out2 + (((unsigned)(rb0 + J % RPW) * (unsigned)l_ld + (unsigned)fbc[J / RPW]) * 32u + (unsigned)r_o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (FULL || fbc[J / RPW] + 8 * g + q < nlim) b_blk[(8 * g + q) * 32] = x2[q];
                  } else if (FULL) ex[J & 1][g] = x2;      // (its s values are consumed: the second output leaves by full rows below)
                  else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (fbc[J / RPW] + 8 * g + q < nlim) b_out2[rowoff + 8 * g + q] = x2[q];
                  }
                }
              }
              if (MODE == 2 && FULL && !BLK && p_side_out2) {
                const gptr<float> b_out2 = p_side_out2 + tile_off;
                quad_transpose(ex[J & 1], lane_o);
                const unsigned rowoff_t = (unsigned)((rb0 + J % RPW) * 32 + (r_o & ~3)) * (unsigned)l_ld + (unsigned)(nbc[J / RPW] * 32 + 8 * (r_o & 3) + 4 * hh);
#pragma unroll
                for (int g = 0; g < 4; ++g) *((gptr<f32x4>)(b_out2 + (rowoff_t + (unsigned)g * (unsigned)l_ld))) = ex[J & 1][g];
              }
            }
            // (backward / tangent: the block's side stores wait until every block's side loads have been consumed, see `run`)
            if (MODE != 1) side_store(jt, ft, lt);
            // row maximum of the block's 16 values of this point: v_max ignores NaN; a value set holding an Inf (or only
            // NaN) goes through the bit-pattern filter.  One LDS atomic per lane (the two half-waves of a row: 2-way).
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(acc0[J][i]));
            if (!(m < 3.0e38f)) {
              unsigned mb = 0;
#pragma unroll
              for (int i = 0; i < 16; ++i) { const unsigned b = finite_abs_bits(acc0[J][i]); mb = b > mb ? b : mb; }
              m = __uint_as_float(mb);
            }
            atomicMax(&s_rmax[cur][R], __float_as_uint(m));
          };
          auto run = [&](auto ft, auto et, auto lt) {
            lane_values();
            load_winv();
            load_bias(ft);
            if (MODE != 0) side_loads(I0{}, ft, et, lt);
            if (MODE != 0) side_loads(I1{}, ft, et, lt);
            sum_accs();
            pin_winv();
            scale_bias();
            __builtin_amdgcn_sched_barrier(0);
            // (diagnostic build -DNDJIR_CHAIN_SUBSTAMP, tools/chain_timeline.py sub: stamps of layer 2's row blocks in timeline slot 5)
#ifdef NDJIR_CHAIN_SUBSTAMP
#define NDJIR_SUB(P) if (li == 2) stamp(5, P)
#else
#define NDJIR_SUB(P)
#endif
            // (fences between the blocks: the loads of block J + 2 must not drift above the math of block J, whose registers
            // they take over; the forward pass, which has no side loads, measured the same with and without them)
            NDJIR_SUB(0);
            hidden_block(I0{}, ft, et, lt);
            __builtin_amdgcn_sched_barrier(0);
            NDJIR_SUB(1);
            if constexpr (NBLK > 2) { if (MODE != 0 && two) side_loads(I2{}, ft, et, lt); }
            hidden_block(I1{}, ft, et, lt);
            __builtin_amdgcn_sched_barrier(0);
            NDJIR_SUB(2);
            if constexpr (NBLK > 2) {
              if (two) {
                if (MODE != 0) side_loads(I3{}, ft, et, lt);
                hidden_block(I2{}, ft, et, lt);
                __builtin_amdgcn_sched_barrier(0);
                NDJIR_SUB(3);
                hidden_block(I3{}, ft, et, lt);
                __builtin_amdgcn_sched_barrier(0);
                NDJIR_SUB(4);
              }
            }
#undef NDJIR_SUB
            if (MODE == 1) {
              side_store(I0{}, ft, lt);
              side_store(I1{}, ft, lt);
              if constexpr (NBLK > 2) { if (two) { side_store(I2{}, ft, lt); side_store(I3{}, ft, lt); } }
            }
          };
          if (a.side_blocked) {
            // (point-blocked side tensors: a block is "full" whatever the row stride)
            const bool full_b = full_cols && (!p_rowbias || (l_N & 3) == 0);
            if constexpr (MODE == 1) {
              if (p_side_ex) { if (full_b) run(TT{}, TT{}, TT{}); else run(FF{}, TT{}, TT{}); }
              else { if (full_b) run(TT{}, FF{}, TT{}); else run(FF{}, FF{}, TT{}); }
            } else if constexpr (MODE == 2) {
              if (full_b) run(TT{}, TT{}, TT{}); else run(FF{}, TT{}, TT{});
            } else {
              if (full_b) run(TT{}, FF{}, TT{}); else run(FF{}, FF{}, TT{});
            }
          } else if constexpr (MODE == 1) {
            if (p_side_ex) { if (full) run(TT{}, TT{}, FF{}); else run(FF{}, TT{}, FF{}); }
            else { if (full) run(TT{}, FF{}, FF{}); else run(FF{}, FF{}, FF{}); }
          } else if constexpr (MODE == 2) {
            if (full) run(TT{}, TT{}, FF{}); else run(FF{}, TT{}, FF{});
          } else {
            if (full) run(TT{}, FF{}, FF{}); else run(FF{}, FF{}, FF{});
          }
          // bias gradient: column sums of the deltas over the wave's points (its blocks first: they share the features) --
          // 16-lane rows by DPP (xor 1, xor 2, half mirror, mirror), then one LDS atomic per feature from the first lane of
          // every row
          if (MODE != 0 && p_bgrad) {
            lane_values();
#pragma unroll
            for (int cb = 0; cb < CPW; ++cb) {
              if (cb > 0 && !two) break;
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                float c = acc0[cb * RPW][i] + acc0[cb * RPW + 1][i];
                if constexpr (RPW > 2) c += acc0[2][i] + acc0[3][i];
                c += dpp<DPP_XOR1>(c);
                c += dpp<DPP_XOR2>(c);
                c += dpp<DPP_HALF_MIRROR>(c);
                c += dpp<DPP_MIRROR>(c);
                const int f = fbc[cb] + acc_feat(i, 0);
                if ((lane_o & 15) == 0 && f < nlim) atomicAdd(p_bgrad + f, c);
              }
            }
          }
        }
        stamp(li, 2);

        // ---- row maxima of this layer's outputs are in s_rmax[cur] (incl. the skip concatenation's input part) ----
        if (MODE != 1 && is_skip && tid < TM) {
          const float xm = __uint_as_float(s_xmax[xpar][tid]) * fabsf(a.skip_scale);
          atomicMax(&s_rmax[cur][tid], __float_as_uint(xm));
        }
        __syncthreads();          // every wave has read the planes and contributed its maxima
        stamp(li, 3);

        // ================= phase B: per-row scale, 2-way split, planes updated in place =================
        if (active) {
          const int lane_o = fresh_lane();
          const int r_o = lane_o & 31, hh = lane_o >> 5;
#pragma unroll
          for (int J = 0; J < NBLK; ++J) {
            if (J >= RPW && !actc[CPW - 1]) break;
            const int kb = nbc[J / RPW] * 32 + 4 * hh;
            const int R = (rb0 + J % RPW) * 32 + r_o;
            float s_row, inv_row;
            scale_from_max(s_rmax[cur][R], s_row, inv_row);
#pragma unroll
            for (int g = 0; g < 4; ++g)
              put4(kb + 8 * g, R, f32x4{acc0[J][4 * g], acc0[J][4 * g + 1], acc0[J][4 * g + 2], acc0[J][4 * g + 3]}, s_row);
          }
        }
      }
      if (tid < TM) {
        float s_row, inv_row;
        scale_from_max(s_rmax[cur][tid], s_row, inv_row);
        s_ainv[tid] = inv_row;           // read by the next layer's phase A, two barriers on
      }
      if (ly.side_amax && tid < TM) {
        const float wm = wave_max(__uint_as_float(s_rmax[cur][tid]));
        if (lane == 0) atomicMax(ly.side_amax, __float_as_uint(wm));
      }

      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output ----
      if (MODE != 1 && is_skip) {
        __syncthreads();
        const int K0 = a.K0, base = l_N;
        const float* X = a.X + row0 * a.ldx;
        for (int t = tid; t < K0 * TM; t += NTHREADS) {
          const int k = t % K0, m = t / K0;
          const float v = X[(long long)m * a.ldx + k] * a.skip_scale;
          const int kk = base + k;
          float s_row, inv_row;
          scale_from_max(s_rmax[cur][m], s_row, inv_row);
          put1(kk, m, v, s_row);
          if (ly.side_out && !a.side_blocked) ly.side_out[(row0 + m) * ly.ld_side + kk] = v;
        }
        if (ly.side_out && a.side_blocked) {      // point-blocked: points fastest, full 128-byte lines (the tile's X is in L1 / L2)
          float* so = ly.side_out + row0 * ly.ld_side;
          for (int t = tid; t < K0 * TM; t += NTHREADS) {
            const int m = t % TM, k = t / TM;
            so[((unsigned)(m >> 5) * (unsigned)ly.ld_side + (unsigned)(base + k)) * 32u + (unsigned)(m & 31)] = X[(long long)m * a.ldx + k] * a.skip_scale;
          }
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
   }   // nets
  }
  stamp(MAX_CHAIN_LAYERS - 1, 3);
  stamp_rt(1);
  if (TEAMS == 2 && team == 0) __syncthreads();      // (pairs with team 1's last barrier)
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

// TMv = 128: <RPW, NWAVES> = <4, 8>, <2, 8>, <4, 4>;  TMv = 64: <2, 4> with two column blocks per wave (two workgroups per CU)
//            <2, 8> with TMv = 64: two TEAMS of four waves in one workgroup, a 64-point tile each
template <int MODE, int RPW, int NWAVES, int TMv = 128>
__global__ void __launch_bounds__(NWAVES * 64, 2) k_chainw(ChainArgs a) {
  constexpr int TEAMS = (TMv == 64 && NWAVES == 8) ? 2 : 1;
  chainw_body<MODE, RPW, NWAVES / TEAMS, TMv, TMv == 64 ? 2 : 1, TEAMS>(OneNet{a});
}
template <int MODE, int RPW, int NWAVES, int TMv = 128>
__global__ void __launch_bounds__(NWAVES * 64, 2) k_chainw_nets(ChainGroup /* read in the kernel-argument segment */) {
  constexpr int TEAMS = (TMv == 64 && NWAVES == 8) ? 2 : 1;
  chainw_body<MODE, RPW, NWAVES / TEAMS, TMv, TMv == 64 ? 2 : 1, TEAMS>(ManyNets{(const KGroup*)__builtin_amdgcn_kernarg_segment_ptr()});
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace x3w

#ifndef NDJIR_NO_LAUNCHER      // (tools/isa_one.sh compiles ONE instantiation of the kernel above for a look at its ISA)
// mlp3p.hip: the same tile, planes and arithmetic with the two 64-point halves of the tile software-pipelined (round 6)
int launch_chainp_group(const ChainGroup& grp, int mode, int blocks, size_t lds_bytes, hipStream_t stream);

// Which modes of the <mode, 4, 8, 128> launches go to the pipelined kernel: bit 0 forward, bit 1 backward, bit 2 tangent (training
// passes only, see launch_chainw_group).  set >= 0 sets it (ndjir_mlp_set_chain_pipeline); the first query reads NDJIR_CHAINP.
constexpr int CHAINP_DEFAULT = 0;
static int g_chainp = -1;
int chain_pipeline(int set) {
  if (set >= 0) g_chainp = set & 7;
  if (g_chainp < 0) { const char* e = getenv("NDJIR_CHAINP"); g_chainp = e ? (atoi(e) & 7) : CHAINP_DEFAULT; }
  return g_chainp;
}

// One net's part of a launch plan: the argument block with everything but the LDS offsets filled in.
struct WidePlan {
  ChainArgs b;
  int rpw;                  // row blocks per wave its widest hidden layer asks for
  int wmax;                 // widest activation its planes hold
  int bg_n, bg_total;
  float* bg_ptr[MAX_CHAIN_LAYERS + 1];
  int bg_off[MAX_CHAIN_LAYERS + 1];
};

static int wide_plan(const ChainArgs& a, int mode, WidePlan& p) {
  using namespace x3w;
  if ((a.P % TM_MAX) != 0) return NDJIR_ERR_UNSUPPORTED;
  if (a.P < 128 * 256 && a.tile_rows != 128) return NDJIR_ERR_UNSUPPORTED;      // (a forced 128 takes small launches too: tests)
  int wmax = round_up(a.K0, 16), hmax = 0;
  for (int i = 0; i < a.L; ++i) {
    const bool last = a.has_output && i == a.L - 1;
    if (!last && (a.layers[i].Np > 8 * 32 || a.layers[i].Np < 64)) return NDJIR_ERR_UNSUPPORTED;
    if (!last && a.layers[i].Np > wmax) wmax = a.layers[i].Np;
    if (!last && a.layers[i].Np > hmax) hmax = a.layers[i].Np;
  }
  p.rpw = hmax <= 128 ? 2 : 4;      // widest hidden layer: <= 4 column blocks -> two waves per column block
  if (a.skip_layer >= 0 && mode != 1) { int w = round_up(a.layers[a.skip_layer].N + a.K0, 16); if (w > wmax) wmax = w; }
  p.wmax = wmax;
  p.b = a;
  p.b.K0p = round_up(a.K0, 16);
  p.b.n_tiles = a.P / TM_MAX;       // (the launcher divides by its tile height)
  p.bg_n = 0;
  int bg_total = 0;
  if (mode != 0) {
    for (int i = 0; i < a.L; ++i) if (a.layers[i].bgrad && !(a.has_output && i == a.L - 1)) {
      p.b.layers[i].bg_off = bg_total;
      p.bg_ptr[p.bg_n] = a.layers[i].bgrad;
      p.bg_off[p.bg_n] = bg_total;
      ++p.bg_n;
      bg_total += a.layers[i].N;
    } else p.b.layers[i].bgrad = nullptr;
  }
  if (mode != 0 && a.in_bgrad) {
    p.b.in_bg_off = bg_total;
    p.bg_ptr[p.bg_n] = a.in_bgrad;
    p.bg_off[p.bg_n] = bg_total;
    ++p.bg_n;
    bg_total += a.K0;
  }
  bg_total = (bg_total + 3) & ~3;       // (partial rows a multiple of 16 bytes long: the step's reduction reads them with 16-byte loads)
  p.b.bg_total = p.bg_total = bg_total;
  if (bg_total > 0 && !a.bg_partial) return NDJIR_ERR_ARG;
  return NDJIR_OK;
}

// NDJIR_ERR_UNSUPPORTED = "not a launch for this kernel" (the caller falls back to mlp3.hip's 64 / 32-point tiles; a group is
// launched net by net).
int launch_chainw_group(const ChainArgs* nets, int n, int mode, hipStream_t stream) {
  using namespace x3w;
  constexpr int LDS_DYN_MAX = 160 * 1024 - 4096;      // the kernel also holds 2.5 KB of static LDS (row maxima / scales)
  if (n < 1 || n > MAX_GROUP_NETS) return NDJIR_ERR_UNSUPPORTED;
  static bool attr_set = false;
  static int nw4 = 1, t64 = 0, teams_on = 0;
  if (!attr_set) {
#define NDJIR_SET(M, R, W) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainw<M, R, W>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX); \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainw_nets<M, R, W>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX)
    NDJIR_SET(0, 4, 8); NDJIR_SET(1, 4, 8); NDJIR_SET(2, 4, 8); NDJIR_SET(0, 2, 8); NDJIR_SET(1, 2, 8); NDJIR_SET(2, 2, 8);
    NDJIR_SET(0, 4, 4); NDJIR_SET(1, 4, 4); NDJIR_SET(2, 4, 4);
#undef NDJIR_SET
    // Two round-5 experiments, both measured slower and NOT built by default (-DNDJIR_CHAINW_EXPERIMENTS builds them:
    // tools/build_variant.sh exp mlp3w.hip -DNDJIR_CHAINW_EXPERIMENTS; then NDJIR_CHAINW_T64=7 / NDJIR_CHAINW_TEAMS=7, bit 0
    // forward, 1 backward, 2 tangent; nets with hidden layers wider than 128 columns whose planes + bias sums fit):
    //   * T64: 64-point tiles, TWO 4-wave workgroups per CU (chainw_body, TM = 64 / CPW = 2).  Default step 8.40 ms against 8.10
    //     (same box), with or without a static priority for one workgroup of a CU's pair: the two fall into step, a layer of a
    //     64-point tile takes the 30 - 33 k cycles the 128-point tile takes, the tangent chain +40 % (weights stream twice).
    //   * TEAMS: the same two tiles as two teams of ONE 8-wave workgroup, one barrier apart (chainw_body, TEAMS = 2), so that
    //     the intervals pair up as (k-loop, split), (epilogue, k-loop), (split, epilogue).  9.62 ms against 8.13.  The pairing
    //     works -- but a wave's activation epilogue ALONE on its SIMD takes 12 k cycles for its 64 elements (15 cycles per
    //     vector instruction: dependent chains through exp / log), while the two waves of a SIMD in their epilogues TOGETHER
    //     take 11 k for both: the epilogue is latency-bound per wave, and the lock-step schedule is what hides it.
#ifdef NDJIR_CHAINW_EXPERIMENTS
#define NDJIR_SETX(M, W) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainw<M, 2, W, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX); \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chainw_nets<M, 2, W, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_MAX)
    NDJIR_SETX(0, 4); NDJIR_SETX(1, 4); NDJIR_SETX(2, 4); NDJIR_SETX(0, 8); NDJIR_SETX(1, 8); NDJIR_SETX(2, 8);
#undef NDJIR_SETX
    const char* e64 = getenv("NDJIR_CHAINW_T64");
    t64 = e64 ? atoi(e64) : 0;
    const char* et = getenv("NDJIR_CHAINW_TEAMS");
    teams_on = et ? atoi(et) : 0;
#else
    t64 = 0; teams_on = 0;
#endif
    // 4-wave workgroups (two per CU) for nets of up to 4 column blocks whose planes fit twice: bit 0 forward, bit 1 backward,
    // bit 2 tangent.  Measured (env-light / soft-visibility nets, 131072 points): forward 145 -> 133 us, backward 166 -> 173 us
    const char* e = getenv("NDJIR_CHAINW_NW4");
    nw4 = e ? atoi(e) : 1;
    attr_set = true;
  }
  // (static: 3.5 KB of argument blocks + plans stay off the stack of every chain call; launches are serialised per thread)
  static thread_local WidePlan plan[MAX_GROUP_NETS];
  static thread_local ChainGroup grp;
  int wmax = 0, bg_sum = 0;
  bool own_four[MAX_GROUP_NETS];
  for (int i = 0; i < n; ++i) {
    const int rc = wide_plan(nets[i], mode, plan[i]);
    if (rc != NDJIR_OK) return rc;
    if (plan[i].rpw != plan[0].rpw || nets[i].P != nets[0].P) return NDJIR_ERR_UNSUPPORTED;
    if (plan[i].wmax > wmax) wmax = plan[i].wmax;
    bg_sum += plan[i].bg_total;
    // (the decision a launch of this net alone takes: its planes + its own accumulators)
    own_four[i] = plan[i].rpw != 4 && ((nw4 >> mode) & 1) && (size_t)2 * (plan[i].wmax / 8) * tile_pad(128) * 16 + (size_t)plan[i].bg_total * 4 <= 77 * 1024;
  }
  const int rpw = plan[0].rpw;
  // (two workgroups per CU: 2 x (dynamic + static) <= 160 KB; the 64-point kernel holds 1.3 KB of static LDS)
  constexpr size_t LDS_HALF = 80 * 1024 - 1536;
  const bool tile64 = rpw == 4 && ((t64 >> mode) & 1) && (size_t)2 * (wmax / 8) * tile_pad(64) * 16 + (size_t)bg_sum * 4 <= LDS_HALF;
  if (n > 1)            // (a group runs the kernel each of its nets would run alone: the grid its deferred bias partials were laid out for)
    for (int i = 0; i < n; ++i)
      if (((size_t)2 * (plan[i].wmax / 8) * tile_pad(64) * 16 + (size_t)plan[i].bg_total * 4 <= LDS_HALF) != tile64 && rpw == 4 && ((t64 >> mode) & 1))
        return NDJIR_ERR_UNSUPPORTED;
  // two teams in one workgroup: both teams' planes + the bias sums in the one workgroup's LDS
  const bool teams = !tile64 && rpw == 4 && ((teams_on >> mode) & 1) &&
                     (size_t)2 * 2 * (wmax / 8) * tile_pad(64) * 16 + (size_t)bg_sum * 4 <= (size_t)LDS_DYN_MAX;
  const int tm = (tile64 || teams) ? 64 : 128;
  const int lds_split = (wmax / 8) * tile_pad(tm);             // 16-byte units per plane
  size_t lds_bytes = (size_t)2 * lds_split * 16 * (teams ? 2 : 1);
  int bg_lds = (int)(lds_bytes / 4);
  lds_bytes += (size_t)bg_sum * 4;
  if (lds_bytes > LDS_DYN_MAX) return NDJIR_ERR_UNSUPPORTED;
  // (two workgroups per CU: 2 x (dynamic + 2.5 KB static) <= 160 KB)
  const bool four = rpw != 4 && ((nw4 >> mode) & 1) && lds_bytes <= 77 * 1024;
  // a group runs the kernel each of its nets would run alone (the symbol a profile is keyed by, the grid the deferred bias
  // partials were laid out for)
  for (int i = 0; i < n && n > 1; ++i) if (own_four[i] != four) return NDJIR_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i) plan[i].b.n_tiles = nets[i].P / (teams ? 128 : tm);
  long long blocks = plan[0].b.n_tiles;
  if (blocks > 256LL * 8) blocks = 256LL * 8;
  if (tile64 && blocks > 512) blocks = 512;         // (two per CU, resident for the whole launch: the priority pairing above)
  static const int t64_prio = [] { const char* e = getenv("NDJIR_CHAINW_T64_PRIO"); return e ? atoi(e) : 1; }();
  // the pipelined kernel (mlp3p.hip) for the <mode, 4, 8, 128> launches: bit 0 forward, bit 1 backward, bit 2 tangent (NDJIR_CHAINP=0:
  // this file's kernel)
  const int chainp = chain_pipeline(-1);
  // ... for TRAINING passes only (point-blocked side tensors, every hidden layer of a forward launch stores its activation): the
  // pipelined kernel accumulates the three partial products in ONE register (mlp3p.hip) -- its results agree with this file's
  // and mlp3.hip's to round-off, not bit for bit, and forward values that are merged across launches (the sampler's SDF
  // rounds, the SDF volume) must not depend on the launch's size
  bool pipe = rpw == 4 && !tile64 && !teams && ((chainp >> mode) & 1);
  for (int i = 0; i < n && pipe; ++i) {
    if (!nets[i].side_blocked) pipe = false;
    const int hidden = nets[i].has_output ? nets[i].L - 1 : nets[i].L;
    for (int j = 0; j < hidden && pipe; ++j) if (!nets[i].layers[j].side_out) pipe = false;
  }
  if (bg_sum > 0 && blocks > CHAIN_MAX_GRID_BG) blocks = CHAIN_MAX_GRID_BG;
  if (n > 1) {          // ... which caps the grid of a net with bias gradients only: every member has to agree on it
    for (int i = 0; i < n; ++i) {
      long long own = plan[i].b.n_tiles > 256LL * 8 ? 256LL * 8 : plan[i].b.n_tiles;
      if (plan[i].bg_total > 0 && own > CHAIN_MAX_GRID_BG) own = CHAIN_MAX_GRID_BG;
      if (plan[i].bg_total > 0 && own != blocks) return NDJIR_ERR_UNSUPPORTED;
    }
  }
  grp.n = n; grp.pad = 0;
  for (int i = 0; i < n; ++i) {
    plan[i].b.lds_split = lds_split;
    plan[i].b.t64_prio = t64_prio;
    plan[i].b.bg_lds = bg_lds;
    bg_lds += plan[i].bg_total;
    grp.net[i] = plan[i].b;
  }
  if (nets[0].dry) {
    if (teams) snprintf(nets[0].dry->name, 64, "ndjir::x3w::k_chainw%s<%d, 2, 8, 64>", n > 1 ? "_nets" : "", mode);
    else if (tile64) snprintf(nets[0].dry->name, 64, "ndjir::x3w::k_chainw%s<%d, 2, 4, 64>", n > 1 ? "_nets" : "", mode);
    else if (pipe) snprintf(nets[0].dry->name, 64, "ndjir::x3p::k_chainp%s<%d>", n > 1 ? "_nets" : "", mode);
    else snprintf(nets[0].dry->name, 64, "ndjir::x3w::k_chainw%s<%d, %d, %d, 128>", n > 1 ? "_nets" : "", mode, (rpw == 4 || four) ? 4 : 2, four ? 4 : 8);
    nets[0].dry->blocks = (int)blocks; nets[0].dry->bg_total = plan[0].bg_total;
    return NDJIR_OK;
  }
#define NDJIR_GO(M, R, W)                                                                                                   \
  do {                                                                                                                      \
    if (n == 1) hipLaunchKernelGGL((k_chainw<M, R, W>), dim3((unsigned)blocks), dim3(W * 64), lds_bytes, stream, grp.net[0]); \
    else hipLaunchKernelGGL((k_chainw_nets<M, R, W>), dim3((unsigned)blocks), dim3(W * 64), lds_bytes, stream, grp);         \
  } while (0)
#ifdef NDJIR_CHAINW_EXPERIMENTS
#define NDJIR_GO64(M)                                                                                                          \
  do {                                                                                                                      \
    if (n == 1) hipLaunchKernelGGL((k_chainw<M, 2, 4, 64>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, grp.net[0]); \
    else hipLaunchKernelGGL((k_chainw_nets<M, 2, 4, 64>), dim3((unsigned)blocks), dim3(256), lds_bytes, stream, grp);         \
  } while (0)
#define NDJIR_GOT(M)                                                                                                           \
  do {                                                                                                                      \
    if (n == 1) hipLaunchKernelGGL((k_chainw<M, 2, 8, 64>), dim3((unsigned)blocks), dim3(512), lds_bytes, stream, grp.net[0]); \
    else hipLaunchKernelGGL((k_chainw_nets<M, 2, 8, 64>), dim3((unsigned)blocks), dim3(512), lds_bytes, stream, grp);         \
  } while (0)
  if (teams) { if (mode == 0) NDJIR_GOT(0); else if (mode == 1) NDJIR_GOT(1); else NDJIR_GOT(2); }
  else if (tile64) { if (mode == 0) NDJIR_GO64(0); else if (mode == 1) NDJIR_GO64(1); else NDJIR_GO64(2); }
  else
#endif
  if (pipe) { const int rcp = launch_chainp_group(grp, mode, (int)blocks, lds_bytes, stream); if (rcp != NDJIR_OK) return rcp; }
  else if (rpw == 4) { if (mode == 0) NDJIR_GO(0, 4, 8); else if (mode == 1) NDJIR_GO(1, 4, 8); else NDJIR_GO(2, 4, 8); }
  else if (four) { if (mode == 0) NDJIR_GO(0, 4, 4); else if (mode == 1) NDJIR_GO(1, 4, 4); else NDJIR_GO(2, 4, 4); }
  else { if (mode == 0) NDJIR_GO(0, 2, 8); else if (mode == 1) NDJIR_GO(1, 2, 8); else NDJIR_GO(2, 2, 8); }
#undef NDJIR_GO
#undef NDJIR_GO64
#undef NDJIR_GOT
  int rc = ndjir_check_launch();
  for (int i = 0; i < n && rc == NDJIR_OK; ++i)
    if (plan[i].bg_total > 0 && !nets[i].defer_bg_reduce)
      rc = launch_bgrad_reduce(nets[i].bg_partial, (int)blocks, plan[i].bg_total, plan[i].bg_ptr, plan[i].bg_off, plan[i].bg_n, nets[i].bg_accum, stream);
  return rc;
}

int launch_chainw(const ChainArgs& a, int mode, hipStream_t stream) { return launch_chainw_group(&a, 1, mode, stream); }
#endif

}  // namespace ndjir
