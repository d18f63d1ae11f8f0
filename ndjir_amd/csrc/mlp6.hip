// mlp6.hip -- the fused MLP chain on the bf16 matrix cores with fp32-equivalent arithmetic.
//
// gfx950 has no fp32-rate advantage on the matrix pipe: v_mfma_f32_32x32x2_f32 runs at the fp32
// VECTOR rate (157 TFLOP/s), v_mfma_f32_32x32x16_bf16 at 16x that.  Every fp32 operand is therefore
// split into three bf16 planes  x = hi + mid + lo  (8 + 8 + 8 mantissa bits: the split is EXACT)
// and a product of two such numbers is formed from the six partial products
//     lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi            (mid*lo, lo*mid, lo*lo ~ 2^-32: dropped)
// each an exact bf16 x bf16 product accumulated in fp32 by the MFMA.  The result carries the same
// ~2^-24 relative error as an fp32 FMA chain (measured on the 8-layer net: 2.9e-7 vs 5.5e-7 for plain
// fp32), at 6/16 of the fp32 MFMA time.  Weights are split once at pack time, activations once in the
// producing layer's epilogue (they are kept in LDS as three bf16 planes), so the k-loop is pure
// ds_read_b128 / global_load_dwordx4 / MFMA.
//
// Structure per tile of TM points, 8 waves:
//   for every layer:  k-loops of all column blocks (accumulators stay in registers)
//                     barrier (the activation planes are updated IN PLACE)
//                     epilogue per 32x32 block: accumulators -> per-wave fp32 staging tile in LDS ->
//                       row-major pass (bias + softplus / softplus' product, coalesced float4 side
//                       stores and loads, bias-gradient column sums) -> bf16 planes
//                     barrier
// The epilogue math, the side-array protocol (ChainArgs) and the bias-gradient path are those of
// mlp.hip; only the matrix arithmetic and the LDS formats differ.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#include <type_traits>

#include "common.h"
#include "mlp.h"

#ifdef X6_NO_SCHEDB
#define X6_SCHED_BARRIER()
#else
#define X6_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

namespace ndjir {
namespace x6 {

constexpr int NWAVES = 8;
constexpr int NTHREADS = NWAVES * 64;
constexpr int MAXNB = 16;        // widest layer: 512 columns (2 column blocks per wave)
constexpr int GPS = 32 * 4 + 4;  // staging tile: dwords per group of 4 columns (32 rows + pad)
constexpr int STG = 8 * GPS;     // staging dwords per wave (32 x 32 tile)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

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

__device__ __forceinline__ unsigned short bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }

// exact 3-way split of an fp32 value into bf16 planes (round-to-nearest at each step)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 hb = (__bf16)x;
  const float r1 = x - (float)hb;
  const __bf16 mb = (__bf16)r1;
  const float r2 = r1 - (float)mb;
  const __bf16 lb = (__bf16)r2;
  h = bf16_bits(hb); m = bf16_bits(mb); l = bf16_bits(lb);
}

// ---- weight packing -----------------------------------------------------------------------------
// dst (16-byte units): [Np/32][Kp/16][plane 0..2][lane 0..63], lane (c = lane & 31, h = lane >> 5) holds
// W[16 ks + 8 h + j][32 nb + c], j = 0..7, of plane p (0 = hi, 1 = mid, 2 = lo).  transpose packs W^T.
__global__ void __launch_bounds__(256) k_pack6(const float* __restrict__ W, unsigned short* __restrict__ dst, int K, int N,
                                               int transpose, int Kp, int Np) {
  const long long total = (long long)Kp * Np;       // one thread per matrix element
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t & 7);
    const int lane = (int)((t >> 3) & 63);
    const long long rest = t >> 9;
    const int KS = Kp >> 4;
    const int ks = (int)(rest % KS);
    const int nb = (int)(rest / KS);
    const int k = ks * 16 + 8 * (lane >> 5) + j;
    const int n = nb * 32 + (lane & 31);
    float v = 0.f;
    if (!transpose) { if (k < K && n < N) v = W[(long long)k * N + n]; }
    else { if (k < N && n < K) v = W[(long long)n * N + k]; }
    unsigned short h, m, l;
    split3(v, h, m, l);
    const long long base = (((long long)nb * KS + ks) * 3) * 64 * 8 + lane * 8 + j;
    dst[base] = h;
    dst[base + 64 * 8] = m;
    dst[base + 2 * 64 * 8] = l;
  }
}

// ---- the chain kernel ---------------------------------------------------------------------------
template <int MODE, int TM>
__global__ void __launch_bounds__(NTHREADS, 2) k_chain6(ChainArgs a) {
  constexpr bool BWD = (MODE == 1);
  constexpr int RB = TM / 32;
  constexpr int TMP = TM + 4;          // rows per k-group incl. pad: (TMP * 16) % 256 == 64 -> conflict-free plane writes
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PLANE = a.lds_split;       // 16-byte units per plane ( = k-groups * TMP )
  bf16x8* act = reinterpret_cast<bf16x8*>(lds);
  char* actb = reinterpret_cast<char*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  float* stage_all = lds + (size_t)3 * PLANE * 4;
  float* stage = stage_all + wave * STG;
  float* bsum = lds + a.bg_lds;
  const float beta = a.beta;
  auto stamp = [&](int li, int phase) {
    if (a.timeline && blockIdx.x == 0 && lane == 0) a.timeline[(li * 5 + phase) * NWAVES + wave] = (long long)__builtin_amdgcn_s_memtime();
  };
  // write 4 consecutive features k..k+3 (k % 4 == 0) of row m into the three planes
  auto put4 = [&](int k, int m, f32x4 v) {
    u16x4 ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) { unsigned short x, y, z; split3(v[q], x, y, z); ph[q] = x; pm[q] = y; pl[q] = z; }
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<u16x4*>(p) = ph;
    *reinterpret_cast<u16x4*>(p + (size_t)PLANE * 16) = pm;
    *reinterpret_cast<u16x4*>(p + (size_t)PLANE * 32) = pl;
  };
  auto put1 = [&](int k, int m, float v) {
    unsigned short x, y, z;
    split3(v, x, y, z);
    char* p = actb + ((size_t)((k >> 3) * TMP + m) * 16 + (k & 7) * 2);
    *reinterpret_cast<unsigned short*>(p) = x;
    *reinterpret_cast<unsigned short*>(p + (size_t)PLANE * 16) = y;
    *reinterpret_cast<unsigned short*>(p + (size_t)PLANE * 32) = z;
  };

  if (MODE != 0) for (int i = tid; i < a.bg_total; i += NTHREADS) bsum[i] = 0.f;
  if (MODE != 0 && a.in_bgrad) __syncthreads();    // the first tile's input load already accumulates into bsum

  for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long row0 = tile * TM;
    const int rows = (int)((a.P - row0) < TM ? (a.P - row0) : TM);

    // ---- chain input tile -> planes (zero padded to a multiple of 16 features) ----
    {
      const int K0p = a.K0p, K0 = a.K0;
      const float* X = a.X + row0 * a.ldx;
      const int groups = K0p >> 2;
      const bool vec = (a.ldx & 3) == 0 && ((uintptr_t)a.X & 15) == 0;
      for (int t = tid; t < groups * TM; t += NTHREADS) {
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
        put4(k, m, v);
        if (MODE != 0 && a.in_bgrad && m < rows) {      // bias gradient of the output layer: column sums of the input
#pragma unroll
          for (int q = 0; q < 4; ++q) if (k + q < K0) atomicAdd(bsum + a.in_bg_off + k + q, v[q]);
        }
      }
    }
    __syncthreads();

    for (int li = 0; li < a.L; ++li) {
      const ChainLayer& ly = a.layers[li];
      const int KS = (ly.Kp + 15) >> 4;          // k-steps of 16 (planes are zero beyond Kp)
      const int NB = ly.Np >> 5;
      const bool last = a.has_output && (li == a.L - 1);
      stamp(li, 0);
      const gptr<const bf16x8> p_wp = (gptr<const bf16x8>)pin(ly.Wp);
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

      // one k-loop: RBU row blocks of column block nb, accumulators acc[SLOT .. SLOT + RBU)
      f32x16 acc[4];   // static indices only (an accumulator array indexed under run-time branches goes to scratch)
      auto kloop = [&](auto rbu_tag, auto slot_tag, const int nb, const int rb0, const int ks0, const int ks1) {
        constexpr int RBU = decltype(rbu_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
        for (int q = 0; q < RBU; ++q) acc[SLOT + q] = f32x16{0};
        const gptr<const bf16x8> Bp = p_wp + ((long long)nb * KS) * 3 * 64 + lane;
        const bf16x8* A0 = act + h * TMP + rb0 * 32 + r;
        // Software pipeline (static register slots, unrolled by 3): weight fragments 3 steps ahead;
        // the lo / mid activation planes are re-loaded in place as soon as their last MFMA of the step
        // has issued, the hi plane (needed until the end of the step) rotates through 3 buffers.
        bf16x8 b[3][3];                 // [slot][plane]
        bf16x8 alo[RBU], amid[RBU], ahi[3][RBU];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int p = 0; p < 3; ++p) b[s][p] = Bp[(long long)((ks0 + s < ks1 ? ks0 + s : ks0) * 3 + p) * 64];
          X6_SCHED_BARRIER();
        }
        {
          const bf16x8* An = A0 + 2 * ks0 * TMP;
#pragma unroll
          for (int q = 0; q < RBU; ++q) { ahi[0][q] = An[q * 32]; amid[q] = An[PLANE + q * 32]; alo[q] = An[2 * PLANE + q * 32]; }
        }
        X6_SCHED_BARRIER();
        auto kstep = [&](auto stag, auto gtag, const int ks) {
          constexpr int S = decltype(stag)::value;
          constexpr bool GUARD = decltype(gtag)::value;
          const bool nxt = !GUARD || ks + 1 < ks1;
          const bf16x8* An = A0 + 2 * (ks + 1) * TMP;
          // six partial products; planes: hi, mid, lo.  Order: lo*hi, mid*mid, mid*hi, hi*lo, hi*mid, hi*hi
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[q], b[S][0], acc[SLOT + q], 0, 0, 0);
          if (nxt) {
#pragma unroll
            for (int q = 0; q < RBU; ++q) alo[q] = An[2 * PLANE + q * 32];
          }
          X6_SCHED_BARRIER();
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[q], b[S][1], acc[SLOT + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid[q], b[S][0], acc[SLOT + q], 0, 0, 0);
          if (nxt) {
#pragma unroll
            for (int q = 0; q < RBU; ++q) { amid[q] = An[PLANE + q * 32]; ahi[(S + 1) % 3][q] = An[q * 32]; }
          }
          X6_SCHED_BARRIER();
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[S][q], b[S][2], acc[SLOT + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[S][q], b[S][1], acc[SLOT + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < RBU; ++q) acc[SLOT + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[S][q], b[S][0], acc[SLOT + q], 0, 0, 0);
          if (!GUARD || ks + 3 < ks1) {
#pragma unroll
            for (int p = 0; p < 3; ++p) b[S][p] = Bp[(long long)((ks + 3) * 3 + p) * 64];
          }
          X6_SCHED_BARRIER();
        };
        {
          using T = std::true_type;
          using F = std::false_type;
          using S0 = std::integral_constant<int, 0>;
          using S1 = std::integral_constant<int, 1>;
          using S2 = std::integral_constant<int, 2>;
          int ks = ks0;
          for (; ks + 6 <= ks1; ks += 3) { kstep(S0{}, F{}, ks); kstep(S1{}, F{}, ks + 1); kstep(S2{}, F{}, ks + 2); }
          for (; ks < ks1; ks += 3) {
            kstep(S0{}, T{}, ks);
            if (ks + 1 < ks1) kstep(S1{}, T{}, ks + 1);
            if (ks + 2 < ks1) kstep(S2{}, T{}, ks + 2);
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
        for (int i = 0; i < 16; ++i) if (kq < KSPLIT) part[(kq * TM + rb * 32 + acc_row(i, h)) * 32 + r] = acc[0][i];
        __syncthreads();
        for (int t = tid; t < TM * 32; t += NTHREADS) {
          const int n = t & 31, m = t >> 5;
          float z = 0.f;
#pragma unroll
          for (int q = 0; q < KSPLIT; ++q) z += part[q * TM * 32 + t];
          if (n < l_N && m < rows) {
            if (MODE == 0) {
              z += p_bias ? p_bias[n] : 0.f;
              if (!last) { const float bz = beta * z; z = bz > 20.f ? z : log1pf(__expf(bz)) / beta; }
            }
            if (last) {
              float* y = a.Y + (row0 + m) * a.ldy + n;
              *y = a.accum_y ? *y + z : z;
            }
          }
        }
        __syncthreads();
        continue;
      }

      // ---- general layer: this wave's 32 x 32 output blocks ("jobs"), at most 4 ----
      // Column blocks that fill whole rounds of 8 waves go to one wave with all row blocks (the weight
      // fragments are fetched once per tile); the NB % 8 remainder blocks are split by row block.
      const int rem = (RB == 1) ? 0 : (NB % NWAVES);
      const int full = NB - rem;
      unsigned jobs = 0;    // job j (= accumulator slot j) in bits 8j..8j+7: column block | row block << 5
      int njobs = 0;
      auto job = [&](int j, int nb, int rb) { jobs |= (unsigned)(nb | (rb << 5)) << (8 * j); };
      if (RB == 2) {
        if (wave < full) { kloop(IRB{}, I0{}, wave, 0, 0, KS); job(0, wave, 0); job(1, wave, 1); njobs = 2; }
        if (wave + NWAVES < full) { kloop(IRB{}, I2{}, wave + NWAVES, 0, 0, KS); job(2, wave + NWAVES, 0); job(3, wave + NWAVES, 1); njobs = 4; }
        if (wave < rem * RB) {
          const int nb = full + wave / RB, rb = wave % RB;
          if (njobs == 0) kloop(I1{}, I0{}, nb, rb, 0, KS); else kloop(I1{}, I2{}, nb, rb, 0, KS);
          job(njobs, nb, rb); ++njobs;
        }
        if (wave + NWAVES < rem * RB) {
          const int nb = full + (wave + NWAVES) / RB, rb = (wave + NWAVES) % RB;
          if (njobs == 1) kloop(I1{}, I1{}, nb, rb, 0, KS); else kloop(I1{}, I3{}, nb, rb, 0, KS);
          job(njobs, nb, rb); ++njobs;
        }
      } else {
        if (wave < NB) { kloop(I1{}, I0{}, wave, 0, 0, KS); job(0, wave, 0); njobs = 1; }
        if (wave + NWAVES < NB) { kloop(I1{}, I1{}, wave + NWAVES, 0, 0, KS); job(1, wave + NWAVES, 0); njobs = 2; }
      }
      stamp(li, 1);
      if (!last) __syncthreads();          // every wave has read the planes: they may be overwritten now
      stamp(li, 2);

      constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
      const float b2 = beta * LOG2E, ib2 = LN2 / beta;
      const float hsc = (MODE != 0 && is_skip) ? 1.f / sc : 1.f;
      const float nb2 = -b2 * hsc;
      const int g = lane & 7;
#pragma unroll 1
      for (int j = 0; j < njobs; ++j) {
        const int nb = (jobs >> (8 * j)) & 31, rb0 = (jobs >> (8 * j + 5)) & 7;
        // pass 1: accumulators -> staging tile (conflict-free: column groups GPS apart, rows 4 dwords apart)
        {
          float* dst = stage + (r >> 2) * GPS + (r & 3);
          f32x16 v;
          if (j == 0) v = acc[0]; else if (j == 1) v = acc[1]; else if (j == 2) v = acc[2]; else v = acc[3];
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[acc_row(i, h) * 4] = v[i];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // pass 2: 4 steps of 8 rows; lane = (column group g of 4 columns, row)
        const int n4 = nb * 32 + g * 4;
        const int mbase = rb0 * 32 + (lane >> 3);
        const long long off0 = (row0 + mbase) * l_ld + n4;
        const bool fast = !last && (nb * 32 + 31 < nlim) && (l_ld & 3) == 0 && rows == TM && (!p_rowbias || (l_N & 3) == 0);
        f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
        const float* lp = stage + g * GPS + (lane >> 3) * 4;
        if (fast) {
          f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
          if (MODE == 0 && p_bias) bias4 = *((gptr<const f32x4>)(p_bias + n4));
          f32x4 hs[4];
          if (MODE != 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it) hs[it] = *((gptr<const f32x4>)(p_side_in + off0 + (long long)it * 8 * l_ld));
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const long long off = off0 + (long long)it * 8 * l_ld;
            const f32x4 z = *reinterpret_cast<const f32x4*>(lp + it * 32);
            f32x4 v;
            if (MODE == 0) {
              f32x4 rb = {0.f, 0.f, 0.f, 0.f};
              if (p_rowbias) rb = *((gptr<const f32x4>)(p_rowbias + ((row0 + mbase + 8 * it) / rb_div) * (long long)l_N + n4));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float t = z[q] + bias4[q] + rb[q];
                const float u = b2 * t;
                const float sp = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(u)) * ib2;
                v[q] = (u > 20.f * LOG2E ? t : sp) * sc;
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
            } else {
              f32x4 ex = {0.f, 0.f, 0.f, 0.f}, x2;
              if (MODE == 1 && p_side_add) ex = *((gptr<const f32x4>)(p_side_add + off));
              if (MODE == 2 && p_side_in2) ex = *((gptr<const f32x4>)(p_side_in2 + off));
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float e = __builtin_amdgcn_exp2f(nb2 * hs[it][q]);
                const float sp = (1.f - e) * sc;
                if (MODE == 1) v[q] = z[q] * sp + ex[q];
                else { v[q] = z[q] * sp; x2[q] = beta * z[q] * ex[q] * e; }
              }
              if (p_side_out) *((gptr<f32x4>)(p_side_out + off)) = v;
              if (MODE == 2 && p_side_out2) *((gptr<f32x4>)(p_side_out2 + off)) = x2;
              colsum += v;
            }
            put4(n4, mbase + 8 * it, v);
          }
        } else {
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
          for (int it = 0; it < 4; ++it) {
            const int m = mbase + 8 * it;
            const bool mrow = m < rows;
            const float rm = mrow ? 1.f : 0.f;
            const long long grow = row0 + m;
            const long long off = off0 + (long long)it * 8 * l_ld;
            const f32x4 z = *reinterpret_cast<const f32x4*>(lp + it * 32);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (last) {
              if (mrow) {
                float* y = a.Y + grow * a.ldy + n4;
                if (vec_ok && (a.ldy & 3) == 0 && ((uintptr_t)a.Y & 15) == 0) {
                  f32x4 t = z;
                  if (MODE == 0) t += bias4;
                  if (a.accum_y) t += *reinterpret_cast<const f32x4*>(y);
                  *reinterpret_cast<f32x4*>(y) = t;
                } else {
#pragma unroll
                  for (int q = 0; q < 4; ++q) {
                    if (n4 + q < l_N) {
                      const float t = z[q] + (MODE == 0 ? bias4[q] : 0.f);
                      y[q] = a.accum_y ? y[q] + t : t;
                    }
                  }
                }
              }
              continue;
            }
            if (MODE == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                float t = z[q] + bias4[q];
                if (p_rowbias && mrow && n4 + q < l_N) t += p_rowbias[(grow / rb_div) * (long long)l_N + n4 + q];
                const float u = b2 * t;
                const float sp = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(u)) * ib2;
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
                  if (p_side_out) p_side_out[off + q] = v[q];
                  if (MODE == 2 && p_side_out2) p_side_out2[off + q] = x2[q];
                }
              }
              colsum += v;
            }
            put4(n4, m, v);
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
        __builtin_amdgcn_wave_barrier();   // staging tile is reused by the next job
      }
      stamp(li, 3);

      // the planes beyond this layer's padded width must read as zero for the next layer's k-loop:
      // column blocks are written whole (32 columns), the k-loop reads multiples of 16 <= Np
      // ---- forward skip connection: append the (scaled) chain input after the skip layer's output ----
      if (MODE != 1 && li == a.skip_layer) {
        __syncthreads();
        const int K0 = a.K0, base = l_N;
        const float* X = a.X + row0 * a.ldx;
        for (int t = tid; t < K0 * TM; t += NTHREADS) {
          const int k = t % K0, m = t / K0;
          const float v = (m < rows) ? X[(long long)m * a.ldx + k] * a.skip_scale : 0.f;
          const int kk = base + k;
          put1(kk, m, v);
          if (m < rows && ly.side_out) ly.side_out[(row0 + m) * ly.ld_side + kk] = v;
        }
        // zero the tail up to the next multiple of 16
        const int wcat = base + K0, wpad = (wcat + 15) & ~15;
        for (int t = tid; t < (wpad - wcat) * TM; t += NTHREADS) put1(wcat + t % (wpad - wcat), t / (wpad - wcat), 0.f);
      }
      __syncthreads();
      stamp(li, 4);
    }
  }
  if (MODE != 0 && a.bg_total > 0) {
    __syncthreads();
    float* part = a.bg_partial + (long long)blockIdx.x * a.bg_total;
    for (int i = tid; i < a.bg_total; i += NTHREADS) part[i] = bsum[i];
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace x6

long long packed_size6(int K, int N, int transpose) {
  const int Kp = x6::round_up(transpose ? N : K, 16), Np = x6::round_up(transpose ? K : N, 32);
  return ((long long)Kp * Np * 6 + 3) / 4;          // three bf16 planes, in floats
}

int launch_pack6(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream) {
  const int Kp = x6::round_up(transpose ? N : K, 16), Np = x6::round_up(transpose ? K : N, 32);
  const long long total = (long long)Kp * Np;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(x6::k_pack6, dim3(blocks), dim3(256), 0, stream, W, reinterpret_cast<unsigned short*>(dst), K, N, transpose,
                     Kp, Np);
  return ndjir_check_launch();
}

int launch_chain6(const ChainArgs& a, int mode, hipStream_t stream) {
  using namespace x6;
  if (a.P <= 0) return NDJIR_OK;
  const int TM = a.tile_rows == 32 ? 32 : 64;
  const int TMP = TM + 4;
  // widest activation the planes have to hold: chain input, every hidden output (+ skip concat)
  int wmax = round_up(a.K0, 16);
  for (int i = 0; i < a.L; ++i) {
    const bool last = a.has_output && i == a.L - 1;
    if (a.layers[i].Np > MAXNB * 32) return NDJIR_ERR_UNSUPPORTED;
    if (!last && a.layers[i].Np > 32) { if (a.layers[i].Np > wmax) wmax = a.layers[i].Np; }
  }
  if (a.skip_layer >= 0 && mode != 1) { int w = round_up(a.layers[a.skip_layer].N + a.K0, 16); if (w > wmax) wmax = w; }
  ChainArgs b = a;
  b.K0p = round_up(a.K0, 16);
  b.lds_split = (wmax / 8) * TMP;                          // 16-byte units per plane
  size_t lds_bytes = (size_t)3 * b.lds_split * 16;
  size_t stage_bytes = (size_t)NWAVES * STG * 4;
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
#define NDJIR_SET(M, T) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain6<M, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    NDJIR_SET(0, 64); NDJIR_SET(1, 64); NDJIR_SET(2, 64); NDJIR_SET(0, 32); NDJIR_SET(1, 32); NDJIR_SET(2, 32);
#undef NDJIR_SET
    attr_set = true;
  }
  if (a.dry) { snprintf(a.dry->name, 64, "ndjir::x6::k_chain6<%d, %d>", mode, TM); a.dry->blocks = (int)blocks; a.dry->bg_total = -1; return NDJIR_OK; }
#define NDJIR_GO(M, T) hipLaunchKernelGGL((k_chain6<M, T>), dim3((unsigned)blocks), dim3(NTHREADS), lds_bytes, stream, b)
  if (TM == 64) { if (mode == 0) NDJIR_GO(0, 64); else if (mode == 1) NDJIR_GO(1, 64); else NDJIR_GO(2, 64); }
  else { if (mode == 0) NDJIR_GO(0, 32); else if (mode == 1) NDJIR_GO(1, 32); else NDJIR_GO(2, 32); }
#undef NDJIR_GO
  int rc = ndjir_check_launch();
  if (rc != NDJIR_OK) return rc;
  if (bg_total > 0) return launch_bgrad_reduce(a.bg_partial, (int)blocks, bg_total, bg_ptr, bg_off, bg_n, a.bg_accum, stream);
  return NDJIR_OK;
}

}  // namespace ndjir
