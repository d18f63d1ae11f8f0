// wgrad.hip -- weight-gradient GEMM  dW (K x N) = A^T B,  A (P x K) activations, B (P x N) deltas.
//
// The reduction runs over P = 10^4..10^5 points while K, N <= ~300: a "tall-skinny^T" GEMM that
// library heuristics serve with tiny tiles (measured 37 TFLOP/s).  Here: 128 x 128 output tiles,
// the point axis split over workgroups (partials + a reduce pass), 32-point chunks staged through
// double-buffered LDS, fp32 MFMA 32x32x2 with both operands read as conflict-free 128-byte rows
// (the contraction index is the row index of both matrices, so no transpose is ever needed).
// Workgroup -> (split, tile) mapping keeps the tiles that share a P-range on one XCD (shared L2).
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "mlp.h"

namespace ndjir {

constexpr int WG_T = 128;     // output tile edge
constexpr int WG_C = 32;      // points per chunk
constexpr int WG_THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Pointers that a kernel reads out of a table in device memory (k_wgrad_group) are GENERIC to the compiler: their loads become
// flat_load, which counts in lgkmcnt as well as vmcnt -- the first `s_waitcnt lgkmcnt(0)` in front of an LDS-fed MFMA then waits
// for every prefetched global load, and the chunk loop ran at HBM latency (9 k cycles per chunk against 768 cycles of MFMA,
// found in round 5).  Device functions reached from such kernels move their pointers into the global address space first.
#define WG_G __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ WG_G T* wg_global(T* p) { return (WG_G T*)p; }

// Tile = (WK*BK*32) x (WN*BN*32) outputs: WK x WN waves, each BK x BN MFMA blocks.  <2,2,2,2> is the
// 128 x 128 main tile; <1,4,1,1> (32 x 128) and <4,1,1,1> (128 x 32) cover the ragged strips of shapes
// like 259 x 256 or 256 x 257 without paying for a whole extra tile.  (k_off, n_off) = origin of the
// region this launch covers; the output / partial layout is always the full K x N matrix.
template <int WK, int WN, int BK, int BN>
__global__ void __launch_bounds__(WG_THREADS, 2) k_wgrad(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                         int ldb, int K, int N, long long P, float* __restrict__ partial,
                                                         int S, int tiles_k, int tiles_n, long long rows_per_split,
                                                         int k_off, int n_off, int k_end, int n_end) {
  static_assert(WK * WN == 4, "4 waves");
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  constexpr int EA = WG_C * TK / WG_THREADS, EB = WG_C * TN / WG_THREADS;   // staged elements per thread
  constexpr int RA = WG_THREADS / TK, RB = WG_THREADS / TN;                 // rows covered per pass
  __shared__ float As[2][WG_C][TK];
  __shared__ float Bs[2][WG_C][TN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave / WN, wn = wave % WN;
  const int T = tiles_k * tiles_n;
  // XCD-aware: blocks are dealt round-robin to the 8 XCDs; give every XCD whole splits
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int vid = (nblk & 7) == 0 ? xcd * (nblk >> 3) + local : blockIdx.x;
  const int split = vid / T, tile = vid - split * T;
  const int tk = tile / tiles_n, tn = tile - tk * tiles_n;
  const int k0 = k_off + tk * TK, n0 = n_off + tn * TN;
  const long long p_begin = (long long)split * rows_per_split;
  long long p_end = p_begin + rows_per_split;
  if (p_end > P) p_end = P;

  f32x16 acc[BK][BN] = {};
  float ra[EA], rb[EB];

  const int cola = tid % TK, rowa = tid / TK, colb = tid % TN, rowb = tid / TN;
  const bool acol = (k0 + cola) < k_end, bcol = (n0 + colb) < n_end;
  const float* Ap = A + (long long)rowa * lda + k0 + cola;
  const float* Bp = B + (long long)rowb * ldb + n0 + colb;
  const long long astep = (long long)RA * lda, bstep = (long long)RB * ldb;
  auto load_chunk = [&](long long p0) {
    const float* ap = Ap + p0 * lda;
    const float* bp = Bp + p0 * ldb;
    if (p0 + WG_C <= p_end) {
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = acol ? ap[i * astep] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = bcol ? bp[i * bstep] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < EA; ++i) ra[i] = (acol && p0 + rowa + RA * i < p_end) ? ap[i * astep] : 0.f;
#pragma unroll
      for (int i = 0; i < EB; ++i) rb[i] = (bcol && p0 + rowb + RB * i < p_end) ? bp[i * bstep] : 0.f;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) As[buf][rowa + RA * i][cola] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[buf][rowb + RB * i][colb] = rb[i];
  };

  if (p_begin < p_end) {
    load_chunk(p_begin);
    store_chunk(0);
    __syncthreads();
    int cur = 0;
    for (long long p0 = p_begin; p0 < p_end; p0 += WG_C) {
      const bool more = p0 + WG_C < p_end;
      if (more) load_chunk(p0 + WG_C);
#pragma unroll 4
      for (int s = 0; s < WG_C / 2; ++s) {
        const float* ar = &As[cur][2 * s + h][wk * BK * 32 + r];
        const float* br = &Bs[cur][2 * s + h][wn * BN * 32 + r];
        float av[BK], bv[BN];
#pragma unroll
        for (int i = 0; i < BK; ++i) av[i] = ar[32 * i];
#pragma unroll
        for (int j = 0; j < BN; ++j) bv[j] = br[32 * j];
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_chunk(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  float* out = partial + (long long)split * K * N;
#pragma unroll
  for (int bi = 0; bi < BK; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj) {
      const int n = n0 + (wn * BN + bj) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + (wk * BK + bi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < k_end && n < n_end) out[(long long)k * N + n] = acc[bi][bj][i];
      }
    }
}


// ---- the same tiles on the bf16 matrix cores (three-way split, six partial products; see mlp6.hip) ----
// A and B arrive as fp32 rows [point][feature]; the MFMA wants, per lane, 8 consecutive POINTS of one
// feature.  The loader therefore gives every thread one feature column and 4-point groups of it: the
// global loads stay coalesced along the features, the three bf16 planes are written to LDS as
// [plane][feature][point] (8-byte writes), and fragments are plain ds_read_b128.  Row pitch 40 bf16
// (80 B) keeps both the writes and the fragment reads (almost) conflict-free.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short wg_u16x4 __attribute__((ext_vector_type(4)));
constexpr int WG_CP = 40;

__device__ __forceinline__ void wg_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 hb = (__bf16)x;
  const float r1 = x - (float)hb;
  const __bf16 mb = (__bf16)r1;
  const float r2 = r1 - (float)mb;
  const __bf16 lb = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, hb);
  m = __builtin_bit_cast(unsigned short, mb);
  l = __builtin_bit_cast(unsigned short, lb);
}

template <int WK, int WN, int BK, int BN>
__global__ void __launch_bounds__(WG_THREADS, 2) k_wgrad6(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                          int ldb, int K, int N, long long P, float* __restrict__ partial,
                                                          int S, int tiles_k, int tiles_n, long long rows_per_split,
                                                          int k_off, int n_off, int k_end, int n_end) {
  static_assert(WK * WN == 4, "4 waves");
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  constexpr int PA = TK / 8, PB = TN / 8;          // points per thread and chunk (multiples of 4)
  extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
  unsigned short* As = wg_lds;                      // [3][TK][WG_CP]
  unsigned short* Bs = wg_lds + 3 * TK * WG_CP;     // [3][TN][WG_CP]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave / WN, wn = wave % WN;
  const int T = tiles_k * tiles_n;
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int vid = (nblk & 7) == 0 ? xcd * (nblk >> 3) + local : blockIdx.x;
  const int split = vid / T, tile = vid - split * T;
  const int tk = tile / tiles_n, tn = tile - tk * tiles_n;
  const int k0 = k_off + tk * TK, n0 = n_off + tn * TN;
  const long long p_begin = (long long)split * rows_per_split;
  long long p_end = p_begin + rows_per_split;
  if (p_end > P) p_end = P;

  f32x16 acc[BK][BN] = {};
  float ra[PA], rb[PB];
  const int fa = tid % TK, ga = tid / TK, fb = tid % TN, gb = tid / TN;   // feature column, point group
  const bool acol = (k0 + fa) < k_end, bcol = (n0 + fb) < n_end;
  const float* Ap = A + (long long)(ga * PA) * lda + k0 + fa;
  const float* Bp = B + (long long)(gb * PB) * ldb + n0 + fb;
  auto load_chunk = [&](long long p0) {
    const float* ap = Ap + p0 * lda;
    const float* bp = Bp + p0 * ldb;
    if (p0 + WG_C <= p_end) {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = acol ? ap[(long long)i * lda] : 0.f;
#pragma unroll
      for (int i = 0; i < PB; ++i) rb[i] = bcol ? bp[(long long)i * ldb] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = (acol && p0 + ga * PA + i < p_end) ? ap[(long long)i * lda] : 0.f;
#pragma unroll
      for (int i = 0; i < PB; ++i) rb[i] = (bcol && p0 + gb * PB + i < p_end) ? bp[(long long)i * ldb] : 0.f;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int g4 = 0; g4 < PA / 4; ++g4) {
      wg_u16x4 ph, pm, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) { unsigned short x, y, z; wg_split3(ra[4 * g4 + q], x, y, z); ph[q] = x; pm[q] = y; pl[q] = z; }
      unsigned short* d = As + fa * WG_CP + ga * PA + 4 * g4;
      *reinterpret_cast<wg_u16x4*>(d) = ph;
      *reinterpret_cast<wg_u16x4*>(d + TK * WG_CP) = pm;
      *reinterpret_cast<wg_u16x4*>(d + 2 * TK * WG_CP) = pl;
    }
#pragma unroll
    for (int g4 = 0; g4 < PB / 4; ++g4) {
      wg_u16x4 ph, pm, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) { unsigned short x, y, z; wg_split3(rb[4 * g4 + q], x, y, z); ph[q] = x; pm[q] = y; pl[q] = z; }
      unsigned short* d = Bs + fb * WG_CP + gb * PB + 4 * g4;
      *reinterpret_cast<wg_u16x4*>(d) = ph;
      *reinterpret_cast<wg_u16x4*>(d + TN * WG_CP) = pm;
      *reinterpret_cast<wg_u16x4*>(d + 2 * TN * WG_CP) = pl;
    }
  };

  if (p_begin < p_end) {
    load_chunk(p_begin);
    store_chunk();
    __syncthreads();
    for (long long p0 = p_begin; p0 < p_end; p0 += WG_C) {
      const bool more = p0 + WG_C < p_end;
      if (more) load_chunk(p0 + WG_C);
#pragma unroll
      for (int s = 0; s < WG_C / 16; ++s) {
        wg_bf16x8 av[BK][3], bv[BN][3];
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            av[i][p] = *reinterpret_cast<const wg_bf16x8*>(As + p * TK * WG_CP + ((wk * BK + i) * 32 + r) * WG_CP + 16 * s + 8 * h);
#pragma unroll
        for (int j = 0; j < BN; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            bv[j][p] = *reinterpret_cast<const wg_bf16x8*>(Bs + p * TN * WG_CP + ((wn * BN + j) * 32 + r) * WG_CP + 16 * s + 8 * h);
        // six partial products (planes 0 = hi, 1 = mid, 2 = lo), small terms first
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][2], bv[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][2], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][1], bv[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BK; ++i)
#pragma unroll
          for (int j = 0; j < BN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i][0], bv[j][0], acc[i][j], 0, 0, 0);
      }
      __syncthreads();                 // every wave is done with this chunk's planes
      if (more) store_chunk();
      __syncthreads();
    }
  }

  float* out = partial + (long long)split * K * N;
#pragma unroll
  for (int bi = 0; bi < BK; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj) {
      const int n = n0 + (wn * BN + bj) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + (wk * BK + bi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < k_end && n < n_end) out[(long long)k * N + n] = acc[bi][bj][i];
      }
    }
}

template <int WK, int WN, int BK, int BN>
static inline void wgrad6_go(int blocks, hipStream_t stream, const float* A, int lda, const float* B, int ldb, int K, int N,
                             long long P, float* ws, int S, int tk, int tn, long long rows, int k_off, int n_off, int k_end,
                             int n_end) {
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  const size_t lds = (size_t)3 * (TK + TN) * WG_CP * sizeof(unsigned short);
  hipLaunchKernelGGL((k_wgrad6<WK, WN, BK, BN>), dim3(blocks), dim3(WG_THREADS), lds, stream, A, lda, B, ldb, K, N, P, ws, S, tk,
                     tn, rows, k_off, n_off, k_end, n_end);
}

// ---- the same tiles on the f16 matrix cores: two-way split, three partial products (see mlp3.hip) ----
// A and B are scaled by per-tensor powers of two taken from their largest finite magnitudes (amax_a / amax_b: bit
// patterns of max |.|, produced by the chain kernels that wrote the tensors, or by k_absmax2 below) and split into
// hi = f16(x s), lo = f16((x s - hi) 2^11); acc0 += hi hi', acc1 += hi lo' + lo hi', dW = (acc0 + acc1 2^-11) / (s s').
// LDS planes [plane][feature][point] as in k_wgrad6, two planes instead of three.
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wg_f16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ void wg_scale_from_max(unsigned mbits, float& s, float& inv) {
  int E = (int)(mbits >> 23);
  if (E < 1) E = 1;
  int se = 268 - E;
  if (se > 253) se = 253;
  if (se < 1) se = 1;
  union { int i; float f; } a, b;
  a.i = se << 23;
  b.i = (254 - se) << 23;
  s = a.f;
  inv = b.f;
}

#ifdef WGG_TIMELINE      // tools/wgrad_timeline.py: s_memtime stamps of the chunk loop's phases (wave 0 of two workgroups)
__device__ long long wgg_stamps[2][64][12];
#define WGG_STAMP(i)                                                                                      \
  do {                                                                                                    \
    const int ck_ = (int)((p0 - p_begin) / WG_C);                                                        \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 1200) && ck_ < 64) {                       \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      wgg_stamps[blockIdx.x != 0][ck_][i] = __builtin_readcyclecounter();                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
    }                                                                                                     \
  } while (0)
#else
#define WGG_STAMP(i)
#endif

// One output tile of dW over the points [p_begin, p_end): `out` = this split's (K x N) partial slab.
template <int WK, int WN, int BK, int BN, int LAY = -1>
__device__ __forceinline__ void wgrad3_tile(const float* __restrict__ A_, int lda, const float* __restrict__ B_, int ldb, int N,
                                            long long p_begin, long long p_end, float* __restrict__ out_, int k0, int n0,
                                            int k_end, int n_end, float sa, float ia, float sb, float ib,
                                            unsigned short* wg_lds, int lay_rt = 0) {
  const int lay = LAY >= 0 ? LAY : lay_rt;          // (the main tile is instantiated per layout: one straight chunk loop each)
  const WG_G float* A = wg_global(A_);
  const WG_G float* B = wg_global(B_);
  WG_G float* out = wg_global(out_);
  // lay: bit 0 = A is point-blocked, bit 1 = B is (ChainArgs::side_blocked: element (p, f) at ((p >> 5) * ld + f) * 32 + (p & 31)).
  // A 32-point chunk of TK features of a blocked operand is ONE contiguous run of TK * 128 bytes: thread t takes the
  // 16-byte groups t, t + NT, ... of it (4 consecutive points of feature (group >> 3)) -- fully coalesced 16-byte loads, and
  // exactly the 4-point groups the planes are written in.  (Row-major operands: a thread owns a feature column and reads
  // its points one dword at a time.)  Blocked operands have P % 32 == 0 and splits of whole chunks: no ragged chunk.
  constexpr int NT = WK * WN * 64;                  // threads: one wave per (WK, WN) position
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  constexpr int PA = WG_C * TK / NT, PB = WG_C * TN / NT;          // points per thread and chunk (multiples of 4)
  static_assert(PA % 4 == 0 && PB % 4 == 0 && NT % TK == 0 && NT % TN == 0, "loader shape");
  _Float16* As = reinterpret_cast<_Float16*>(wg_lds);      // [2][TK][WG_CP]
  _Float16* Bs = As + 2 * TK * WG_CP;                      // [2][TN][WG_CP]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave / WN, wn = wave % WN;

  f32x16 acc0[BK][BN] = {}, acc1[BK][BN] = {};
  // (one register set: the next chunk's loads are in flight while this one is multiplied.  A second set spills at this
  //  tile's 128 accumulator registers and ran 3 x slower -- the pipelined tile below gets its second set from a single accumulator)
  float ra[1][PA], rb[1][PB];
  const int fa = tid % TK, ga = tid / TK, fb = tid % TN, gb = tid / TN;   // feature column, point group
  const bool acol = (k0 + fa) < k_end, bcol = (n0 + fb) < n_end;
  const WG_G float* Ap = A + (long long)(ga * PA) * lda + k0 + fa;
  const WG_G float* Bp = B + (long long)(gb * PB) * ldb + n0 + fb;
  const bool blk_a = lay & 1, blk_b = lay & 2;
  auto load_chunk = [&](long long p0, const int set) {
    const bool whole = p0 + WG_C <= p_end;
    if (blk_a) {
      const WG_G wg_f32x4* ap = reinterpret_cast<const WG_G wg_f32x4*>(A + ((p0 >> 5) * lda + k0) * 32) + tid;
#pragma unroll
      for (int i = 0; i < PA / 4; ++i) {
        wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k0 + ((i * NT + tid) >> 3) < k_end) v = ap[i * NT];
        ra[set][4 * i] = v[0]; ra[set][4 * i + 1] = v[1]; ra[set][4 * i + 2] = v[2]; ra[set][4 * i + 3] = v[3];
      }
    } else {
      const WG_G float* ap = Ap + p0 * lda;
      if (whole) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[set][i] = acol ? ap[(long long)i * lda] : 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[set][i] = (acol && p0 + ga * PA + i < p_end) ? ap[(long long)i * lda] : 0.f;
      }
    }
    if (blk_b) {
      const WG_G wg_f32x4* bp = reinterpret_cast<const WG_G wg_f32x4*>(B + ((p0 >> 5) * ldb + n0) * 32) + tid;
#pragma unroll
      for (int i = 0; i < PB / 4; ++i) {
        wg_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n0 + ((i * NT + tid) >> 3) < n_end) v = bp[i * NT];
        rb[set][4 * i] = v[0]; rb[set][4 * i + 1] = v[1]; rb[set][4 * i + 2] = v[2]; rb[set][4 * i + 3] = v[3];
      }
    } else {
      const WG_G float* bp = Bp + p0 * ldb;
      if (whole) {
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[set][i] = bcol ? bp[(long long)i * ldb] : 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[set][i] = (bcol && p0 + gb * PB + i < p_end) ? bp[(long long)i * ldb] : 0.f;
      }
    }
  };
  // hi = f16(x s), lo = f16((x s - hi) 2^11).  Three mixed-precision FMAs per element (v_fma_mix*: fp32 or f16 sources, fp32
  // arithmetic, one rounding, the f16 result written to its half of the packed pair): hi = f16(fma(x, s, 0)); r = fma(x, -s, hi)
  // = -(x s - hi), exact; lo = f16(fma(r, -2^11, 0)).  The same values as scale / convert / convert back / subtract / scale /
  // convert / pack (7 instructions per element: 896 of a chunk's ~4800 cycles per wave were this split -- a wave64 vector
  // instruction occupies its SIMD for 4 cycles, and the timeline shows the matrix and vector phases adding, not overlapping).
  auto split4 = [&](const float* v, float s, _Float16* d, int plane_stride) {
#ifdef NDJIR_WGRAD_SPLIT7
    wg_f32x4 xs = {v[0] * s, v[1] * s, v[2] * s, v[3] * s};
    const wg_f16x4 ph = __builtin_convertvector(xs, wg_f16x4);
    const wg_f32x4 res = (xs - __builtin_convertvector(ph, wg_f32x4)) * 2048.f;
    const wg_f16x4 pl = __builtin_convertvector(res, wg_f16x4);
    *reinterpret_cast<wg_f16x4*>(d) = ph;
    *reinterpret_cast<wg_f16x4*>(d + plane_stride) = pl;
#else
    typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));
    const float ms = -s, m2k = -2048.f;
    wg_u32x2 h, l;
    float r0, r1, r2, r3;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[0]) : "v"(v[0]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[0]) : "v"(v[1]), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[1]) : "v"(v[2]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[1]) : "v"(v[3]), "v"(s));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v[0]), "v"(ms), "v"(h[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v[1]), "v"(ms), "v"(h[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(v[2]), "v"(ms), "v"(h[1]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r3) : "v"(v[3]), "v"(ms), "v"(h[1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l[0]) : "v"(r0), "v"(m2k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l[0]) : "v"(r1), "v"(m2k));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l[1]) : "v"(r2), "v"(m2k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l[1]) : "v"(r3), "v"(m2k));
    *reinterpret_cast<wg_u32x2*>(d) = h;
    *reinterpret_cast<wg_u32x2*>(d + plane_stride) = l;
#endif
  };
  auto store_chunk = [&](const int set) {
    if (blk_a) {       // group e = i * NT + tid: feature e >> 3, points 4 (e & 7) ..
#pragma unroll
      for (int g4 = 0; g4 < PA / 4; ++g4) split4(ra[set] + 4 * g4, sa, As + ((g4 * NT + tid) >> 3) * WG_CP + 4 * (tid & 7), TK * WG_CP);
    } else {
#pragma unroll
      for (int g4 = 0; g4 < PA / 4; ++g4) split4(ra[set] + 4 * g4, sa, As + fa * WG_CP + ga * PA + 4 * g4, TK * WG_CP);
    }
    if (blk_b) {
#pragma unroll
      for (int g4 = 0; g4 < PB / 4; ++g4) split4(rb[set] + 4 * g4, sb, Bs + ((g4 * NT + tid) >> 3) * WG_CP + 4 * (tid & 7), TN * WG_CP);
    } else {
#pragma unroll
      for (int g4 = 0; g4 < PB / 4; ++g4) split4(rb[set] + 4 * g4, sb, Bs + fb * WG_CP + gb * PB + 4 * g4, TN * WG_CP);
    }
  };
  auto multiply_chunk = [&]() {
#pragma unroll
    for (int s = 0; s < WG_C / 16; ++s) {
      wg_f16x8 av[BK][2], bv[BN][2];
#pragma unroll
      for (int i = 0; i < BK; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          av[i][p] = *reinterpret_cast<const wg_f16x8*>(As + p * TK * WG_CP + ((wk * BK + i) * 32 + r) * WG_CP + 16 * s + 8 * h);
#pragma unroll
      for (int j = 0; j < BN; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bv[j][p] = *reinterpret_cast<const wg_f16x8*>(Bs + p * TN * WG_CP + ((wn * BN + j) * 32 + r) * WG_CP + 16 * s + 8 * h);
#pragma unroll
      for (int i = 0; i < BK; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][1], bv[j][0], acc1[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < BK; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j) acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][0], bv[j][0], acc0[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < BK; ++i)
#pragma unroll
        for (int j = 0; j < BN; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][0], bv[j][1], acc1[i][j], 0, 0, 0);
    }
  };

  if (p_begin < p_end) {
    load_chunk(p_begin, 0);
    store_chunk(0);
    __syncthreads();
    for (long long p0 = p_begin;;) {
      const bool more = p0 + WG_C < p_end;
      WGG_STAMP(0);
      if (more) load_chunk(p0 + WG_C, 0);
      WGG_STAMP(1);
      multiply_chunk();
      WGG_STAMP(2);
      __syncthreads();                 // every wave is done with this chunk's planes
      WGG_STAMP(3);
      if (!more) break;
#ifdef WGG_TIMELINE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WGG_STAMP(4);
#endif
      store_chunk(0);
      WGG_STAMP(5);
      __syncthreads();
      WGG_STAMP(6);
      p0 += WG_C;
    }
  }

#pragma unroll
  for (int bi = 0; bi < BK; ++bi)
#pragma unroll
    for (int bj = 0; bj < BN; ++bj) {
      const int n = n0 + (wn * BN + bj) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + (wk * BK + bi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (k < k_end && n < n_end) out[(long long)k * N + n] = fmaf(acc1[bi][bj][i], 1.f / 2048.f, acc0[bi][bj][i]) * ia * ib;
      }
    }
}

// ---- the 128 x 128 main tile, software-pipelined (round 5) --------------------------------------------------------------------
// What the chunk loop above costs a wave per 32-point chunk (tools/wgrad_timeline.py, 256 x 256 x 65536, both operands
// blocked): issue loads 1270 cycles, 24 MFMAs + their LDS reads 1150, barrier 160, wait for the loads 500, split + LDS writes
// 1340, barrier 170 -- 4800 cycles for 768 cycles of matrix work, and tools/ubench/coissue.hip shows why two workgroups per CU
// do not hide it: on gfx950 the matrix pipe and the vector ALU of a SIMD do NOT overlap across waves (4 MFMA waves + 4 FMA
// waves take the SUM of their solo times, whatever the priorities), while ONE wave that puts up to 4 vector instructions
// behind each 32x32x16 MFMA gets them for free.  So the phases must interleave inside the wave:
//   * LDS planes double-buffered (2 x 40 KB): while chunk c is multiplied out of one buffer, chunk c + 1 is split and written
//     into the other -- one barrier per chunk, and the split's vector instructions sit in the MFMAs' shadows;
//   * two register sets of raw operands: chunk c + 1 (being split) and chunk c + 2 (in flight); the moment a 4-point group of
//     chunk c + 1 is written, its registers take the load of the same group of chunk c + 3 -- every load has two chunk times
//     to arrive, and vmcnt (in order) never waits for a younger load;
//   * ONE accumulator per block: lo = f16(x s - hi) unscaled, so hi hi' + hi lo' + lo hi' accumulate together (the matrix
//     cores keep f16 denormals, tools/ubench/mfma_denorm.hip: x s is represented to 2^-22 relative or 2^-25 absolute at
//     max |x s| in [2^14, 2^15), i.e. 2^-39 of the operand's largest element) -- 64 registers instead of 128, which is what
//     makes room for the second operand set and a second fragment set (the next k-step's LDS reads under this one's MFMAs).
// Loads past the item's last chunk re-read that chunk (L2 hits), its planes land in the buffer nobody reads.
#define WGP_FS(s) (F2 ? (s) : 0)
constexpr int WGP_CP = 32;                 // plane row pitch of the pipelined tile, in halves: 32 points, no padding
constexpr int WGP_LDS = 2 * 2 * (WG_T + WG_T) * WGP_CP * 2;      // bytes: two buffers of [A hi][A lo][B hi][B lo]
constexpr int WGP_LDS_WIDE = 2 * 2 * (WG_T + 2 * WG_T) * WGP_CP * 2;      // the 128 x 256 tile's
// A feature's 32 points = four 16-byte segments; segment q of feature f is stored at q ^ ((f >> 2) & 3): the 16 lanes of a
// fragment read's pass (16 consecutive features, one segment) then cover all 64 banks, as do the 8-byte writes of a group.
__device__ __forceinline__ int wgp_at(int f, int seg) { return f * WGP_CP + ((seg ^ ((f >> 2) & 3)) << 3); }

// NBJ = 32-column blocks per wave: 2 = the 128 x 128 tile (wave tile 64 x 64, two workgroups per CU), 4 = the 128 x 256 tile of
// k_wgrad_group_wide (wave tile 64 x 128, ONE workgroup per CU with 512 registers per lane).  Why the wide tile: the 64 x 64
// wave tile reads 16 KB of fragments out of LDS per chunk for 24 MFMAs, and with the planes' writes a CU moves 192 KB through
// LDS per pair of chunks -- 1500 cycles at 128 B / clk against 1536 cycles of matrix work: the loop is LDS-bound (the variant
// without MFMAs and without global loads still takes 20 of the 39 us per 256 x 256 x 65536 layer).  A 64 x 128 wave tile reads
// 24 KB for 48 MFMAs, and the chunk's vector work (132 instructions) fits the MFMAs' shadows.
template <int LAY, bool FULL, int NBJ>
__device__ __forceinline__ void wgrad3_pipe(const float* __restrict__ A_, int lda, const float* __restrict__ B_, int ldb, int N,
                                            long long p_begin, long long p_end, float* __restrict__ out_, int k0, int n0,
                                            int k_end, int n_end, float sa, float ia, float sb, float ib,
                                            unsigned short* wg_lds) {
  // FULL: the tile lies inside the K x N matrix (no feature masks).
  constexpr int NT = 256, TK = WG_T, TN = 64 * NBJ;
  constexpr int GA = 4, GB = 2 * NBJ;                     // 4-point groups per thread and chunk
  constexpr int NSIDE = GA + GB, NMF = 6 * NBJ;           // side items per chunk; MFMAs per k-step (2 x NBJ blocks x 3 products)
  constexpr int MPS = NMF / (NSIDE / 2);                  // MFMAs in front of each side item (3 | 4)
  static_assert(MPS * (NSIDE / 2) == NMF && (NSIDE % 2) == 0, "MFMAs per side item");
  constexpr bool BLA = (LAY & 1) != 0, BLB = (LAY & 2) != 0;
  constexpr int PLA = TK * WGP_CP, PLB = TN * WGP_CP;     // halves per plane
  constexpr int BUF = 2 * PLA + 2 * PLB;                  // halves per buffer: [A hi][A lo][B hi][B lo]
  WG_G float* out = wg_global(out_);
  _Float16* L = reinterpret_cast<_Float16*>(wg_lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wk = wave >> 1, wn = wave & 1;

  f32x16 acc[2][NBJ] = {};
  wg_f32x4 ra[2][GA], rb[2][GB];                          // [set][group]: 4 consecutive points of one feature

  // Group g of a thread = 4 consecutive points (8-byte slot j of the feature's row: segment j >> 1, half j & 1).
  // Blocked operand: 16-byte run e = g * NT + tid of the chunk -- feature g * 32 + (tid >> 3), slot tid & 7;
  // row-major: feature tid & 127, slot (tid >> 7) * 4 + g.  A feature beyond the operand's width reads the last valid one and
  // is multiplied by a scale of 0.  Buffer loads: the descriptor's base is the item's first row at the tile's first feature,
  // the chunk moves in the SCALAR offset, the thread's place in the chunk is one 32-bit vector register per group and
  // operand -- no vector address arithmetic in the loop.
  const int ka = k_end - k0 - 1, nb = n_end - n0 - 1;     // last valid feature of the tile (>= 0)
  // row-major operand of T features: feature tid % T, the thread's 32 * T / 256 points start at (tid / T) * that
  const int fa_r = tid & (TK - 1), fb_r = tid & (TN - 1), f_b = tid >> 3, j_b = tid & 7;
  const int pta_r = (tid / TK) * (4 * GA), ptb_r = (tid / TN) * (4 * GB);
  // LDS slot of group g inside a plane: blocked = lds_blk + g * 32 rows; row-major = slot (pt / 4 + g) of the feature's row
  const int lds_blk = wgp_at(f_b, j_b >> 1) + (j_b & 1) * 4;
  auto lds_a = [&](int g) {
    const int j = pta_r / 4 + g;
    return BLA ? lds_blk + g * 32 * WGP_CP : wgp_at(fa_r, j >> 1) + (j & 1) * 4;
  };
  auto lds_b = [&](int g) {
    const int j = ptb_r / 4 + g;
    return BLB ? lds_blk + g * 32 * WGP_CP : wgp_at(fb_r, j >> 1) + (j & 1) * 4;
  };
  unsigned oa[GA], ob[GB];                                // byte offset of the group's first element inside a chunk
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int fga = BLA ? g * 32 + f_b : fa_r;
    const int fca = FULL ? fga : (fga < ka ? fga : ka);
    oa[g] = BLA ? (unsigned)((fca * 32 + 4 * j_b) * 4) : (unsigned)(((pta_r + 4 * g) * lda + fca) * 4);
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int fgb = BLB ? g * 32 + f_b : fb_r;
    const int fcb = FULL ? fgb : (fgb < nb ? fgb : nb);
    ob[g] = BLB ? (unsigned)((fcb * 32 + 4 * j_b) * 4) : (unsigned)(((ptb_r + 4 * g) * ldb + fcb) * 4);
  }
  auto scale_a = [&](int g) { return FULL || (BLA ? g * 32 + f_b : fa_r) <= ka ? sa : 0.f; };
  auto scale_b = [&](int g) { return FULL || (BLB ? g * 32 + f_b : fb_r) <= nb ? sb : 0.f; };
  auto uniform_ptr = [](const float* q) {      // (the item's operands are the same for every lane: keep the descriptors scalar)
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(BLA ? A_ + ((p_begin >> 5) * lda + k0) * 32 : A_ + p_begin * lda + k0), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(BLB ? B_ + ((p_begin >> 5) * ldb + n0) * 32 : B_ + p_begin * ldb + n0), 0, 0x7fffffff, 0x00020000);
  const int n_rows = (int)(p_end - p_begin);
  const int c_last = (n_rows - 1) / WG_C;                 // the item's last chunk
  const int rag = n_rows - c_last * WG_C;                 // its rows (32: none missing)
  const int cha = lda * 4 * WG_C, chb = ldb * 4 * WG_C;   // bytes per chunk (either layout: 32 rows of lda floats)
  auto at = [&](int c) { return c < c_last ? c : c_last; };

  // (row-major, ragged last chunk: a row past the item's end re-reads the item's last row -- never memory past the operand --
  //  and the split zeroes it)
  auto load_rows = [&](wg_f32x4& dst, const __amdgpu_buffer_rsrc_t& rs, unsigned off, int ld, int pt0, int g, int c, int ch) {
    const int so = c * ch;
    if (rag < WG_C && c == c_last) {                      // (uniform)
      const int row0 = pt0 + 4 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int back = row0 + i - (rag - 1);            // rows past the last one
        dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + (unsigned)((i - (back > 0 ? back : 0)) * ld * 4), so, 0));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, so + i * ld * 4, 0));
    }
  };
  auto load_a = [&](const int set, const int g, int c) {
    if (BLA) ra[set][g] = __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, oa[g], c * cha, 0));
    else load_rows(ra[set][g], rsa, oa[g], lda, pta_r, g, c, cha);
  };
  auto load_b = [&](const int set, const int g, int c) {
    if (BLB) rb[set][g] = __builtin_bit_cast(wg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, ob[g], c * chb, 0));
    else load_rows(rb[set][g], rsb, ob[g], ldb, ptb_r, g, c, chb);
  };
  // hi = f16(x s), lo = f16(x s - hi): both planes of a 4-point group, 8 bytes each.  Mixed-precision FMAs (v_fma_mix*: fp32 or
  // f16 sources, fp32 arithmetic, one rounding): hi = f16(fma(x, s, 0)) written to its half of the packed pair, r = fma(x, s, -hi)
  // (exact), lo = cvt_pk(r): 10 vector instructions per group where multiply / convert / convert back / subtract / convert
  // takes 16 -- these do not hide under other waves' MFMAs (tools/ubench/coissue.hip), so they are chunk time.
  auto split_store = [&](wg_f32x4 v, float s, bool ragged, int pt0, _Float16* d, int plane) {
#pragma clang fp contract(off)
    if (ragged) {                        // (row-major operand, last chunk of the item: rows past p_end count as zeros)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = pt0 + i < rag ? v[i] : 0.f;
    }
#ifdef WGP_SPLIT16
    const wg_f32x4 xs = v * s;
    const wg_f16x4 ph = __builtin_convertvector(xs, wg_f16x4);
    const wg_f32x4 res = xs - __builtin_convertvector(ph, wg_f32x4);
    const wg_f16x4 pl = __builtin_convertvector(res, wg_f16x4);
    *reinterpret_cast<wg_f16x4*>(d) = ph;
    *reinterpret_cast<wg_f16x4*>(d + plane) = pl;
#else
    typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));
    typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
    wg_u32x2 hp;
    float r0, r1, r2, r3;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hp[0]) : "v"(v[0]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hp[0]) : "v"(v[1]), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hp[1]) : "v"(v[2]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hp[1]) : "v"(v[3]), "v"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v[0]), "v"(s), "v"(hp[0]));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v[1]), "v"(s), "v"(hp[0]));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(v[2]), "v"(s), "v"(hp[1]));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r3) : "v"(v[3]), "v"(s), "v"(hp[1]));
    const wg_f16x2 l01 = __builtin_convertvector(wg_f32x2{r0, r1}, wg_f16x2), l23 = __builtin_convertvector(wg_f32x2{r2, r3}, wg_f16x2);
    const wg_u32x2 lp = {__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
    *reinterpret_cast<wg_u32x2*>(d) = hp;
    *reinterpret_cast<wg_u32x2*>(d + plane) = lp;
#endif
  };
#ifdef WGP_NO_LOAD      // (experiment: the loop without its global loads -- the registers keep the first chunks)
#define WGP_LOAD(x)
#else
#define WGP_LOAD(x) x
#endif
  // side item i of an iteration: split + store group i of A (i < GA) or group i - GA of B of chunk `cn` out of `set` into
  // `buf`, then refill the group's registers with chunk `cr`
  auto side = [&](const int i, const int set, const int buf, int cn_, int cr) {
    // (a chunk past the item's end is split into planes of zeros: the chunk loop runs in pairs, and multiplying one chunk too
    //  many must add nothing)
    const float live = cn_ <= c_last ? 1.f : 0.f;
    const int cn = at(cn_);
    const bool ragged = rag < WG_C && cn == c_last;
    if (i < GA) {
      const int g = i;
      split_store(ra[set][g], scale_a(g) * live, !BLA && ragged, pta_r + 4 * g, L + buf * BUF + lds_a(g), PLA);
      WGP_LOAD(load_a(set, g, cr));
    } else {
      const int g = i - GA;
      split_store(rb[set][g], scale_b(g) * live, !BLB && ragged, ptb_r + 4 * g, L + buf * BUF + 2 * PLA + lds_b(g), PLB);
      WGP_LOAD(load_b(set, g, cr));
    }
  };

  // fragment of block row i, plane pl, k-step s: feature (w * 2 + i) * 32 + r, segment 2 s + h
  const int fr_a[2] = {wgp_at((wk * 2) * 32 + r, h), wgp_at((wk * 2) * 32 + r, 2 + h)};      // [k-step]; + i * 32 rows, + plane
  const int fr_b[2] = {wgp_at((wn * NBJ) * 32 + r, h), wgp_at((wn * NBJ) * 32 + r, 2 + h)};
  // one k-step's fragments at a time where a second wave of the SIMD covers the LDS latency (128 x 128 tile: a second set
  // measured 39.8 against 39.1 us); the one-wave-per-SIMD 128 x 256 item reads the second k-step's under the first's MFMAs
#ifdef WGP_FRAG2_OFF
  constexpr bool F2 = false;
#else
  constexpr bool F2 = NBJ == 4;
#endif
  wg_f16x8 fa[F2 ? 2 : 1][2][2], fb[F2 ? 2 : 1][NBJ][2];  // [k-step][block][plane]
  auto frags = [&](const int s_, const int buf) {
    const int s = s_;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[WGP_FS(s)][i][pl] = *reinterpret_cast<const wg_f16x8*>(L + buf * BUF + pl * PLA + i * 32 * WGP_CP + fr_a[s]);
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        fb[WGP_FS(s)][j][pl] = *reinterpret_cast<const wg_f16x8*>(L + buf * BUF + 2 * PLA + pl * PLB + j * 32 * WGP_CP + fr_b[s]);
    }
  };
  // MFMA q (0 .. NMF - 1) of k-step s: product q / (2 NBJ) (lo hi', hi lo', hi hi') of block q % (2 NBJ) -- two MFMAs into
  // one accumulator are 2 NBJ apart
  auto mfma1 = [&](const int s, const int q) {
    const int pr = q / (2 * NBJ), bl = q % (2 * NBJ), i = bl / NBJ, j = bl % NBJ;
#ifdef WGP_NO_MFMA      // (experiment: what the loop costs without its matrix work)
    acc[i][j][0] += (float)fa[WGP_FS(s)][i][pr == 0][0] + (float)fb[WGP_FS(s)][j][pr == 1][0];
    return;
#endif
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[WGP_FS(s)][i][pr == 0 ? 1 : 0], fb[WGP_FS(s)][j][pr == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
  };
  auto mfmas = [&](const int s, const int m) {
#pragma unroll
    for (int q = 0; q < MPS; ++q) mfma1(s, MPS * m + q);
  };
  // one chunk: multiply `buf`, split chunk cn (set `set`) into the other buffer, refill `set` with chunk cr
#ifdef WGG_TIMELINE
  int tl_c = 0;
#define WGP_STAMP(i)                                                                                       \
  do {                                                                                                     \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 1200) && tl_c < 64) {                       \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      wgg_stamps[blockIdx.x != 0][tl_c][i] = (i) == 11 ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_readcyclecounter(); \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
    }                                                                                                      \
  } while (0)
#else
#define WGP_STAMP(i)
#endif
  // (with the compiler's split -- plain vector instructions the scheduler can classify -- ask for one MFMA, then a slice of the
  //  side item's vector work, MPS times: the in-order front end issues nothing behind an MFMA that waits for the matrix pipe)
#if defined(WGP_SPLIT16) && !defined(WGP_NO_INTERLEAVE)
#define WGP_INTERLEAVE()                                               \
  do {                                                                 \
    for (int q_ = 0; q_ < MPS; ++q_) {                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
      __builtin_amdgcn_sched_group_barrier(0x002, 16 / MPS, 0);        \
    }                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                 \
  } while (0)
#else
#define WGP_INTERLEAVE()
#endif
  auto iteration = [&](const int buf, const int set, int cn, int cr) {
    WGP_STAMP(0);
    if (!F2) frags(0, buf);              // (F2: loaded by the previous iteration's tail / the prologue)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NSIDE / 2; ++m) {
      mfmas(0, m);
      if (F2 && m == 0) frags(1, buf);
      side(m, set, buf ^ 1, cn, cr);
      WGP_INTERLEAVE();
      __builtin_amdgcn_sched_barrier(0);
      if (m < 4) WGP_STAMP(1 + m);
    }
    if (!F2) {
      frags(1, buf);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NSIDE / 2; ++m) {
        mfmas(1, m);
        side(NSIDE / 2 + m, set, buf ^ 1, cn, cr);
        WGP_INTERLEAVE();
        __builtin_amdgcn_sched_barrier(0);
        if (m < 4) WGP_STAMP(5 + m);
      }
      __syncthreads();
    } else {
      // One wave per SIMD: nobody covers the LDS latency of the next chunk's first fragments behind the barrier.  The second
      // k-step's side items go behind its first NMF / 2 MFMAs, then the barrier, then the NEXT chunk's first fragments are
      // requested (their registers are free: the first k-step's MFMAs are issued) and arrive under the last NMF / 2 MFMAs.
      constexpr int HM = NMF / 2 / (NSIDE / 2);
#pragma unroll
      for (int m = 0; m < NSIDE / 2; ++m) {
#pragma unroll
        for (int q = 0; q < HM; ++q) mfma1(1, HM * m + q);
        side(NSIDE / 2 + m, set, buf ^ 1, cn, cr);
        __builtin_amdgcn_sched_barrier(0);
        if (m < 4) WGP_STAMP(5 + m);
      }
      __syncthreads();
      frags(0, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = NMF / 2; q < NMF; ++q) mfma1(1, q);
      __builtin_amdgcn_sched_barrier(0);
    }
    WGP_STAMP(9);
    WGP_STAMP(11);
#ifdef WGG_TIMELINE
    ++tl_c;
#endif
  };

  if (n_rows > 0) {
    // (issue order = the order the iterations consume and refill in: A groups, then B groups, set by set)
#pragma unroll
    for (int g = 0; g < GA; ++g) load_a(0, g, 0);
#pragma unroll
    for (int g = 0; g < GB; ++g) load_b(0, g, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < GA; ++g) load_a(1, g, at(1));
#pragma unroll
    for (int g = 0; g < GB; ++g) load_b(1, g, at(1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NSIDE; ++i) {
      side(i, 0, 0, 0, at(2));
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (F2) frags(0, 0);
    for (int c = 0; c <= c_last; c += 2) {               // (pairs: one exit, the accumulators stay where they are)
      iteration(0, 1, c + 1, at(c + 3));
      iteration(1, 0, c + 2, at(c + 4));
    }
  }

  const float de = ia * ib;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int bj = 0; bj < NBJ; ++bj) {
      const int n = n0 + (wn * NBJ + bj) * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + (wk * 2 + bi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (FULL || (k < k_end && n < n_end)) out[(long long)k * N + n] = acc[bi][bj][i] * de;
      }
    }
}

template <int WK, int WN, int BK, int BN>
__global__ void __launch_bounds__(WG_THREADS, 2) k_wgrad3(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                          int ldb, int K, int N, long long P, float* __restrict__ partial,
                                                          int S, int tiles_k, int tiles_n, long long rows_per_split,
                                                          int k_off, int n_off, int k_end, int n_end,
                                                          const unsigned* __restrict__ amax_a, const unsigned* __restrict__ amax_b) {
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
  const int T = tiles_k * tiles_n;
  const int nblk = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int vid = (nblk & 7) == 0 ? xcd * (nblk >> 3) + local : blockIdx.x;
  const int split = vid / T, tile = vid - split * T;
  const int tk = tile / tiles_n, tn = tile - tk * tiles_n;
  const int k0 = k_off + tk * TK, n0 = n_off + tn * TN;
  const long long p_begin = (long long)split * rows_per_split;
  long long p_end = p_begin + rows_per_split;
  if (p_end > P) p_end = P;
  float sa, ia, sb, ib;
  wg_scale_from_max(*amax_a, sa, ia);
  wg_scale_from_max(*amax_b, sb, ib);
  wgrad3_tile<WK, WN, BK, BN>(A, lda, B, ldb, N, p_begin, p_end, partial + (long long)split * K * N, k0, n0, k_end, n_end, sa, ia,
                              sb, ib, wg_lds);
}

template <int WK, int WN, int BK, int BN>
static inline void wgrad3_go(int blocks, hipStream_t stream, const float* A, int lda, const float* B, int ldb, int K, int N,
                             long long P, float* ws, int S, int tk, int tn, long long rows, int k_off, int n_off, int k_end,
                             int n_end, const unsigned* amax_a, const unsigned* amax_b) {
  constexpr int TK = WK * BK * 32, TN = WN * BN * 32;
  const size_t lds = (size_t)2 * (TK + TN) * WG_CP * sizeof(unsigned short);
  hipLaunchKernelGGL((k_wgrad3<WK, WN, BK, BN>), dim3(blocks), dim3(WG_THREADS), lds, stream, A, lda, B, ldb, K, N, P, ws, S, tk,
                     tn, rows, k_off, n_off, k_end, n_end, amax_a, amax_b);
}

// largest finite |.| of two row-strided matrices -> out[0], out[1] (bit patterns; zeroed by the caller): the scales
// of k_wgrad3 when the producer of a tensor did not record its maximum
__global__ void __launch_bounds__(256) k_absmax2(const float* __restrict__ A, int lda, int K, const float* __restrict__ B, int ldb,
                                                 int N, long long P, unsigned* __restrict__ out, int need_a, int need_b) {
  __shared__ unsigned red[2][256];
  unsigned ma = 0, mb = 0;
  const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (need_a)
    for (long long t = t0; t < P * K; t += stride) {          // flat over (row, column): every thread has loads in flight
      const unsigned b = __float_as_uint(A[(t / K) * lda + (t % K)]) & 0x7fffffffu;
      if (b < 0x7f800000u && b > ma) ma = b;
    }
  if (need_b)
    for (long long t = t0; t < P * N; t += stride) {
      const unsigned b = __float_as_uint(B[(t / N) * ldb + (t % N)]) & 0x7fffffffu;
      if (b < 0x7f800000u && b > mb) mb = b;
    }
  red[0][threadIdx.x] = ma;
  red[1][threadIdx.x] = mb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      if (red[0][threadIdx.x + s] > red[0][threadIdx.x]) red[0][threadIdx.x] = red[0][threadIdx.x + s];
      if (red[1][threadIdx.x + s] > red[1][threadIdx.x]) red[1][threadIdx.x] = red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (need_a) atomicMax(out, red[0][0]);
    if (need_b) atomicMax(out + 1, red[1][0]);
  }
}

__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ out,
                                                      long long KN, int S, int accum) {
  // one float4 of the output per thread; 8 independent loads in flight per thread
  const long long n4 = KN >> 2;
  const bool vec = ((KN & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);      // (out may be a view into a flat gradient bucket at any 4-byte offset)
  if (vec) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* p = reinterpret_cast<const float4*>(partial) + i;
      int k = 0;
      for (; k + 8 <= S; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; k < S; ++k) { float4 v = p[(long long)k * n4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      float4* o = reinterpret_cast<float4*>(out) + i;
      if (accum) { float4 c = *o; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
      *o = s;
    }
  } else {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < KN; i += (long long)gridDim.x * blockDim.x) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < S; ++k) s += partial[(long long)k * KN + i];
      out[i] = accum ? out[i] + s : s;
    }
  }
}

// Region plan of a K x N weight gradient: 128 x 128 main tiles over [0, Km) x [0, Nm); remainders of
// at most 64 go to 32-wide strips (K strip spans all of N, N strip spans [0, Km)), larger ones to a
// further (ragged) main tile.
constexpr int WG_STRIP = 32;
struct WgradPlan {
  int Km, Nm;           // extent covered by main tiles (>= K / >= N when there is no strip)
  int tk, tn;           // main tiles
  int sk, sn;           // strips (of WG_STRIP) in K / in N
  int units;            // work in 128x128-tile equivalents x 16 (a 32x128 strip tile = 4)
};

static inline WgradPlan wgrad_plan(int K, int N) {
  WgradPlan p;
  const int rk = K % WG_T, rn = N % WG_T;
  p.Km = (rk != 0 && rk <= 2 * WG_STRIP) ? K - rk : K;
  p.Nm = (rn != 0 && rn <= 2 * WG_STRIP) ? N - rn : N;
  p.tk = (p.Km + WG_T - 1) / WG_T;
  p.tn = (p.Nm + WG_T - 1) / WG_T;
  p.sk = (K - p.Km + WG_STRIP - 1) / WG_STRIP;
  p.sn = (N - p.Nm + WG_STRIP - 1) / WG_STRIP;
  const int tn_all = (N + WG_T - 1) / WG_T;
  p.units = 16 * p.tk * p.tn + 4 * p.sk * tn_all + 4 * p.sn * p.tk;
  return p;
}

// number of point-axis splits.  fp32 engine: ~1 workgroup (4 waves) of main-tile work per CU keeps every
// SIMD's matrix pipe busy (its LDS is double-buffered).  bf16x6 engine: 2 per CU -- its load/convert/store
// phase is not overlapped inside a workgroup, a second resident workgroup fills it (measured 87 -> 72 us
// at 256 x 256 x 65536).
static inline int pick_splits(int K, int N, long long P, int target_blocks) {
  const WgradPlan pl = wgrad_plan(K, N);
  int T16 = pl.units < 16 ? 16 : pl.units;
  int S = (target_blocks * 16 + T16 - 1) / T16;
  long long max_s = (P + WG_C - 1) / WG_C;
  if (S > max_s) S = (int)max_s;
  if (S < 1) S = 1;
  while (S & 7) ++S;                               // whole splits per XCD whatever the tile count
  return S;
}


// ---- narrow outputs (N <= 8) and column sums: streaming reductions, no matrix cores ----------------
// The 128x128 MFMA tiles above would be > 90 % padding for the 1..6-wide output layers; these are
// bandwidth problems (read A once).  Threads (tx = column phase, ty = row phase); every workgroup
// writes one partial result, the split reduction kernel above sums the partials (no float atomics:
// device-scope atomics on one address from 8 XCDs serialise in the memory-side cache).
constexpr int SW_NMAX = 8;
constexpr int SW_ROWS = 128;          // rows per workgroup

__device__ __forceinline__ int pow2_at_least(int v) {
  int t = 1;
  while (t < v && t < 256) t <<= 1;
  return t;
}

// A is read as float4 (needs K % 4 == 0, lda % 4 == 0, 16-byte aligned base): thread = (k quad, row phase)
template <int NMAX>
__global__ void __launch_bounds__(256) k_wgrad_narrow(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                      int ldb, int K, int N, long long P, float* __restrict__ ws) {
  __shared__ float4 red[256];
  const int KQ = K >> 2;
  const int TX = pow2_at_least(KQ), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const long long p0 = (long long)blockIdx.x * SW_ROWS;
  const long long p1 = (p0 + SW_ROWS < P) ? p0 + SW_ROWS : P;
  float* part = ws + (long long)blockIdx.x * K * N;
  for (int q0 = 0; q0 < KQ; q0 += TX) {
    const int kq = q0 + tx;
    float4 acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kq < KQ) {
#pragma unroll 4
      for (long long p = p0 + ty; p < p1; p += TY) {
        const float4 a = *reinterpret_cast<const float4*>(A + p * lda + 4 * kq);
        const float* b = B + p * ldb;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) if (n < N) {
          const float bn = b[n];
          acc[n].x = fmaf(a.x, bn, acc[n].x); acc[n].y = fmaf(a.y, bn, acc[n].y);
          acc[n].z = fmaf(a.z, bn, acc[n].z); acc[n].w = fmaf(a.w, bn, acc[n].w);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {                                   // uniform
        red[threadIdx.x] = acc[n];
        __syncthreads();
        for (int s = TY / 2; s > 0; s >>= 1) {
          if (ty < s) {
            float4 o = red[threadIdx.x + s * TX], m = red[threadIdx.x];
            red[threadIdx.x] = make_float4(m.x + o.x, m.y + o.y, m.z + o.z, m.w + o.w);
          }
          __syncthreads();
        }
        if (ty == 0 && kq < KQ) {
          const float4 t = red[tx];
          float* o = part + (long long)(4 * kq) * N + n;
          o[0] = t.x; o[N] = t.y; o[2 * N] = t.z; o[3 * N] = t.w;
        }
        __syncthreads();
      }
    }
  }
}

// partial column sums of X (P x N): ws[block][n].  Narrow N: threads (column, row phase) + LDS tree;
// wide N (> 256): one thread per column phase, all of the row's column passes inside the row loop.
constexpr int CS_MAXPASS = 8;          // wide path: N <= 2048
__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ X, int ldx, int N, long long P,
                                                float* __restrict__ ws, int rows_per_block) {
  __shared__ float red[256];
  const long long p0 = (long long)blockIdx.x * rows_per_block;
  const long long p1 = (p0 + rows_per_block < P) ? p0 + rows_per_block : P;
  float* part = ws + (long long)blockIdx.x * N;
  if (N > 256) {
    float acc[CS_MAXPASS];
#pragma unroll
    for (int j = 0; j < CS_MAXPASS; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (long long p = p0; p < p1; ++p) {
      const float* x = X + p * ldx + threadIdx.x;
#pragma unroll
      for (int j = 0; j < CS_MAXPASS; ++j) if (j * 256 + (int)threadIdx.x < N) acc[j] += x[j * 256];
    }
#pragma unroll
    for (int j = 0; j < CS_MAXPASS; ++j) if (j * 256 + (int)threadIdx.x < N) part[j * 256 + threadIdx.x] = acc[j];
    return;
  }
  const int TX = pow2_at_least(N), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  float acc = 0.f;
  if (tx < N) {
#pragma unroll 8
    for (long long p = p0 + ty; p < p1; p += TY) acc += X[p * ldx + tx];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = TY / 2; s > 0; s >>= 1) {
    if (ty < s) red[threadIdx.x] += red[threadIdx.x + s * TX];
    __syncthreads();
  }
  if (ty == 0 && tx < N) part[tx] = red[tx];
}

// out[i] (+)= sum_s ws[s][i] for many partials (S > 32): 32 outputs x 8 split phases per workgroup
__global__ void __launch_bounds__(256) k_reduce_tall(const float* __restrict__ ws, float* __restrict__ out, long long KN,
                                                     int S, int accum) {
  __shared__ float red[256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (i < KN) {
#pragma unroll 8
    for (int sidx = ty; sidx < S; sidx += 8) acc += ws[(long long)sidx * KN + i];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (ty == 0 && i < KN) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q * 32 + tx];
    out[i] = accum ? out[i] + t : t;
  }
}

static void colsum_plan(int N, long long P, int* rows, int* blocks) {
  int TX = 1;
  while (TX < N && TX < 256) TX <<= 1;
  const int TY = 256 / TX;
  if (N > 256) TX = 256;
  int r = (N > 256) ? 64 : 32 * TY;                     // rows per workgroup
  long long b = (P + r - 1) / r;
  if (b > 1024) { r = (int)((P + 1023) / 1024); r = (r + TY - 1) / TY * TY; b = (P + r - 1) / r; }
  if (b < 1) b = 1;
  *rows = r; *blocks = (int)b;
}

long long colsum_workspace(int N, long long P) {
  int rows, blocks;
  colsum_plan(N, P, &rows, &blocks);
  return (long long)blocks * N;
}

static int launch_split_reduce(const float* ws, float* out, long long KN, int S, int accum, hipStream_t stream);

int launch_colsum(const float* X, int ldx, int N, long long P, float* out, int accum, float* workspace,
                  hipStream_t stream) {
  if (N <= 0) return NDJIR_OK;
  if (P <= 0) {
    if (!accum && hipMemsetAsync(out, 0, (size_t)N * sizeof(float), stream) != hipSuccess) return NDJIR_ERR_LAUNCH;
    return NDJIR_OK;
  }
  int rows, blocks;
  colsum_plan(N, P, &rows, &blocks);
  hipLaunchKernelGGL(k_colsum, dim3(blocks), dim3(256), 0, stream, X, ldx, N, P, workspace, rows);
  return launch_split_reduce(workspace, out, N, blocks, accum, stream);
}

// out[g][n] = sum over the `div` consecutive rows of group g of X[row][n]  (gradient of a per-group additive term)
// X point-blocked (ChainArgs::side_blocked), groups of whole 32-point blocks: lanes along the points of a block (128-byte lines),
// 8 features per pass, 5 DPP-free shuffle steps per feature
__global__ void __launch_bounds__(256) k_group_colsum_blocked(const float* __restrict__ X, int ldx, int N, int div, float* __restrict__ out) {
  const long long gidx = blockIdx.x;
  const int pp = threadIdx.x & 31, fq = threadIdx.x >> 5;
  const int nb = div >> 5;
  const long long b0 = gidx * nb;
  // 4 features per lane and pass (32 per workgroup): 4 x nb loads in flight before the first shuffle (one feature per pass left a
  // workgroup waiting for 4 loads at a time: 65 us per launch in the step, round 5)
  for (int n0 = 0; n0 < N; n0 += 32) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int bb = 0; bb < nb; bb += 4) {            // 16 independent loads per lane, then their sums
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + 8 * j + fq;
#pragma unroll
        for (int b = 0; b < 4; ++b) v[j][b] = (n < N && bb + b < nb) ? X[((b0 + bb + b) * ldx + n) * 32 + pp] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off);
      const int n = n0 + 8 * j + fq;
      if (pp == 0 && n < N) out[gidx * N + n] = acc[j];
    }
  }
}

__global__ void __launch_bounds__(256) k_group_colsum(const float* __restrict__ X, int ldx, int N, int div, float* __restrict__ out,
                                                      int blocked) {
  __shared__ float red[256];
  const long long gidx = blockIdx.x;
  const int TX = pow2_at_least(N), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const float* Xg = X + gidx * (long long)div * ldx;
  const long long r0 = gidx * (long long)div;
  for (int n0 = 0; n0 < N; n0 += TX) {
    const int n = n0 + tx;
    float acc = 0.f;
    if (n < N) {
      if (blocked) {       // (groups that are not whole blocks: correct, uncoalesced)
        for (int r = ty; r < div; r += TY) acc += X[(((r0 + r) >> 5) * ldx + n) * 32 + ((r0 + r) & 31)];
      } else {
#pragma unroll 8
        for (int r = ty; r < div; r += TY) acc += Xg[(long long)r * ldx + n];
      }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = TY / 2; s > 0; s >>= 1) {
      if (ty < s) red[threadIdx.x] += red[threadIdx.x + s * TX];
      __syncthreads();
    }
    if (ty == 0 && n < N) out[gidx * N + n] = red[tx];
    __syncthreads();
  }
}

int launch_group_colsum(const float* X, int ldx, int N, long long G, int div, float* out, int blocked, hipStream_t stream) {
  if (blocked && (div & 31) == 0) hipLaunchKernelGGL(k_group_colsum_blocked, dim3((unsigned)G), dim3(256), 0, stream, X, ldx, N, div, out);
  else hipLaunchKernelGGL(k_group_colsum, dim3((unsigned)G), dim3(256), 0, stream, X, ldx, N, div, out, blocked);
  return ndjir_check_launch();
}

long long wgrad_workspace(int K, int N, long long P) {
  long long splits = pick_splits(K, N, P, 1024);     // upper bound over both engines' plans (and NDJIR_WGRAD_BLOCKS)
  if (N <= SW_NMAX) {                                  // narrow path: one partial per SW_ROWS rows
    const long long nb = (P + SW_ROWS - 1) / SW_ROWS;
    if (nb > splits) splits = nb;
  }
  return splits * K * N + 4;        // + two maxima (k_absmax2) behind the partial sums
}

static int launch_split_reduce(const float* ws, float* out, long long KN, int S, int accum, hipStream_t stream) {
  if (S > 32) {
    hipLaunchKernelGGL(k_reduce_tall, dim3((unsigned)((KN + 31) / 32)), dim3(256), 0, stream, ws, out, KN, S, accum);
    return ndjir_check_launch();
  }
  int blocks = (int)(((KN + 3) / 4 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(256), 0, stream, ws, out, KN, S, accum);
  return ndjir_check_launch();
}

int launch_wgrad(const float* A, int lda, const float* B, int ldb, int K, int N, long long P, float* out, int accum,
                 float* workspace, int math, const unsigned* amax_a, const unsigned* amax_b, hipStream_t stream) {
  const int bf16x6 = (math == 1);
  const int f16x3 = (math == 2);
  if (K <= 0 || N <= 0) return NDJIR_OK;
  if (N <= SW_NMAX && (K & 3) == 0 && (lda & 3) == 0 && ((uintptr_t)A & 15) == 0) {
    if (P <= 0) {
      if (!accum && hipMemsetAsync(out, 0, (size_t)K * N * sizeof(float), stream) != hipSuccess) return NDJIR_ERR_LAUNCH;
      return NDJIR_OK;
    }
    const unsigned blocks = (unsigned)((P + SW_ROWS - 1) / SW_ROWS);
    hipLaunchKernelGGL(k_wgrad_narrow<SW_NMAX>, dim3(blocks), dim3(256), 0, stream, A, lda, B, ldb, K, N, P, workspace);
    return launch_split_reduce(workspace, out, (long long)K * N, (int)blocks, accum, stream);
  }
  const WgradPlan pl = wgrad_plan(K, N);
  static const int x6_blocks = [] { const char* e = getenv("NDJIR_WGRAD_BLOCKS"); int v = e ? atoi(e) : 512; return (v >= 64 && v <= 1024) ? v : 512; }();
  const int S = pick_splits(K, N, P, (bf16x6 || f16x3) ? x6_blocks : 256);
  long long rows = (P + S - 1) / S;
  rows = (rows + WG_C - 1) / WG_C * WG_C;
  if (f16x3) {
    static bool attr3 = false;
    if (!attr3) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3<2, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr3 = true;
    }
    if (!amax_a || !amax_b) {
      // a tensor whose producer did not record its maximum: one streaming pass over it (the slots live behind the
      // partial sums; pick_splits' upper bound sized the workspace)
      unsigned* slots = reinterpret_cast<unsigned*>(workspace + wgrad_workspace(K, N, P) - 4);
      zero_fill(reinterpret_cast<float*>(slots), 2, stream);      // (a kernel: memset nodes misbehaved inside captured graphs, see csrc/grid.hip)
      long long nb = (P * (K > N ? K : N) + 256 * 16 - 1) / (256 * 16);     // ~16 elements per thread
      if (nb > 2048) nb = 2048;
      if (nb < 1) nb = 1;
      hipLaunchKernelGGL(k_absmax2, dim3((unsigned)nb), dim3(256), 0, stream, A, lda, K, B, ldb, N, P, slots, amax_a ? 0 : 1,
                         amax_b ? 0 : 1);
      if (!amax_a) amax_a = slots;
      if (!amax_b) amax_b = slots + 1;
    }
    if (pl.tk > 0 && pl.tn > 0)
      wgrad3_go<2, 2, 2, 2>(S * pl.tk * pl.tn, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.tk, pl.tn, rows, 0, 0,
                            pl.Km < K ? pl.Km : K, pl.Nm < N ? pl.Nm : N, amax_a, amax_b);
    if (pl.sk > 0) {
      const int tn_all = (N + WG_T - 1) / WG_T;
      wgrad3_go<1, 4, 1, 1>(S * pl.sk * tn_all, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.sk, tn_all, rows, pl.Km, 0, K, N,
                            amax_a, amax_b);
    }
    if (pl.sn > 0 && pl.tk > 0)
      wgrad3_go<4, 1, 1, 1>(S * pl.tk * pl.sn, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.tk, pl.sn, rows, 0, pl.Nm, pl.Km, N,
                            amax_a, amax_b);
    return launch_split_reduce(workspace, out, (long long)K * N, S, accum, stream);
  }
  if (bf16x6) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad6<2, 2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr = true;
    }
    if (pl.tk > 0 && pl.tn > 0)
      wgrad6_go<2, 2, 2, 2>(S * pl.tk * pl.tn, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.tk, pl.tn, rows, 0, 0,
                            pl.Km < K ? pl.Km : K, pl.Nm < N ? pl.Nm : N);
    if (pl.sk > 0) {
      const int tn_all = (N + WG_T - 1) / WG_T;
      wgrad6_go<1, 4, 1, 1>(S * pl.sk * tn_all, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.sk, tn_all, rows, pl.Km, 0, K, N);
    }
    if (pl.sn > 0 && pl.tk > 0)
      wgrad6_go<4, 1, 1, 1>(S * pl.tk * pl.sn, stream, A, lda, B, ldb, K, N, P, workspace, S, pl.tk, pl.sn, rows, 0, pl.Nm, pl.Km, N);
    return launch_split_reduce(workspace, out, (long long)K * N, S, accum, stream);
  }
  if (pl.tk > 0 && pl.tn > 0)
    hipLaunchKernelGGL((k_wgrad<2, 2, 2, 2>), dim3(S * pl.tk * pl.tn), dim3(WG_THREADS), 0, stream, A, lda, B, ldb, K, N, P,
                       workspace, S, pl.tk, pl.tn, rows, 0, 0, pl.Km < K ? pl.Km : K, pl.Nm < N ? pl.Nm : N);
  if (pl.sk > 0) {       // rows [Km, K) of dW, all columns
    const int tn_all = (N + WG_T - 1) / WG_T;
    hipLaunchKernelGGL((k_wgrad<1, 4, 1, 1>), dim3(S * pl.sk * tn_all), dim3(WG_THREADS), 0, stream, A, lda, B, ldb, K, N, P,
                       workspace, S, pl.sk, tn_all, rows, pl.Km, 0, K, N);
  }
  if (pl.sn > 0 && pl.tk > 0) {   // columns [Nm, N), rows [0, Km)
    hipLaunchKernelGGL((k_wgrad<4, 1, 1, 1>), dim3(S * pl.tk * pl.sn), dim3(WG_THREADS), 0, stream, A, lda, B, ldb, K, N, P,
                       workspace, S, pl.tk, pl.sn, rows, 0, pl.Nm, pl.Km, N);
  }
  return launch_split_reduce(workspace, out, (long long)K * N, S, accum, stream);
}


// ---- grouped weight gradients: MANY dW_o = sum_src A^T B in ONE launch sequence + ONE split-reduction launch (round 4) ----
// A training step issues ~47 weight-gradient GEMMs of one net layer each.  Launched one by one, each has to fill 256 CUs on
// its own: 128 point-axis splits per 256 x 256 gradient, i.e. 32 MB of partial slabs written and re-read per layer (1.5 GB
// per step), a prologue / epilogue per 16 chunks of work, and a reduction launch (k_reduce_tall, 61 per step).  Grouped, the
// work items of all layers share the machine: a split is thousands of points long, the slabs shrink by the number of layers
// in the group, and one reduction launch serves them all.  An output may have several operand pairs ("sources": the
// geometric net's dW_j = A_j^T delta_j + gbar_j^T s_j), each with its own operand scales: every work item writes a
// de-scaled partial slab, the reduction sums the slabs of all sources of an output.
// A source's K x N output is cut into regions, each tiled by one kind of work item (k_wgrad_group, two workgroups per CU):
//   kind 0      128 x 128 tile (ragged edges masked), kind 1 = 32 x 128 strip (ragged K <= 32 rows), kind 2 = 128 x 32 strip
//   (ragged N), kind 4 = 64 x 128 item (ragged K of 33 .. 64 rows: the 39- / 43- / 52-wide first layers; -0.08 ms per step
//   against two rows of strips -- while 32 x 256 / 256 x 32 strips, half as many items for the same bytes, measured +0.06 ms:
//   their planes and registers are paid by every item of the launch), kind 3 = streaming reduction for outputs <= 8 wide.
// Three other tile designs were written, tested bit-for-bit against these and measured in round 4 on 8 x (256 x 256 x 65 536)
// (tools/wgrad_group_time.py; this kernel: 50.8 us per layer) -- none is in the tree:
//   * one 4-wave workgroup per CU with 512 registers per lane, 128 x 256 tile, 16-byte loads, two chunks of operands in
//     flight, double-buffered planes, split interleaved with the MFMAs by the compiler: 55.1 us;
//   * 8-wave workgroup with specialised waves (4 multiply, 4 load two chunks ahead and split), one barrier per chunk: 73.8 us;
//   * this tile template at 256 x 128 with eight waves, one workgroup per CU: 48.6 us alone, but +0.5 ms per step (its ragged
//     leftovers need a second launch of this kernel);
//   * the two tiles that share an A operand as the two halves of ONE 512-thread workgroup (same code, barriers shared, so the
//     second read of A hits L1): 101 us on a box where this kernel takes 60 -- the two independent workgroups of a CU hide
//     each other's load and split phases, two halves in lockstep do not.
// Also measured: segments laid out source by source (a source's strips next to its tiles, same XCD, so that their second read of
// the other operand finds it in L2) instead of kind by kind: 8.25 vs 8.26 ms per step -- no difference.
// PMC on the grouped kernel (profiles/r04_pmc_wgrad_group.txt): FETCH_SIZE = the algorithmic operand bytes, half of the L2
// requests hit (the tile pair of a split), the matrix pipe busy 21 %, 63 % of the wave cycles waiting: ~10 bytes / clk / CU
// of loads whatever the structure -- the rate MI355X_MICROARCH.md gives for HBM-bound global_load_dwordx4.
constexpr int WGG_MAX_SRC = 20;      // per writer launch
constexpr int WGG_MAX_SEG = 48;
constexpr int WGG_MAX_OUT = 20;
constexpr int WGG_NARROW_ROWS = 128;      // points per narrow work item: its row loop is a latency chain (32 steps of 4 rows), so
                                          // items are kept short (a 1024-row item ran ~100 us and was the tail of every launch) and, since round 6, issued LAST: they fill the tail

struct WggSrc {
  const float* A;
  const float* B;
  const unsigned* amax_a;
  const unsigned* amax_b;
  float* partial;            // this source's slabs: partial[split][K][N]
  long long P;
  long long rows;            // points per split (multiple of WG_C)
  int lda, ldb, K, N;
  int S, layout;             // layout: bit 0 = A point-blocked, bit 1 = B point-blocked (ChainArgs::side_blocked)
};
struct WggSeg {              // a run of consecutive workgroups tiling one region of one source's output with one kind of item
  int first, count;          // first workgroup (multiple of 8), workgroups (multiple of 8; the ones past S * tiles idle)
  short src, kind;
  short tiles, tiles_n;      // tiles of the region, tiles per tile row
  short k_off, n_off, k_end, n_end;   // the region: rows [k_off, k_end), columns [n_off, n_end) of dW
};
struct WggOut {
  float* out;                // (K, N), row stride ldo
  const float* partial;      // S slabs of KN floats, `pstride` floats apart
  int KN, N, ldo, S, accum;
  int first;                 // first workgroup of this output in the reduction launch
  int pstride, pad;          // (a weight gradient: KN; deferred bias gradients of a chain launch: the launch's partial-row length)
};
// 16-byte accesses in the reduction: contiguous destination, every slab 16-byte aligned.  The DESTINATION only needs its 4 bytes:
// gradients live at arbitrary float offsets of the step's flat bucket (round 4: 37 147 of a step's 39 835 reduction workgroups
// took the 4-byte path for that reason alone) and gfx950 moves a dwordx4 at any 4-byte alignment (tools/ubench/unaligned.hip).
__host__ __device__ static inline bool wgg_out_vec(const WggOut& o) {
  return (o.KN & 3) == 0 && o.ldo == o.N && (o.pstride & 3) == 0 && (reinterpret_cast<uintptr_t>(o.partial) & 15) == 0;
}
typedef float wgg_f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// The whole group's work description lives in DEVICE memory (the head of the caller's workspace): a training step's ~57
// operand pairs / ~150 segments do not fit the 4 KB of kernel arguments, and cutting the group into several launches costs
// a drain of the machine per cut.  The table is written by tiny launches that carry pieces of it in THEIR arguments
// (k_wgg_write: capturable into a HIP graph, no host memory for a replay to re-read), then ONE k_wgrad_group launch and ONE
// k_wgrad_group_reduce launch read it.
constexpr int WGT_MAX_SRC = 256;
constexpr int WGT_MAX_SEG = 768;
constexpr int WGT_MAX_OUT = 256;
struct WggTable {
  int n_seg, n_out, pad0, pad1;
  unsigned amax_fill[WGT_MAX_SRC][2];      // maxima of the operands that came without a recorded one (k_wgg_absmax), as bit patterns
  WggSrc src[WGT_MAX_SRC];
  WggSeg seg[WGT_MAX_SEG];
  WggOut out[WGT_MAX_OUT];
};
struct WggPiece;
struct WggPiece {            // what one writer launch carries (<= 4 KB of kernel arguments)
  int n_src, n_seg, n_out, src0, seg0, out0, tot_seg, tot_out;
  WggSrc src[WGG_MAX_SRC];
  WggSeg seg[WGG_MAX_SEG];
  WggOut out[WGG_MAX_OUT];
};

// streaming item for N <= 8 (K % 4 == 0, lda % 4 == 0, A 16-byte aligned): the body of k_wgrad_narrow over [p0, p1)
template <int NMAX>
__device__ __forceinline__ void wgrad_narrow_rows(const float* __restrict__ A_, int lda, const float* __restrict__ B_, int ldb,
                                                  int K, int N, long long p0, long long p1, float* __restrict__ part_,
                                                  float4* red) {
  const WG_G float* A = wg_global(A_);
  const WG_G float* B = wg_global(B_);
  WG_G float* part = wg_global(part_);
  const int KQ = K >> 2;
  const int TX = pow2_at_least(KQ), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  for (int q0 = 0; q0 < KQ; q0 += TX) {
    const int kq = q0 + tx;
    float4 acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kq < KQ) {
#pragma unroll 4
      for (long long p = p0 + ty; p < p1; p += TY) {
        const wg_f32x4 av = *reinterpret_cast<const WG_G wg_f32x4*>(A + p * lda + 4 * kq);
        const float4 a = make_float4(av[0], av[1], av[2], av[3]);
        const WG_G float* b = B + p * ldb;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) if (n < N) {
          const float bn = b[n];
          acc[n].x = fmaf(a.x, bn, acc[n].x); acc[n].y = fmaf(a.y, bn, acc[n].y);
          acc[n].z = fmaf(a.z, bn, acc[n].z); acc[n].w = fmaf(a.w, bn, acc[n].w);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {                                   // uniform
        red[threadIdx.x] = acc[n];
        __syncthreads();
        for (int s = TY / 2; s > 0; s >>= 1) {
          if (ty < s) {
            float4 o = red[threadIdx.x + s * TX], m = red[threadIdx.x];
            red[threadIdx.x] = make_float4(m.x + o.x, m.y + o.y, m.z + o.z, m.w + o.w);
          }
          __syncthreads();
        }
        if (ty == 0 && kq < KQ) {
          const float4 t = red[tx];
          WG_G float* o = part + (long long)(4 * kq) * N + n;
          o[0] = t.x; o[N] = t.y; o[2 * N] = t.z; o[3 * N] = t.w;
        }
        __syncthreads();
      }
    }
  }
}

// the same item with A point-blocked (the last hidden activation of a net whose output layer is <= 8 wide): thread = (feature,
// block phase).  A feature's 32 points of a block are 128 contiguous bytes (8 x 16-byte loads in flight per thread); B's rows of
// the item (<= WGG_NARROW_ROWS x 8 floats) are staged in LDS once, zero-padded to 8 columns, and read back as broadcasts.
template <int NMAX>
__device__ __forceinline__ void wgrad_narrow_rows_blocked(const float* __restrict__ A_, int lda, const float* __restrict__ B_, int ldb,
                                                          int K, int N, long long p0, long long p1, float* __restrict__ part_,
                                                          float* lds) {
  static_assert(NMAX == 8, "two 16-byte broadcasts per point");
  const WG_G float* A = wg_global(A_);
  const WG_G float* B = wg_global(B_);
  WG_G float* part = wg_global(part_);
  float* Bs = lds;                                   // [rows][8]
  float* red = lds + WGG_NARROW_ROWS * 8;            // [256]
  const int rows = (int)(p1 - p0);                   // (multiple of 32)
  for (int t = threadIdx.x; t < rows * 8; t += 256) {
    const int r = t >> 3, n = t & 7;
    Bs[t] = n < N ? B[(p0 + r) * ldb + n] : 0.f;
  }
  __syncthreads();
  const int TX = pow2_at_least(K), TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int nblk = rows >> 5;
  for (int k0 = 0; k0 < K; k0 += TX) {
    const int k = k0 + tx;
    float acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
    if (k < K) {
      for (int b = ty; b < nblk; b += TY) {
        const WG_G wg_f32x4* ap = reinterpret_cast<const WG_G wg_f32x4*>(A + (((p0 >> 5) + b) * lda + k) * 32);
        wg_f32x4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = ap[i];
        const wg_f32x4* bs = reinterpret_cast<const wg_f32x4*>(Bs + b * 32 * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const wg_f32x4 b0 = bs[(4 * i + q) * 2], b1 = bs[(4 * i + q) * 2 + 1];
            const float av = a[i][q];
            acc[0] = fmaf(av, b0[0], acc[0]); acc[1] = fmaf(av, b0[1], acc[1]); acc[2] = fmaf(av, b0[2], acc[2]); acc[3] = fmaf(av, b0[3], acc[3]);
            acc[4] = fmaf(av, b1[0], acc[4]); acc[5] = fmaf(av, b1[1], acc[5]); acc[6] = fmaf(av, b1[2], acc[6]); acc[7] = fmaf(av, b1[3], acc[7]);
          }
      }
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {                                   // uniform
        red[threadIdx.x] = acc[n];
        __syncthreads();
        for (int s = TY / 2; s > 0; s >>= 1) {
          if (ty < s) red[threadIdx.x] += red[threadIdx.x + s * TX];
          __syncthreads();
        }
        if (ty == 0 && k < K) part[(long long)k * N + n] = red[tx];
        __syncthreads();
      }
    }
  }
}

// rows x columns of dW one item of a kind covers
__host__ __device__ static inline int wgg_tile_k(int kind) { return kind == 1 ? WG_STRIP : kind == 4 ? 2 * WG_STRIP : WG_T; }
__host__ __device__ static inline int wgg_tile_n(int kind) { return kind == 2 ? WG_STRIP : kind == 5 ? 2 * WG_T : WG_T; }

// workgroup -> (segment, split, tile): whole splits per XCD (workgroups are dealt round-robin to the 8 XCDs; first % 8 == 0,
// count % 8 == 0), so that the tiles that share a range of points share an L2.  Returns false for an idle workgroup.
__device__ __forceinline__ bool wgg_locate(const WggTable& a, int& si, int& split, int& tile, int block_base = 0) {
  const int b = blockIdx.x + block_base;
  int lo = 0, hi = a.n_seg - 1;          // the last segment whose first workgroup is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.seg[mid].first <= b) lo = mid; else hi = mid - 1;
  }
  si = lo;
  const int local = b - a.seg[si].first, count = a.seg[si].count, T = a.seg[si].tiles;
  const int vid = (local & 7) * (count >> 3) + (local >> 3);
  split = vid / T;
  tile = vid - split * T;
  return split < a.src[a.seg[si].src].S;
}

// Operands without a recorded maximum (single-layer operators on per-ray rows, the nets of the 16 384-point passes): one pre-pass
// over each such operand, many workgroups per operand, before the item launches.  (Rounds 4 - 5: every work item scanned its own
// rows and columns first, one load per lane and step at memory latency -- 57 us for 512 rows, ~300 us for a split of 2 752; the
// items of these sources were the long pole of both item launches, tools/wgrad_items.py.)  Largest finite |x| as a bit pattern.
constexpr int WGG_AMAX_CHUNKS = 32;        // workgroups per operand
struct WggAmaxJobs { short job[2 * WGT_MAX_SRC]; };      // 2 x source + operand, in the kernel arguments (1 KB)
__global__ void __launch_bounds__(256) k_wgg_absmax(WggTable* __restrict__ tab, const WggAmaxJobs jobs) {
  __shared__ unsigned red[256];
  const int i = (int)jobs.job[blockIdx.y] >> 1, op = (int)jobs.job[blockIdx.y] & 1;
  const WggSrc& s = tab->src[i];
  const WG_G float* X = wg_global(op ? s.B : s.A);
  const int ld = op ? s.ldb : s.lda, W = op ? s.N : s.K;
  const bool blocked = (s.layout >> op) & 1;
  long long per = (s.P + WGG_AMAX_CHUNKS - 1) / WGG_AMAX_CHUNKS;
  per = (per + 31) / 32 * 32;
  const long long p0 = (long long)blockIdx.x * per;
  long long p1 = p0 + per;
  if (p1 > s.P) p1 = s.P;
  unsigned m = 0;
  if (p0 < p1) {
    const long long n = (p1 - p0) * W;
    constexpr int U = 8;
    for (long long t0 = threadIdx.x; t0 < n; t0 += 256 * U) {
      unsigned b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long t = t0 + 256 * u;
        b[u] = 0u;
        if (t < n) {
          long long idx;
          if (blocked) { const long long q = t >> 5; idx = (((p0 >> 5) + q / W) * ld + q % W) * 32 + (t & 31); }     // (32 points of a feature contiguous)
          else idx = (p0 + t / W) * ld + t % W;
          b[u] = __float_as_uint(X[idx]) & 0x7fffffffu;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) if (b[u] < 0x7f800000u && b[u] > m) m = b[u];
    }
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st && red[threadIdx.x + st] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] > 0u) atomicMax(&tab->amax_fill[i][op], red[0]);
}

__global__ void __launch_bounds__(256) k_wgg_write(const WggPiece p, WggTable* __restrict__ t) {
  const int tid = threadIdx.x;
  if (tid == 0) { t->n_seg = p.tot_seg; t->n_out = p.tot_out; }
  for (int i = tid; i < p.n_src * 2; i += 256) t->amax_fill[p.src0 + (i >> 1)][i & 1] = 0u;
  // (plain word copies: the structs are PODs of 4-byte-aligned members)
  const int ws = sizeof(WggSrc) / 4, wg = sizeof(WggSeg) / 4, wo = sizeof(WggOut) / 4;
  const unsigned* a = reinterpret_cast<const unsigned*>(p.src);
  unsigned* d = reinterpret_cast<unsigned*>(t->src + p.src0);
  for (int i = tid; i < p.n_src * ws; i += 256) d[i] = a[i];
  a = reinterpret_cast<const unsigned*>(p.seg);
  d = reinterpret_cast<unsigned*>(t->seg + p.seg0);
  for (int i = tid; i < p.n_seg * wg; i += 256) d[i] = a[i];
  a = reinterpret_cast<const unsigned*>(p.out);
  d = reinterpret_cast<unsigned*>(t->out + p.out0);
  for (int i = tid; i < p.n_out * wo; i += 256) d[i] = a[i];
}

#ifdef WGG_ITEMLOG      // tools/wgrad_items.py: (kind, first and last 100 MHz tick) of every workgroup of the last k_wgrad_group launch
__device__ long long wgg_item_log[16384][3];
struct WggItemStamp {
  int b;
  __device__ WggItemStamp(int kind, int base = 0) : b((int)blockIdx.x + base) {
    if (threadIdx.x == 0 && b < 16384) { wgg_item_log[b][0] = kind; wgg_item_log[b][1] = (long long)__builtin_amdgcn_s_memrealtime(); }
  }
  __device__ ~WggItemStamp() {
    __syncthreads();
    if (threadIdx.x == 0 && b < 16384) wgg_item_log[b][2] = (long long)__builtin_amdgcn_s_memrealtime();
  }
};
#endif
// The 128 x 256 items (kind 5) of a group: one workgroup per CU, 512 registers per lane, 96 KB of LDS -- see wgrad3_pipe.  Their
// segments come first in the table: workgroups [0, wide blocks) are this launch's, the rest k_wgrad_group's.
__global__ void __launch_bounds__(WG_THREADS, 1) k_wgrad_group_wide(const WggTable* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
  const WggTable& a = *tab;
  int si, split, tile;
  if (!wgg_locate(a, si, split, tile)) return;
  const WggSeg& g = a.seg[si];
#ifdef WGG_ITEMLOG
  WggItemStamp stamp_(g.kind + 16 * (a.src[g.src].layout & 3) + 64 * (a.src[g.src].P > 100000), 8192);
#endif
  const WggSrc& s = a.src[g.src];
  const int K = s.K, N = s.N;
  const long long p_begin = (long long)split * s.rows;
  long long p_end = p_begin + s.rows;
  if (p_end > s.P) p_end = s.P;
  float* slab = s.partial + (long long)split * K * N;
  const int ti = tile / g.tiles_n, tj = tile - ti * g.tiles_n;
  const int k0 = g.k_off + ti * WG_T, n0 = g.n_off + tj * 2 * WG_T;
  unsigned ma, mb;
  if (s.amax_a) ma = *wg_global(s.amax_a);
  else ma = a.amax_fill[g.src][0];
  if (s.amax_b) mb = *wg_global(s.amax_b);
  else mb = a.amax_fill[g.src][1];
  float sa, ia, sb, ib;
  wg_scale_from_max(ma, sa, ia);
  wg_scale_from_max(mb, sb, ib);
  const int lay = s.layout & 3;
  const bool full = g.k_end - k0 >= WG_T && g.n_end - n0 >= 2 * WG_T;
#define WGG_PIPE(LAYV)                                                                                                          \
    if (full) wgrad3_pipe<LAYV, true, 4>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds); \
    else wgrad3_pipe<LAYV, false, 4>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds)
  if (lay == 3) { WGG_PIPE(3); }
  else if (lay == 1) { WGG_PIPE(1); }
  else if (lay == 2) { WGG_PIPE(2); }
  else { WGG_PIPE(0); }
#undef WGG_PIPE
}

__global__ void __launch_bounds__(WG_THREADS, 2) k_wgrad_group(const WggTable* __restrict__ tab, int block_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned short wg_lds[];
  const WggTable& a = *tab;
  int si, split, tile;
  if (!wgg_locate(a, si, split, tile, block_base)) return;
  const WggSeg& g = a.seg[si];
#ifdef WGG_ITEMLOG
  WggItemStamp stamp_(g.kind + 16 * (a.src[g.src].layout & 3) + 64 * (a.src[g.src].P > 100000));
#endif
  const WggSrc& s = a.src[g.src];
  const int K = s.K, N = s.N, kind = g.kind;
  const long long p_begin = (long long)split * s.rows;
  long long p_end = p_begin + s.rows;
  if (p_end > s.P) p_end = s.P;
  float* slab = s.partial + (long long)split * K * N;
  if (kind == 3) {
    if (s.layout & 1) wgrad_narrow_rows_blocked<SW_NMAX>(s.A, s.lda, s.B, s.ldb, K, N, p_begin, p_end, slab, reinterpret_cast<float*>(wg_lds));
    else wgrad_narrow_rows<SW_NMAX>(s.A, s.lda, s.B, s.ldb, K, N, p_begin, p_end, slab, reinterpret_cast<float4*>(wg_lds));
    return;
  }
  const int ti = tile / g.tiles_n, tj = tile - ti * g.tiles_n;
  const int k0 = g.k_off + ti * wgg_tile_k(kind), n0 = g.n_off + tj * wgg_tile_n(kind);
  unsigned ma, mb;
  if (s.amax_a) ma = *wg_global(s.amax_a);
  else ma = a.amax_fill[g.src][0];
  if (s.amax_b) mb = *wg_global(s.amax_b);
  else mb = a.amax_fill[g.src][1];
  float sa, ia, sb, ib;
  wg_scale_from_max(ma, sa, ia);
  wg_scale_from_max(mb, sb, ib);
#ifdef WGG_ONLY
  wgrad3_pipe<WGG_ONLY, true, 2>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds);
  return;
#endif
#ifndef NDJIR_WGRAD_NO_PIPE
  if (kind == 0) {
    const int lay = s.layout & 3;
    const bool full = g.k_end - k0 >= WG_T && g.n_end - n0 >= WG_T;
#define WGG_PIPE(LAYV)                                                                                                          \
    if (full) wgrad3_pipe<LAYV, true, 2>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds); \
    else wgrad3_pipe<LAYV, false, 2>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds)
    if (lay == 3) { WGG_PIPE(3); }
    else if (lay == 1) { WGG_PIPE(1); }
    else if (lay == 2) { WGG_PIPE(2); }
    else { WGG_PIPE(0); }
#undef WGG_PIPE
    return;
  }
#endif
  if (kind == 0) {
    const int lay = s.layout & 3;
    if (lay == 3) wgrad3_tile<2, 2, 2, 2, 3>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds);
    else if (lay == 1) wgrad3_tile<2, 2, 2, 2, 1>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds);
    else if (lay == 2) wgrad3_tile<2, 2, 2, 2, 2>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds);
    else wgrad3_tile<2, 2, 2, 2, 0>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds);
  } else if (kind == 1)
    wgrad3_tile<1, 4, 1, 1>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds, s.layout & 3);
  else if (kind == 4)
    wgrad3_tile<2, 2, 1, 2>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds, s.layout & 3);
  else
    wgrad3_tile<4, 1, 1, 1>(s.A, s.lda, s.B, s.ldb, N, p_begin, p_end, slab, k0, n0, g.k_end, g.n_end, sa, ia, sb, ib, wg_lds, s.layout & 3);
}

// out (+)= sum over the S slabs: a workgroup owns 32 vectors (VEC floats each) of one output, 8 slab phases
template <int VEC>
__device__ __forceinline__ void wgg_reduce_block(const WggOut& o, int blk, float* red) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long long i = ((long long)blk * 32 + tx) * VEC;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (i < o.KN) {
    const WG_G float* p = wg_global(o.partial) + i;
#pragma unroll 4
    for (int sidx = ty; sidx < o.S; sidx += 8) {
      if (VEC == 4) {
        const wg_f32x4 t = *reinterpret_cast<const WG_G wg_f32x4*>(p + (long long)sidx * o.pstride);
        acc[0] += t[0]; acc[1 % VEC] += t[1]; acc[2 % VEC] += t[2]; acc[3 % VEC] += t[3];
      } else {
        acc[0] += p[(long long)sidx * o.pstride];
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[v * 256 + threadIdx.x] = acc[v];
  __syncthreads();
  if (ty == 0 && i < o.KN) {
    float t[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      t[v] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t[v] += red[v * 256 + q * 32 + tx];
    }
    if (VEC == 4) {            // contiguous output, 4-byte aligned
      WG_G wgg_f32x4u* dst = reinterpret_cast<WG_G wgg_f32x4u*>(wg_global(o.out) + i);
      wgg_f32x4u r = {t[0], t[1 % VEC], t[2 % VEC], t[3 % VEC]};
      if (o.accum) { const wgg_f32x4u c = *dst; r += c; }
      *dst = r;
    } else {
      const long long row = i / o.N;
      WG_G float* dst = wg_global(o.out) + row * o.ldo + (i - row * o.N);
      *dst = o.accum ? *dst + t[0] : t[0];
    }
  }
}

__global__ void __launch_bounds__(256) k_wgrad_group_reduce(const WggTable* __restrict__ tab, int block_base) {
  __shared__ float red[4 * 256];
  const WggTable& a = *tab;
  const int b = blockIdx.x + block_base;
  int lo = 0, hi = a.n_out - 1;          // the last output whose first workgroup is <= b
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.out[mid].first <= b) lo = mid; else hi = mid - 1;
  }
  const WggOut& o = a.out[lo];
  const bool vec = wgg_out_vec(o);
  const int blk = b - o.first;
  if (vec) wgg_reduce_block<4>(o, blk, red);
  else wgg_reduce_block<1>(o, blk, red);
}

static inline bool wgg_narrow(const float* A, int lda, int K, int N, int layout = 0) {
  if ((layout & 2) != 0) return false;              // (B of a narrow output is the chain's input gradient: never blocked)
  if ((layout & 1) != 0) return N <= SW_NMAX;       // blocked A: any K (wgrad_narrow_rows_blocked)
  return N <= SW_NMAX && (K & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

constexpr int WGG_DEFAULT_ITEMS = 3072;     // 128 x 128-tile equivalents a grouped call aims for (12 per CU).  Round 5, with the one-workgroup-per-CU wide launch, whose tail counts: step 7.97 / 7.96 / 7.78 / 7.72 / 7.82 ms at 1024 / 1536 / 2048 / 3072 / 4096 (two sweeps, one box)

// The regions of a K x N output and the kind of item that tiles each (at most 3), as the per-layer launcher cuts it: 128 x 128
// tiles over [0, Km) x [0, Nm), remainders of at most 64 as 32-wide strips (K strip spans all of N, N strip spans [0, Km)),
// larger remainders as a further (ragged) tile.
struct WggRegion { int kind, k_off, n_off, k_end, n_end, tiles_k, tiles_n; };
// The 128 x 256 items: on by default (NDJIR_WGRAD_NO_WIDE=1 switches them off).  On 8 x (256 x 256 x 65536) alone they measure
// what the 128 x 128 tile measures (43.7 against 42.9 us per layer, same box: halving the LDS reads per MFMA buys nothing there --
// the loop's cost is the SUM of its matrix, vector, LDS-write and load-issue time, tools/ubench/shadow.hip, and one wave per SIMD
// has no second wave to fill its barrier and LDS-latency gaps); in the STEP, whose 66 operand pairs include ragged and
// mixed-layout shapes and whose tiles then read A half as often, they are worth 2.5 %: 7.93 / 7.92 / 7.91 ms against
// 8.13 / 8.14 / 8.11 in three interleaved pairs on one box (profiles/r05_wgrad_ab.txt).
static bool wgg_wide_enabled() {
  static const bool on = getenv("NDJIR_WGRAD_NO_WIDE") == nullptr;
  return on;
}
static int wgg_regions(int K, int N, WggRegion* rg) {
  int n = 0;
  const int rk = K % WG_T, rn = N % WG_T;
  const int Km = (rk != 0 && rk <= 2 * WG_STRIP) ? K - rk : K;
  const int Nm = (rn != 0 && rn <= 2 * WG_STRIP) ? N - rn : N;
  // main region: 128 x 256 items (kind 5) over the columns they cover at least half of, 128 x 128 tiles over the rest
  const int tw = wgg_wide_enabled() ? (Nm + WG_T - 1) / (2 * WG_T) : 0;
  const int Nw = tw * 2 * WG_T < Nm ? tw * 2 * WG_T : (tw > 0 ? Nm : 0);
  if (Km > 0 && Nw > 0) rg[n++] = {5, 0, 0, Km, Nw, (Km + WG_T - 1) / WG_T, tw};
  if (Km > 0 && Nm > Nw) rg[n++] = {0, 0, Nw, Km, Nm, (Km + WG_T - 1) / WG_T, (Nm - Nw + WG_T - 1) / WG_T};
  if (Km > 0 && N > Nm) rg[n++] = {2, 0, Nm, Km, N, (Km + WG_T - 1) / WG_T, (N - Nm + WG_STRIP - 1) / WG_STRIP};
  if (K > Km) {
    // rows [Km, K): up to 32 of them as one row of 32 x 128 strips, 33..64 as one row of 64 x 128 items (kind 4) -- two rows of
    // strips would read B twice, and a strip's chunk loop costs what a tile's does (round 4: all strips of a step 0.25 ms)
    if (K - Km > WG_STRIP) rg[n++] = {4, Km, 0, K, N, 1, (N + WG_T - 1) / WG_T};
    else rg[n++] = {1, Km, 0, K, N, 1, (N + WG_T - 1) / WG_T};
  }
  return n;
}
static double wgg_units(int K, int N) {        // work of one split in 128 x 128-tile equivalents
  WggRegion rg[5];
  const int n = wgg_regions(K, N, rg);
  double u = 0.0;
  for (int i = 0; i < n; ++i)
    u += ((rg[i].kind == 1 || rg[i].kind == 2) ? 0.25 : rg[i].kind == 4 ? 0.5 : rg[i].kind == 5 ? 2.0 : 1.0) * rg[i].tiles_k * rg[i].tiles_n;
  return u;
}

static_assert(sizeof(WggPiece) <= 4096, "kernel arguments of k_wgg_write");
constexpr long long WGT_FLOATS = (sizeof(WggTable) + 15) / 16 * 4;      // the table at the head of the workspace, in floats

// outputs [o0, o1) whose sources (those with points) and segments fit one table
static int wgg_chunk_end(int o0, int n_src, const long long* P, const int* out_id, int n_out) {
  int ns = 0, o1 = o0;
  while (o1 < n_out && o1 - o0 < WGT_MAX_OUT) {
    int cnt = 0;
    for (int i = 0; i < n_src; ++i) cnt += (out_id[i] == o1 && P[i] > 0);
    if (ns + cnt > WGT_MAX_SRC || 4 * (ns + cnt) > WGT_MAX_SEG) break;      // (<= 4 regions per source)
    ns += cnt;
    ++o1;
  }
  return o1;
}

// splits / rows per split of the sources of outputs [o0, o1) -- one launch (same rule for the workspace size and the launch)
static void wgg_split_plan(int o0, int o1, int n_src, const float* const* A, const int* lda, const long long* P, const int* out_id,
                           const int* K, const int* N, int target_items, int* S, long long* rows, const int* layout) {
  const bool defaulted = target_items <= 0;
  if (defaulted) target_items = WGG_DEFAULT_ITEMS;
  double units = 0.0;               // main-tile equivalents x points
  for (int i = 0; i < n_src; ++i) {
    const int o = out_id[i];
    if (o < o0 || o >= o1 || P[i] <= 0 || wgg_narrow(A ? A[i] : nullptr, lda[i], K[o], N[o], layout ? layout[i] : 0)) continue;
    units += wgg_units(K[o], N[o]) * (double)P[i];
  }
  long long target = (long long)(units / target_items);
  target = (target + WG_C - 1) / WG_C * WG_C;
  if (target < 16 * WG_C) target = 16 * WG_C;
  if (defaulted && wgg_wide_enabled()) {
    // The wide items run one per CU and all take the same time: their launch costs ceil(items / 256) rounds, and a last round
    // that is a third full is a tenth of the launch idle (the item-count sweep of round 5: 2560 items -> 4.2 rounds, 7.85 ms;
    // 3072 -> 4.95 rounds, 7.73 ms).  Among the split lengths within 25 % of the nominal one take the one whose wide items
    // fill their last round best (ties: the nearest).
    auto wide_items = [&](long long t) {
      long long w = 0;
      for (int i = 0; i < n_src; ++i) {
        const int o = out_id[i];
        if (o < o0 || o >= o1 || P[i] <= 0 || wgg_narrow(A ? A[i] : nullptr, lda[i], K[o], N[o], layout ? layout[i] : 0)) continue;
        WggRegion rg[5];
        const int n = wgg_regions(K[o], N[o], rg);
        long long s = (P[i] + t - 1) / t, r = ((P[i] + s - 1) / s + WG_C - 1) / WG_C * WG_C;
        s = (P[i] + r - 1) / r;
        for (int q = 0; q < n; ++q)
          if (rg[q].kind == 5) w += s * rg[q].tiles_k * rg[q].tiles_n;
      }
      return w;
    };
    constexpr long long CUS = 256;
    long long best = target;
    double best_waste = 2.0;
    for (long long t = target * 3 / 4 / WG_C * WG_C; t <= target * 5 / 4; t += WG_C) {
      if (t < 16 * WG_C) continue;
      const long long w = wide_items(t);
      if (w <= 0) { best = target; break; }
      const long long rounds = (w + CUS - 1) / CUS;
      const double waste = (double)(rounds * CUS - w) / (double)(rounds * CUS);
      const long long d = t > target ? t - target : target - t, db = best > target ? best - target : target - best;
      if (waste < best_waste - 1e-9 || (waste < best_waste + 1e-9 && d < db)) { best_waste = waste; best = t; }
    }
    target = best;
  }
  for (int i = 0; i < n_src; ++i) {
    const int o = out_id[i];
    if (o < o0 || o >= o1) continue;
    if (P[i] <= 0) { S[i] = 0; rows[i] = 0; continue; }
    if (wgg_narrow(A ? A[i] : nullptr, lda[i], K[o], N[o], layout ? layout[i] : 0)) {
      rows[i] = WGG_NARROW_ROWS;
      S[i] = (int)((P[i] + WGG_NARROW_ROWS - 1) / WGG_NARROW_ROWS);
      continue;
    }
    long long s = (P[i] + target - 1) / target;
    // A segment's workgroups are dealt to the 8 XCDs in eighths (wgg_locate) and padded to a multiple of 8: 21 splits x 2 tiles
    // = 42 items in 48 workgroups leave the eighth XCD none -- and did so for every 256-wide layer: one XCD sat out most of both
    // item launches (round 6, tools/wgrad_items.py: 909 us for 735 us of packed work).  Splits are rounded up so that
    // splits x tiles of every region of the source is a multiple of 8.
    static const bool no_round = getenv("NDJIR_WGG_NO_SPLIT_ROUNDING") != nullptr;      // A/B switch
    if (!no_round) {
      WggRegion rg[5];
      const int nr = wgg_regions(K[o], N[o], rg);
      int g = 8;
      for (int q = 0; q < nr; ++q) {
        int t = rg[q].tiles_k * rg[q].tiles_n, a = 8;
        while (t % a) a >>= 1;                  // gcd(8, tiles)
        if (a < g) g = a;
      }
      const long long m = 8 / g;
      const long long s2 = (s + m - 1) / m * m;
      if ((P[i] + s2 - 1) / s2 >= 8 * WG_C) s = s2;        // (keep splits of at least 8 chunks)
    }
    long long r = (P[i] + s - 1) / s;
    r = (r + WG_C - 1) / WG_C * WG_C;
    S[i] = (int)((P[i] + r - 1) / r);
    rows[i] = r;
  }
}

// launches of k_wgrad_group (= launches of k_wgrad_group_reduce) a grouped call issues: one per table
int wgrad_group_launches(int n_src, const long long* P, const int* out_id, int n_out) {
  int n = 0;
  for (int o0 = 0; o0 < n_out;) {
    const int o1 = wgg_chunk_end(o0, n_src, P, out_id, n_out);
    if (o1 == o0) return -1;
    ++n;
    o0 = o1;
  }
  return n;
}

long long wgrad_group_workspace(int n_src, const float* const* A, const int* lda, const long long* P, const int* out_id, int n_out,
                                const int* K, const int* N, int target_items, const int* layout) {
  if (n_src <= 0) return 2 * WGT_FLOATS + 4;          // (reduce-only calls: room for their tables)
  if (n_src > 65536) return 0;
  int* S = (int*)alloca(sizeof(int) * n_src);
  long long* rows = (long long*)alloca(sizeof(long long) * n_src);
  long long total = 0;
  for (int o0 = 0; o0 < n_out;) {
    const int o1 = wgg_chunk_end(o0, n_src, P, out_id, n_out);
    if (o1 == o0) return 0;
    wgg_split_plan(o0, o1, n_src, A, lda, P, out_id, K, N, target_items, S, rows, layout);
    total += WGT_FLOATS;
    for (int o = o0; o < o1; ++o) {
      long long st = 0;
      for (int i = 0; i < n_src; ++i)
        if (out_id[i] == o) st += S[i];
      total += (st * K[o] * N[o] + 3) / 4 * 4;
    }
    o0 = o1;
  }
  return total + WGT_FLOATS + 4;      // (+ one table for reduce-only outputs that did not fit the last one)
}

int launch_wgrad_group(int n_src, const float* const* A, const int* lda, const float* const* B, const int* ldb, const long long* P,
                       const unsigned* const* amax_a, const unsigned* const* amax_b, const int* out_id, int n_out,
                       float* const* out, const int* ldo, const int* K, const int* N, const int* accum, float* workspace,
                       int target_items, int n_extra, float* const* ex_out, const float* const* ex_partial, const int* ex_n,
                       const int* ex_S, const int* ex_stride, const int* ex_accum, const int* layout, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_group), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_group_wide), hipFuncAttributeMaxDynamicSharedMemorySize, WGP_LDS_WIDE);
    attr = true;
  }
  if (n_src > 65536) return NDJIR_ERR_ARG;
  for (int i = 0; i < n_src; ++i)
    if (out_id[i] < 0 || out_id[i] >= n_out) return NDJIR_ERR_ARG;
  for (int o = 0; o < n_out; ++o)
    if (K[o] > 32767 || N[o] > 32767) return NDJIR_ERR_UNSUPPORTED;
  int* S = (int*)alloca(sizeof(int) * n_src);
  long long* rows = (long long*)alloca(sizeof(long long) * n_src);
  static thread_local WggTable tab;          // host copy of the table being built (49 KB)
  long long off = 0;                         // running offset in the workspace (mirrors wgrad_group_workspace)
  // per table-full of outputs (a training step: one): writer launches, ONE k_wgrad_group launch, ONE reduction launch
  int ex_done = 0;
  for (int o0 = 0; o0 < n_out || ex_done < n_extra;) {
    const int o1 = o0 < n_out ? wgg_chunk_end(o0, n_src, P, out_id, n_out) : o0;
    if (o0 < n_out && o1 == o0) return NDJIR_ERR_UNSUPPORTED;       // one output with more sources than a table holds
    wgg_split_plan(o0, o1, n_src, A, lda, P, out_id, K, N, target_items, S, rows, layout);
    WggTable* dtab = reinterpret_cast<WggTable*>(workspace + off);
    off += WGT_FLOATS;
    int ns = 0, no = 0, n_amax = 0;
    static thread_local WggAmaxJobs amax_jobs;
    for (int o = o0; o < o1; ++o) {
      // the slabs of an output's sources are consecutive: the reduction sums S_total slabs of K * N floats
      if (!out[o] || ldo[o] < N[o]) return NDJIR_ERR_ARG;
      const long long kn = (long long)K[o] * N[o];
      int s_seen = 0;
      for (int i = 0; i < n_src; ++i) {
        if (out_id[i] != o || S[i] <= 0) continue;
        if (!A[i] || !B[i] || lda[i] < K[o] || ldb[i] < N[o]) return NDJIR_ERR_ARG;
        WggSrc& s = tab.src[ns++];
        s.A = A[i]; s.B = B[i]; s.amax_a = amax_a ? amax_a[i] : nullptr; s.amax_b = amax_b ? amax_b[i] : nullptr;
        s.partial = workspace + off + (long long)s_seen * kn;
        s_seen += S[i];
        s.P = P[i]; s.rows = rows[i]; s.lda = lda[i]; s.ldb = ldb[i]; s.K = K[o]; s.N = N[o]; s.S = S[i]; s.layout = layout ? (layout[i] & 3) : 0;
        if (!wgg_narrow(s.A, s.lda, s.K, s.N, s.layout)) {          // (narrow sources: fp32 items, no operand scale)
          if (!s.amax_a) amax_jobs.job[n_amax++] = (short)(2 * (ns - 1));
          if (!s.amax_b) amax_jobs.job[n_amax++] = (short)(2 * (ns - 1) + 1);
        }
      }
      WggOut& w = tab.out[no++];
      w.out = out[o]; w.partial = workspace + off; w.KN = (int)kn; w.N = N[o]; w.ldo = ldo[o];
      w.S = s_seen; w.accum = accum ? accum[o] : 0; w.pstride = (int)kn; w.pad = 0;
      off += ((long long)s_seen * kn + 3) / 4 * 4;
    }
    // segments: first of all the 128 x 256 items (another launch's: k_wgrad_group_wide), then the items of k_wgrad_group from the
    // longest to the shortest (tools/wgrad_items.py, profiles/r06_wgrad_items.txt: a 64 x 128 item or a strip of a ~3 100-point
    // split runs 160 - 250 us, a tile 150 - 175, a narrow output item of 128 points 10): the short ones fill the launch's tail.
    // Rounds 4 - 5 issued the narrow items first ("they finish under the tiles") and the strips last; with this order, the
    // layout order inside a kind and the split rounding of wgg_split_plan the two item launches take 819 + 490 us against
    // 905 + 652 (same box, same build; packed work 756 + 430 us of the 256 / 512 workgroup slots), the step 7.30 against 7.45 ms.
    // Also measured this round and NOT kept: the K / N remainders of <= 8 rows / columns (K = 259, 260, 262; N = 257: 360 strips
    // for 1 ... 6 useful rows each) as fp32 streaming items over the point-blocked operand, 64 features per item -- they re-read
    // 437 MB of operands the tiles of the same split have just read and run 100 - 145 us each: 7.47 ms against the strips' 7.43.
    int blocks = 0, nseg = 0, wide_blocks = 0;
    static int order[6] = {5, 4, 1, 2, 0, 3};
    static const bool order_env = [] {
      const char* e = getenv("NDJIR_WGG_ORDER");          // A/B switch, e.g. 530412 = the order of rounds 4 - 5
      if (e && strlen(e) == 6) for (int i = 0; i < 6; ++i) order[i] = e[i] - '0';
      return true;
    }();
    (void)order_env;
    for (int q = 0; q < 6; ++q) {
      if (q == 1) wide_blocks = blocks;
      // ... and inside a kind the sources with row-major operands first: 4-byte loads, 242 / 175 / 172 / 137 us per 128 x 256 item at
      // layout 0 / 1 / 2 / 3.  (In source order the 48 layout-0 items of the 16 384-point passes started 690 us into a 970-us launch
      // whose packed work is 725 us.)
      for (int pass = 0; pass < 4 * ns; ++pass) {
        const int lay_pass = pass / ns, i = pass - lay_pass * ns;
        const int kind = order[q];
        const WggSrc& s = tab.src[i];
        if ((s.layout & 3) != lay_pass) continue;
        WggRegion rg[5];
        int nr = 0;
        if (wgg_narrow(s.A, s.lda, s.K, s.N, s.layout)) {
          if (kind == 3) rg[nr++] = {3, 0, 0, s.K, s.N, 1, 1};
        } else {
          nr = wgg_regions(s.K, s.N, rg);
        }
        for (int t = 0; t < nr; ++t) {
          if (rg[t].kind != kind) continue;
          const int tiles = rg[t].tiles_k * rg[t].tiles_n;
          if (tiles <= 0) continue;
          if (nseg >= WGT_MAX_SEG || tiles > 32767) return NDJIR_ERR_UNSUPPORTED;
          WggSeg& g = tab.seg[nseg++];
          g.first = blocks; g.count = (s.S * tiles + 7) / 8 * 8; g.src = (short)i; g.kind = (short)kind;
          g.tiles = (short)tiles; g.tiles_n = (short)rg[t].tiles_n;
          g.k_off = (short)rg[t].k_off; g.n_off = (short)rg[t].n_off; g.k_end = (short)rg[t].k_end; g.n_end = (short)rg[t].n_end;
          blocks += g.count;
        }
      }
    }
    static const bool dump = getenv("NDJIR_WGG_DUMP") != nullptr;       // the plan of every grouped call, on stderr (tools/wgg_plan.py)
    if (dump) {
      fprintf(stderr, "wgg call: %d sources, %d outputs, %d segments, %d workgroups (%d wide)\n", ns, no, nseg, blocks, wide_blocks);
      for (int i = 0; i < ns; ++i)
        fprintf(stderr, "  src %d: K %d N %d P %lld S %d rows %lld lda %d ldb %d layout %d\n", i, tab.src[i].K, tab.src[i].N, tab.src[i].P,
                tab.src[i].S, tab.src[i].rows, tab.src[i].lda, tab.src[i].ldb, tab.src[i].layout);
      for (int i = 0; i < nseg; ++i)
        fprintf(stderr, "  seg %d: src %d kind %d tiles %d (%d per row) rows [%d, %d) cols [%d, %d) workgroups %d\n", i, tab.seg[i].src,
                tab.seg[i].kind, tab.seg[i].tiles, tab.seg[i].tiles_n, tab.seg[i].k_off, tab.seg[i].k_end, tab.seg[i].n_off,
                tab.seg[i].n_end, tab.seg[i].count);
    }
    // the reduce-only outputs (deferred bias gradients) ride in the tables' free output slots
    for (; ex_done < n_extra && no < WGT_MAX_OUT; ++ex_done) {
      WggOut& w = tab.out[no++];
      w.out = ex_out[ex_done]; w.partial = ex_partial[ex_done]; w.KN = ex_n[ex_done]; w.N = ex_n[ex_done]; w.ldo = ex_n[ex_done];
      w.S = ex_S[ex_done]; w.accum = ex_accum ? ex_accum[ex_done] : 0; w.pstride = ex_stride[ex_done]; w.pad = 0;
    }
    // The reduction's workgroups run in no particular order: two outputs that touch the same memory (a parameter and its
    // columns 1..: the packed first-order pass) must not be reduced by the same launch.  Generation of an output = 1 + the
    // largest generation among the EARLIER outputs it overlaps; outputs are sorted by generation, one reduction launch each.
    int gen[WGT_MAX_OUT], n_gen = 1;
    for (int i = 0; i < no; ++i) {
      const WggOut& w = tab.out[i];
      const int K_ = w.KN / w.N;
      const char* lo = reinterpret_cast<const char*>(w.out);
      const char* hi = lo + 4 * ((long long)(K_ - 1) * w.ldo + w.N);
      gen[i] = 0;
      for (int j = 0; j < i; ++j) {
        const WggOut& v = tab.out[j];
        const char* lo2 = reinterpret_cast<const char*>(v.out);
        const char* hi2 = lo2 + 4 * ((long long)(v.KN / v.N - 1) * v.ldo + v.N);
        if (lo < hi2 && lo2 < hi && gen[j] + 1 > gen[i]) gen[i] = gen[j] + 1;
      }
      if (gen[i] + 1 > n_gen) n_gen = gen[i] + 1;
    }
    if (n_gen > 1) {          // stable sort by generation (insertion: a handful of outputs move)
      for (int i = 1; i < no; ++i) {
        const WggOut w = tab.out[i];
        const int g = gen[i];
        int j = i;
        while (j > 0 && gen[j - 1] > g) { tab.out[j] = tab.out[j - 1]; gen[j] = gen[j - 1]; --j; }
        tab.out[j] = w; gen[j] = g;
      }
    }
    int rb = 0;
    int gen_first[WGT_MAX_OUT + 1];
    for (int g = 0; g <= n_gen; ++g) gen_first[g] = 0;
    for (int i = 0; i < no; ++i) {
      WggOut& w = tab.out[i];
      const bool vec = wgg_out_vec(w);
      if (i == 0 || gen[i] != gen[i - 1]) gen_first[gen[i]] = rb;
      w.first = rb;
      rb += (int)(((long long)w.KN + (vec ? 128 : 32) - 1) / (vec ? 128 : 32));
    }
    gen_first[n_gen] = rb;
    // the table -> device memory, a piece per launch
    for (int s0 = 0, g0 = 0, w0 = 0; s0 < ns || g0 < nseg || w0 < no;) {
      WggPiece pc{};
      pc.src0 = s0; pc.seg0 = g0; pc.out0 = w0; pc.tot_seg = nseg; pc.tot_out = no;
      pc.n_src = ns - s0 < WGG_MAX_SRC ? ns - s0 : WGG_MAX_SRC;
      pc.n_seg = nseg - g0 < WGG_MAX_SEG ? nseg - g0 : WGG_MAX_SEG;
      pc.n_out = no - w0 < WGG_MAX_OUT ? no - w0 : WGG_MAX_OUT;
      for (int i = 0; i < pc.n_src; ++i) pc.src[i] = tab.src[s0 + i];
      for (int i = 0; i < pc.n_seg; ++i) pc.seg[i] = tab.seg[g0 + i];
      for (int i = 0; i < pc.n_out; ++i) pc.out[i] = tab.out[w0 + i];
      hipLaunchKernelGGL(k_wgg_write, dim3(1), dim3(256), 0, stream, pc, dtab);
      if (ndjir_check_launch() != NDJIR_OK) return NDJIR_ERR_LAUNCH;
      s0 += pc.n_src; g0 += pc.n_seg; w0 += pc.n_out;
    }
    if (n_amax > 0) {
      hipLaunchKernelGGL(k_wgg_absmax, dim3(WGG_AMAX_CHUNKS, n_amax), dim3(256), 0, stream, dtab, amax_jobs);
      if (ndjir_check_launch() != NDJIR_OK) return NDJIR_ERR_LAUNCH;
    }
    if (wide_blocks > 0) {
      hipLaunchKernelGGL(k_wgrad_group_wide, dim3(wide_blocks), dim3(WG_THREADS), WGP_LDS_WIDE, stream, (const WggTable*)dtab);
      if (ndjir_check_launch() != NDJIR_OK) return NDJIR_ERR_LAUNCH;
    }
    if (blocks > wide_blocks) {
      hipLaunchKernelGGL(k_wgrad_group, dim3(blocks - wide_blocks), dim3(WG_THREADS), WGP_LDS, stream,
                         (const WggTable*)dtab, wide_blocks);
      if (ndjir_check_launch() != NDJIR_OK) return NDJIR_ERR_LAUNCH;
    }
    for (int g = 0; g < n_gen && no > 0; ++g) {
      const int nb = gen_first[g + 1] - gen_first[g];
      if (nb <= 0) continue;
      hipLaunchKernelGGL(k_wgrad_group_reduce, dim3(nb), dim3(256), 0, stream, (const WggTable*)dtab, gen_first[g]);
      if (ndjir_check_launch() != NDJIR_OK) return NDJIR_ERR_LAUNCH;
    }
    o0 = o1;
  }
  return NDJIR_OK;
}

}  // namespace ndjir

#ifdef WGG_ITEMLOG
extern "C" int ndjir_debug_wgrad_items(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ndjir::wgg_item_log), sizeof(long long) * 16384 * 3);
}
#endif
#ifdef WGG_TIMELINE
extern "C" int ndjir_debug_wgrad_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ndjir::wgg_stamps), sizeof(long long) * 2 * 64 * 12);
}
#endif
