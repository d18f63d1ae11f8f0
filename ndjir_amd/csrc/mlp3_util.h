// mlp3_util.h -- what the two f16x3 chain kernels share (mlp3.hip: 32 / 64-point tiles with a staged epilogue; mlp3w.hip:
// 128-point tiles with the epilogue in the accumulator registers): vector types, SGPR pinning, the power-of-two scaling, the
// two-way f16 split, and -- the reason this header exists -- ONE spelling of every forward expression, with floating-point
// contraction switched off and the fused multiply-adds written out.  A point's forward result must not depend on which
// kernel, tile height or code path evaluated it (the sampler evaluates the SDF of new samples only and merges them with
// earlier evaluations bit for bit; tests/test_gpu_extract.py::test_volume_is_batch_invariant), so neither kernel may leave
// the choice between a*b+c and fma(a,b,c) to the optimiser.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndjir {
namespace x3u {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ int pin(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ float pin(float v) { return __int_as_float(pin(__float_as_int(v))); }
template <class T>
using gptr = T __attribute__((address_space(1)))*;
template <class T>
__device__ __forceinline__ gptr<T> pin(T* p) {
  asm volatile("" : "+s"(p));
  return (gptr<T>)p;
}

// Power-of-two scale that puts a group whose largest finite magnitude has bit pattern `mbits` at [2^14, 2^15);
// inv = 1 / s exactly.  An all-zero group gets a large harmless scale.
__host__ __device__ __forceinline__ void scale_from_max(unsigned mbits, float& s, float& inv) {
  int E = (int)(mbits >> 23);
  if (E < 1) E = 1;
  int se = 268 - E;                  // biased exponent of s: max * s = 1.x * 2^14
  if (se > 253) se = 253;
  if (se < 1) se = 1;
  union { int i; float f; } a, b;
  a.i = se << 23;
  b.i = (254 - se) << 23;
  s = a.f;
  inv = b.f;
}

// |v| as ordered bits, Inf / NaN ignored (they stay confined to their own rows; the group's scale is taken from
// the finite values)
__device__ __forceinline__ unsigned finite_abs_bits(float v) {
  const unsigned b = __float_as_uint(v) & 0x7fffffffu;
  return b < 0x7f800000u ? b : 0u;
}

__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  return m;
}

// lane-crossing without the LDS crossbar (ds_bpermute + lgkmcnt round trip): DPP controls of gfx9
//   quad_perm [1,0,3,2] = xor 1, quad_perm [2,3,0,1] = xor 2, row_half_mirror (i -> 7 - i inside a group of 8 lanes),
//   row_mirror (i -> 15 - i inside a row of 16 lanes), row_ror:8 = xor 8 inside a row of 16 lanes
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;

// a 16-byte vector at 4-byte alignment: rows that start at an odd float offset (the geometric net's output inside Z, 257-wide
// gradients) still move as dwordx4 -- correct on gfx950 and within 10 % of the aligned rate (tools/ubench/unaligned.hip)
typedef f32x4 f32x4u __attribute__((aligned(4)));

// 4 x 4 transpose inside every quad of lanes: v[g] (g = 0..3, a 16-byte chunk) of lane j (= lane & 3) becomes the old v[j] of lane g.
// Two butterfly stages (lanes j ^ 1 with chunk pairs (0,1) (2,3); lanes j ^ 2 with (0,2) (1,3)), one DPP move and two selects per
// dword and stage.  An involution.  mlp3w.hip hands its accumulator blocks to memory through it: before, a lane holds four chunks
// of ONE point (every load / store instruction touches 32 rows x 32 bytes); after, chunk j of four consecutive points, so that the
// eight lanes of a point's two quads cover its full 128-byte row in one instruction (8 rows x 128 bytes) -- 3 x what the CU's memory
// path moves in the first shape (tools/ubench/sidepat.hip: 15 vs 37-48 B / clk / CU).
__device__ __forceinline__ void quad_swap(f32x4& a, f32x4& b, bool odd, bool stage2) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float xa = a[q], xb = b[q];
    const float x = odd ? xa : xb;                  // what the partner lane needs
    const float y = stage2 ? dpp<DPP_XOR2>(x) : dpp<DPP_XOR1>(x);
    a[q] = odd ? y : xa;
    b[q] = odd ? xb : y;
  }
}
__device__ __forceinline__ void quad_transpose(f32x4 (&v)[4], int lane) {
  const bool o1 = lane & 1, o2 = lane & 2;
  quad_swap(v[0], v[1], o1, false);
  quad_swap(v[2], v[3], o1, false);
  quad_swap(v[0], v[2], o2, true);
  quad_swap(v[1], v[3], o2, true);
}

// ---- the two-way split:  x s = hi + lo 2^-11 -------------------------------------------------------------------------------
__device__ __forceinline__ void split4(f32x4 v, float s, f16x4& ph, f16x4& pl) {
#pragma clang fp contract(off)
  const f32x4 xs = v * s;
  ph = __builtin_convertvector(xs, f16x4);
#ifndef NDJIR_SPLIT_MIX
  const f32x4 res = (xs - __builtin_convertvector(ph, f32x4)) * LO_SCALE;
  pl = __builtin_convertvector(res, f16x4);
#else
  // lo = f16((xs - hi) 2^11) = f16(fma(hi, -2^11, xs 2^11)) -- the same value (xs - hi is exact, and so are both products): one
  // mixed-precision FMA per element that reads hi where it lies, as an f16 half of the packed pair, and writes its half of the
  // packed result (v_fma_mixlo / mixhi_f16), instead of convert back + subtract + multiply + convert + pack: 2.5 vector
  // instructions per element for the split instead of 4.  OPT-IN (-DNDJIR_SPLIT_MIX): bit-identical results (the step-parity
  // and fp64-accuracy suites pass on it), but the step measured 0.4 % SLOWER on it in three same-box pairs (round 5) -- phase B
  // is not bound by its vector instructions, and four opaque asm statements per group cost the scheduler its freedom.
  // (xs 2^11 from xs, not v (s 2^11): an all-zero row carries the largest scale, whose product with 2^11 is inf.)
  const f32x4 x2 = xs * LO_SCALE;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 hp = __builtin_bit_cast(u32x2, ph);
  const float m = -LO_SCALE;
  unsigned l01, l23;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(hp[0]), "v"(m), "v"(x2[0]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(hp[0]), "v"(m), "v"(x2[1]));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(hp[1]), "v"(m), "v"(x2[2]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(hp[1]), "v"(m), "v"(x2[3]));
  pl = __builtin_bit_cast(f16x4, u32x2{l01, l23});
#endif
}
__device__ __forceinline__ void split1(float v, float s, _Float16& ph, _Float16& pl) {
#pragma clang fp contract(off)
  const float xs = v * s;
  ph = (_Float16)xs;
  pl = (_Float16)((xs - (float)ph) * LO_SCALE);
}

// ---- forward expressions (one spelling for every kernel and code path) -----------------------------------------------------
// the two accumulators of a product:  hi hi'  +  (hi lo' + lo hi') 2^-11
__device__ __forceinline__ float acc_sum(float acc0, float acc1) { return __builtin_fmaf(acc1, LO_INV, acc0); }

// hidden layer: u = beta log2(e) * (pre-activation);  kk = (1 / row scale) * ((1 / column-block scale) * beta log2(e)),
// bb = bias * beta log2(e)
__device__ __forceinline__ float fwd_u(float pv, float kk, float bb) { return __builtin_fmaf(pv, kk, bb); }
__device__ __forceinline__ float fwd_kk(float row_inv, float winv, float b2) {
#pragma clang fp contract(off)
  return row_inv * (winv * b2);
}
// + per-row-group term of the first layer
__device__ __forceinline__ float fwd_u_rowbias(float u, float rb, float b2) { return __builtin_fmaf(rb, b2, u); }
// softplus_beta(t) = (max(u, 0) + log2(1 + 2^-|u|)) ln2 / beta,  u = beta log2(e) t;  ib2sc = ln2 / beta * (skip scale)
__device__ __forceinline__ float softplus_u(float u, float ib2sc) {
#pragma clang fp contract(off)
  const float l2 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-__builtin_fabsf(u)));
  return (__builtin_fmaxf(u, 0.f) + l2) * ib2sc;
}
// output layer / narrow output layer:  z = pv (1 / row scale) (1 / column-block scale)  [+ bias]
__device__ __forceinline__ float out_z(float pv, float row_inv, float winv) {
#pragma clang fp contract(off)
  return (pv * row_inv) * winv;
}
__device__ __forceinline__ float out_add(float z, float b) {
#pragma clang fp contract(off)
  return z + b;
}

}  // namespace x3u
}  // namespace ndjir
