// solver.hip -- the optimizer step of the training loop: Adam with the weight decay
// folded in, the gradient buffer re-armed in the same pass, and the finite-gradient guard as a
// device-side flag so that the whole step stays on the stream (no host round trip, graph-capturable).
//
// Reference: python/solver.py:29-30 (two `S.Adam`), :48-50 (`weight_decay`), :60-62 (`update`),
// :67-69 (`check_inf_or_nan_grad`), python/train.py:136-148 (order of the calls).  The arithmetic
// itself lives in nnabla 1.29.0 (not vendored); its published update rule is
//     m <- b1 m + (1-b1) g ;  v <- b2 v + (1-b2) g^2 ;  w <- w - alpha_t m / (sqrt(v) + eps)
//     alpha_t = alpha sqrt(1-b2^t) / (1-b1^t)                      (alpha_t is formed by the caller)
// and `weight_decay(d)` adds d*w to the gradient.  The reference applies the decay to the zeroed
// gradient buffer before backward accumulates into it; here the dense 2 GiB `d*w` pass is folded into
// the update (g_total = g + d*w): every cell is read and written once per step.
//
// HBM-bound streaming: per float 4 reads (w, g, m, v) + 3 writes (w, m, v) (+1 write when the
// gradient buffer is re-armed) = 28 (32) bytes; one float4 per lane, 4 independent float4 per thread in
// flight, non-temporal accesses (nothing is re-used, the 2 GiB streams must not evict the MLP
// weights from L2 / Infinity Cache).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

// fixed expression order, no fused multiply-add: bit-comparable with the CPU restatement used by the parity tests
#pragma clang fp contract(off)

namespace ndjir {

struct AdamHyper {
  float alpha_t, beta1, beta2, one_m_beta1, one_m_beta2, eps, decay;
};

__device__ __forceinline__ void adam1(float& w, float g, float& m, float& v, const AdamHyper& h) {
  float gt = g + h.decay * w;
  m = h.beta1 * m + h.one_m_beta1 * gt;
  v = h.beta2 * v + h.one_m_beta2 * gt * gt;
  w = w - h.alpha_t * m / (sqrtf(v) + h.eps);
}

typedef float f4 __attribute__((ext_vector_type(4)));

// Device-resident solver state (16 bytes): the learning rate the host schedule last wrote, nnabla's per-solver step
// counter t, the bias-corrected step size of the current step and whether the finite-gradient guard vetoed it.
// Keeping these on the device lets a captured graph replay the step with a changing learning rate and lets the
// guard skip an update (python/train.py:141-143: `continue` -- t is not advanced) without a host round trip.
struct AdamState {
  float alpha;
  int t;
  float alpha_t;
  int skipped;
};

// python/solver.py:67-69: the guards of the two solvers are combined with `and`
__global__ void k_adam_begin(AdamState* __restrict__ st, float beta1, float beta2, const int* __restrict__ flag_a,
                             const int* __restrict__ flag_b) {
  bool skip = false;
  if (flag_a && flag_b) skip = (*flag_a != 0) && (*flag_b != 0);
  else if (flag_a) skip = *flag_a != 0;
  else if (flag_b) skip = *flag_b != 0;
  st->skipped = skip ? 1 : 0;
  if (skip) return;
  int t = st->t + 1;
  st->t = t;
  // nnabla forms alpha_t with std::pow(float, uint32) -> double arithmetic, rounded once to float
  double a = (double)st->alpha * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  st->alpha_t = (float)a;
}

// BM: `touched` holds one bit per float4 of g (set for every float4 that may be non-zero): g is read -- and, with ZERO,
// cleared -- only there (the loss gradient of a voxel grid lives in < 1 % of the cells; everywhere else the step is the
// decay-only update), and the words consumed are cleared for the next step: 24 instead of 32 bytes per parameter.
// The 32 float4s of a bitmap word belong to 32 consecutive lanes of one wave, which all read the word before lane 0
// of the group clears it.
template <int UNROLL, bool ZERO, bool BM>
__global__ void __launch_bounds__(256) k_adam(long long n4, f4* __restrict__ w, f4* __restrict__ g, f4* __restrict__ m,
                                              f4* __restrict__ v, AdamHyper h, const AdamState* __restrict__ st,
                                              unsigned* __restrict__ touched) {
  long long base = ((long long)blockIdx.x * UNROLL) * 256 + threadIdx.x;
  unsigned bits[UNROLL];
  if (BM) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      long long i = base + (long long)u * 256;
      bits[u] = (i < n4) ? touched[i >> 5] : 0u;
    }
  }
  auto hit = [&](int u, long long i) { return !BM || ((bits[u] >> (i & 31)) & 1u); };
  if (st) {
    if (st->skipped) {            // vetoed step: parameters and moments stay, the gradient buffer is still re-armed
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        long long i = base + (long long)u * 256;
        if (i < n4) {
          if (ZERO && hit(u, i)) __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, g + i);
          if (BM && (i & 31) == 0 && bits[u]) touched[i >> 5] = 0u;
        }
      }
      return;
    }
    h.alpha_t = st->alpha_t;
  }
  f4 W[UNROLL], G[UNROLL], M[UNROLL], V[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    long long i = base + (long long)u * 256;
    if (i < n4) {
      W[u] = __builtin_nontemporal_load(w + i);
      G[u] = hit(u, i) ? __builtin_nontemporal_load(g + i) : f4{0.f, 0.f, 0.f, 0.f};
      M[u] = __builtin_nontemporal_load(m + i);
      V[u] = __builtin_nontemporal_load(v + i);
    }
  }
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    long long i = base + (long long)u * 256;
    if (i < n4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float wc = W[u][c], mc = M[u][c], vc = V[u][c];
        adam1(wc, G[u][c], mc, vc, h);
        W[u][c] = wc; M[u][c] = mc; V[u][c] = vc;
      }
      __builtin_nontemporal_store(W[u], w + i);
      __builtin_nontemporal_store(M[u], m + i);
      __builtin_nontemporal_store(V[u], v + i);
      if (ZERO && hit(u, i)) __builtin_nontemporal_store(f4{0.f, 0.f, 0.f, 0.f}, g + i);
      if (BM && (i & 31) == 0 && bits[u]) touched[i >> 5] = 0u;
    }
  }
}

template <bool ZERO>
__global__ void __launch_bounds__(256) k_adam_tail(long long start, long long n, float* __restrict__ w, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, AdamHyper h,
                                                   const AdamState* __restrict__ st) {
  long long i = start + blockIdx.x * 256 + threadIdx.x;
  if (st) {
    if (st->skipped) {
      if (ZERO && i < n) g[i] = 0.f;
      return;
    }
    h.alpha_t = st->alpha_t;
  }
  if (i < n) {
    float W = w[i], M = m[i], V = v[i];
    adam1(W, g[i], M, V, h);
    w[i] = W; m[i] = M; v[i] = V;
    if (ZERO) g[i] = 0.f;
  }
}

// ---- many small tensors in one launch (the MLP parameters: ~90 tensors, 1.46 M floats) --------------------------------------
constexpr int MT_MAX = 24;          // tensors per launch (kernarg struct stays < 1 KB)
constexpr int MT_CHUNK = 1024;      // floats per workgroup
struct MultiTensors {
  float* w[MT_MAX];
  const float* g[MT_MAX];
  float* m[MT_MAX];
  float* v[MT_MAX];
  int first_block[MT_MAX + 1];      // prefix sum of ceil(numel / MT_CHUNK)
  int numel[MT_MAX];
  int n;
};

__global__ void __launch_bounds__(256) k_adam_multi(MultiTensors t, AdamHyper h, const AdamState* __restrict__ st) {
  if (st) {
    if (st->skipped) return;
    h.alpha_t = st->alpha_t;
  }
  int blk = blockIdx.x, k = 0;
  while (k + 1 < t.n && blk >= t.first_block[k + 1]) ++k;
  int off = (blk - t.first_block[k]) * MT_CHUNK, n = t.numel[k];
  float* w = t.w[k]; const float* g = t.g[k]; float* m = t.m[k]; float* v = t.v[k];
#pragma unroll
  for (int u = 0; u < MT_CHUNK / 256; ++u) {
    int i = off + u * 256 + threadIdx.x;
    if (i < n) {
      float W = w[i], M = m[i], V = v[i];
      adam1(W, g ? g[i] : 0.f, M, V, h);
      w[i] = W; m[i] = M; v[i] = V;
    }
  }
}

__global__ void __launch_bounds__(256) k_nonfinite_multi(MultiTensors t, int* __restrict__ flag) {
  int blk = blockIdx.x, k = 0;
  while (k + 1 < t.n && blk >= t.first_block[k + 1]) ++k;
  int off = (blk - t.first_block[k]) * MT_CHUNK, n = t.numel[k];
  const float* g = t.g[k];
  if (!g) return;
  bool bad = false;
#pragma unroll
  for (int u = 0; u < MT_CHUNK / 256; ++u) {
    int i = off + u * 256 + threadIdx.x;
    if (i < n) bad |= !isfinite(g[i]);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// dense: flag |= any(!isfinite(g)); also the sum of squares for clip_grad_by_norm (python/solver.py:53-58)
__global__ void __launch_bounds__(256) k_nonfinite(long long n, const float* __restrict__ g, int* __restrict__ flag) {
  bool bad = false;
  long long n4 = n / 4;
  const f4* g4 = reinterpret_cast<const f4*>(g);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f4 x = __builtin_nontemporal_load(g4 + i);
    bad |= !(isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]) && isfinite(x[3]));
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) bad |= !isfinite(g[n4 * 4 + threadIdx.x]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void __launch_bounds__(256) k_sumsq(long long n, const float* __restrict__ x, double* __restrict__ out) {
  __shared__ double part[4];
  double acc = 0.0;
  long long n4 = n / 4;
  const f4* x4 = reinterpret_cast<const f4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f4 a = __builtin_nontemporal_load(x4 + i);
    acc += (double)(a[0] * a[0] + a[1] * a[1]) + (double)(a[2] * a[2] + a[3] * a[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) { float a = x[n4 * 4 + threadIdx.x]; acc += (double)a * a; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

static AdamHyper hyper(float alpha_t, float beta1, float beta2, float eps, float decay) {
  AdamHyper h;
  h.alpha_t = alpha_t; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.decay = decay;
  h.one_m_beta1 = 1.f - beta1; h.one_m_beta2 = 1.f - beta2;
  return h;
}

static int fill_multi(MultiTensors& t, int n, float* const* w, const float* const* g, float* const* m, float* const* v,
                      const long long* numel) {
  t.n = n;
  int blocks = 0;
  for (int k = 0; k < n; ++k) {
    if (numel[k] <= 0 || numel[k] > 0x7fffffffLL) return -1;
    t.w[k] = w ? w[k] : nullptr; t.g[k] = g[k]; t.m[k] = m ? m[k] : nullptr; t.v[k] = v ? v[k] : nullptr;
    t.numel[k] = (int)numel[k];
    t.first_block[k] = blocks;
    blocks += (int)((numel[k] + MT_CHUNK - 1) / MT_CHUNK);
  }
  t.first_block[n] = blocks;
  return blocks;
}

}  // namespace ndjir

using namespace ndjir;

// Start of a solver's update: raises `skipped` when the guard flags veto the step, otherwise advances t and forms
// alpha_t from the learning rate stored in the state.  flags: device ints written by the check functions (may be null).
extern "C" int ndjir_solver_adam_begin(void* state, float beta1, float beta2, const int* flag_a, const int* flag_b,
                                       hipStream_t stream) {
  if (!state) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_adam_begin, dim3(1), dim3(1), 0, stream, (AdamState*)state, beta1, beta2, flag_a, flag_b);
  return ndjir_check_launch();
}

// One Adam step over a dense parameter (the 2 GiB voxel grid): w, g, m, v of n floats (16-byte aligned).
// decay: weight-decay rate folded into the gradient; zero_grad != 0 re-arms g (writes zeros) in the same
// pass.  state (may be null): device state prepared by ndjir_solver_adam_begin -- its alpha_t replaces the
// argument and a vetoed step only re-arms g.
extern "C" int ndjir_solver_adam(long long n, float* w, float* g, float* m, float* v, float alpha_t, float beta1,
                                 float beta2, float eps, float decay, int zero_grad, const void* state,
                                 hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!w || !g || !m || !v) return NDJIR_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15) return NDJIR_ERR_ARG;
  AdamHyper h = hyper(alpha_t, beta1, beta2, eps, decay);
  const AdamState* st = (const AdamState*)state;
  constexpr int U = 4;
  long long n4 = n / 4;
  if (n4 > 0) {
    long long blocks = (n4 + 256 * U - 1) / (256 * U);
    if (blocks > 0x7fffffffLL) return NDJIR_ERR_ARG;
    if (zero_grad)
      hipLaunchKernelGGL((k_adam<U, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, n4, (f4*)w, (f4*)g, (f4*)m, (f4*)v, h, st, nullptr);
    else
      hipLaunchKernelGGL((k_adam<U, false, false>), dim3((unsigned)blocks), dim3(256), 0, stream, n4, (f4*)w, (f4*)g, (f4*)m, (f4*)v, h, st, nullptr);
  }
  if (n4 * 4 < n) {
    if (zero_grad) hipLaunchKernelGGL((k_adam_tail<true>), dim3(1), dim3(256), 0, stream, n4 * 4, n, w, g, m, v, h, st);
    else hipLaunchKernelGGL((k_adam_tail<false>), dim3(1), dim3(256), 0, stream, n4 * 4, n, w, g, m, v, h, st);
  }
  return ndjir_check_launch();
}

// As ndjir_solver_adam with zero_grad = 1, for a gradient that is non-zero only where `touched` (1 bit per float4 of g, n / 4
// bits rounded up to whole 32-bit words; n % 128 == 0) says so: g is read and cleared only there and the bitmap comes back zero.
extern "C" int ndjir_solver_adam_touched(long long n, float* w, float* g, float* m, float* v, float alpha_t, float beta1,
                                         float beta2, float eps, float decay, unsigned* touched, const void* state,
                                         hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!w || !g || !m || !v || !touched || (n & 127)) return NDJIR_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15) return NDJIR_ERR_ARG;
  AdamHyper h = hyper(alpha_t, beta1, beta2, eps, decay);
  constexpr int U = 4;
  long long n4 = n / 4, blocks = (n4 + 256 * U - 1) / (256 * U);
  if (blocks > 0x7fffffffLL) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL((k_adam<U, true, true>), dim3((unsigned)blocks), dim3(256), 0, stream, n4, (f4*)w, (f4*)g, (f4*)m, (f4*)v, h,
                     (const AdamState*)state, touched);
  return ndjir_check_launch();
}

// The same step over `count` small tensors (host arrays of device pointers; g[k] may be null = zero
// gradient, e.g. a parameter the loss does not reach).  ceil(count / 24) launches.
extern "C" int ndjir_solver_adam_multi(int count, float* const* w, const float* const* g, float* const* m,
                                       float* const* v, const long long* numel, float alpha_t, float beta1, float beta2,
                                       float eps, float decay, const void* state, hipStream_t stream) {
  if (count <= 0) return NDJIR_OK;
  if (!w || !g || !m || !v || !numel) return NDJIR_ERR_ARG;
  AdamHyper h = hyper(alpha_t, beta1, beta2, eps, decay);
  for (int s = 0; s < count; s += MT_MAX) {
    int n = count - s < MT_MAX ? count - s : MT_MAX;
    for (int k = 0; k < n; ++k) if (!w[s + k] || !m[s + k] || !v[s + k]) return NDJIR_ERR_ARG;
    MultiTensors t;
    int blocks = fill_multi(t, n, w + s, g + s, m + s, v + s, numel + s);
    if (blocks < 0) return NDJIR_ERR_ARG;
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, stream, t, h, (const AdamState*)state);
  }
  return ndjir_check_launch();
}

// flag |= 1 if any gradient value is inf or nan (flag: device int, zeroed by the caller).
extern "C" int ndjir_solver_check_inf_or_nan(long long n, const float* g, int* flag, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!g || !flag || (reinterpret_cast<uintptr_t>(g) & 15)) return NDJIR_ERR_ARG;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_nonfinite, dim3((unsigned)blocks), dim3(256), 0, stream, n, g, flag);
  return ndjir_check_launch();
}

extern "C" int ndjir_solver_check_inf_or_nan_multi(int count, const float* const* g, const long long* numel, int* flag,
                                                   hipStream_t stream) {
  if (count <= 0) return NDJIR_OK;
  if (!g || !numel || !flag) return NDJIR_ERR_ARG;
  for (int s = 0; s < count; s += MT_MAX) {
    int n = count - s < MT_MAX ? count - s : MT_MAX;
    MultiTensors t;
    int blocks = fill_multi(t, n, nullptr, g + s, nullptr, nullptr, numel + s);
    if (blocks < 0) return NDJIR_ERR_ARG;
    hipLaunchKernelGGL(k_nonfinite_multi, dim3(blocks), dim3(256), 0, stream, t, flag);
  }
  return ndjir_check_launch();
}

// python/train.py:144-146: `if np.any(np.isnan(loss.d)): continue`.  Raises BOTH guard flags when x holds a NaN, so that
// the `and` of ndjir_solver_adam_begin vetoes the update whatever the gradient checks found.  n is tiny (the loss).
__global__ void k_veto_if_nan(int n, const float* __restrict__ x, int* __restrict__ flag_a, int* __restrict__ flag_b) {
  bool bad = false;
  for (int i = threadIdx.x; i < n; i += 64) bad |= (x[i] != x[i]);
  if (__any(bad) && threadIdx.x == 0) { *flag_a = 1; *flag_b = 1; }
}

extern "C" int ndjir_solver_veto_if_nan(int n, const float* x, int* flag_a, int* flag_b, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!x || !flag_a || !flag_b) return NDJIR_ERR_ARG;
  hipLaunchKernelGGL(k_veto_if_nan, dim3(1), dim3(64), 0, stream, n, x, flag_a, flag_b);
  return ndjir_check_launch();
}

// *out += sum(x^2) (device double, zeroed by the caller): the norm `clip_grad_by_norm` needs
// (python/solver.py:53-58; called on the decay-only gradient d*w, python/train.py:138-139).
extern "C" int ndjir_solver_sum_squares(long long n, const float* x, double* out, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!x || !out || (reinterpret_cast<uintptr_t>(x) & 15)) return NDJIR_ERR_ARG;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_sumsq, dim3((unsigned)blocks), dim3(256), 0, stream, n, x, out);
  return ndjir_check_launch();
}
