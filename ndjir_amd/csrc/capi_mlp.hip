// capi_mlp.hip -- C ABI of the fused MLP engine (declarations: include/ndjir_hip.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/ndjir_hip.h"
#include "common.h"
#include "mlp.h"

using namespace ndjir;

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Matrix arithmetic of the chain engine: NDJIR_MATH_FP32 = v_mfma_f32_32x32x2_f32 (mlp.hip),
// NDJIR_MATH_BF16X6 = three-way bf16 split, six v_mfma_f32_32x32x16_bf16 partial products (mlp6.hip),
// NDJIR_MATH_F16X3 = scaled two-way f16 split, three v_mfma_f32_32x32x16_f16 partial products (mlp3.hip; default):
// all at fp32-FMA-chain accuracy or better, at 16/16, 6/16 and 3/16 of the fp32 matrix time.  Packed weights are
// specific to the mode that was active when they were packed.
static int g_math = NDJIR_MATH_F16X3;

extern "C" int ndjir_mlp_set_math(int math) {
  if (math != NDJIR_MATH_FP32 && math != NDJIR_MATH_BF16X6 && math != NDJIR_MATH_F16X3) return NDJIR_ERR_ARG;
  g_math = math;
  return NDJIR_OK;
}

extern "C" int ndjir_mlp_get_math(void) { return g_math; }

extern "C" long long ndjir_mlp_packed_size(int K, int N, int transpose) {
  if (g_math == NDJIR_MATH_BF16X6) return packed_size6(K, N, transpose);
  if (g_math == NDJIR_MATH_F16X3) return packed_size3(K, N, transpose);
  int Kp = round_up(transpose ? N : K, 8), Np = round_up(transpose ? K : N, 32);
  return (long long)Kp * Np;
}

extern "C" int ndjir_mlp_pack(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream) {
  if (!W || !dst || K <= 0 || N <= 0) return NDJIR_ERR_ARG;
  if (g_math == NDJIR_MATH_BF16X6) return launch_pack6(W, dst, K, N, transpose, stream);
  if (g_math == NDJIR_MATH_F16X3) return launch_pack3(W, N, dst, K, N, transpose, stream);
  return launch_pack(W, dst, K, N, transpose, stream);
}

// f16x3 arithmetic only: W (K, N) with row stride ldw >= N (a column slice of a wider matrix, packed without a copy)
extern "C" int ndjir_mlp_pack_strided(const float* W, int ldw, float* dst, int K, int N, int transpose, hipStream_t stream) {
  if (!W || !dst || K <= 0 || N <= 0 || ldw < N) return NDJIR_ERR_ARG;
  if (g_math != NDJIR_MATH_F16X3) return NDJIR_ERR_UNSUPPORTED;
  return launch_pack3(W, ldw, dst, K, N, transpose, stream);
}

// f16x3 arithmetic only: re-pack many matrices in one launch.  `table`: DEVICE array of n entries of
// ndjir_mlp_pack_entry_bytes() bytes each -- { const float* W; float* dst; int K, N, ldw, transpose, Kp, Np, first_block, pad } with
// Kp / Np the padded dims of the packed matrix (ndjir_mlp_packed_dims) and first_block the running sum of Np / 32;
// total_blocks = that sum over all entries.
extern "C" int ndjir_mlp_pack_table(const void* table, int n, int total_blocks, hipStream_t stream) {
  if (n <= 0) return NDJIR_OK;
  if (!table || total_blocks <= 0) return NDJIR_ERR_ARG;
  if (g_math != NDJIR_MATH_F16X3) return NDJIR_ERR_UNSUPPORTED;
  return launch_pack3_table(static_cast<const PackEntry*>(table), n, total_blocks, stream);
}
extern "C" int ndjir_mlp_pack_entry_bytes(void) { return (int)sizeof(PackEntry); }

static long long* g_timeline = nullptr;   // diagnostics only, see ndjir_mlp_debug_timeline

// ---- chain groups: between ndjir_mlp_chain_group_begin and _end, this thread's chain calls are recorded, not launched ----
constexpr int GROUP_MAX_CALLS = 16;
struct GroupCall { ChainArgs a; int mode; };
static thread_local GroupCall* g_group = nullptr;     // non-null = a group is open on this thread
static thread_local int g_group_n = 0;
static thread_local int g_group_err = NDJIR_OK;
static thread_local ChainDry* g_dry = nullptr;         // set for the duration of the two dry-run queries below

// Points per workgroup tile of the chain kernels: 0 = chosen per launch (128 for large launches the wide-tile kernel
// supports, else 64, 32 for small launches); 32 / 64 / 128 force one (128: where supported).  Results do not depend on it.
static int g_tile_rows = -1;
static int tile_rows_setting() {
  if (g_tile_rows < 0) { const char* e = getenv("NDJIR_MLP_TILE"); g_tile_rows = e ? atoi(e) : 0; }
  return g_tile_rows;
}
extern "C" int ndjir_mlp_set_tile_rows(int rows) {
  if (rows != 0 && rows != 32 && rows != 64 && rows != 128) return NDJIR_ERR_ARG;
  g_tile_rows = rows;
  return NDJIR_OK;
}
extern "C" int ndjir_mlp_get_tile_rows(void) { return tile_rows_setting(); }

// Which training-pass chain launches of nets wider than 128 columns run on the software-pipelined kernel (csrc/mlp3p.hip):
// bit 0 forward, bit 1 backward, bit 2 tangent.  Results agree with the other kernels to round-off (one fp32 accumulator per
// block instead of two).  Process-wide, like the engine and tile settings: not to be changed under a live captured graph.
extern "C" int ndjir_mlp_set_chain_pipeline(int mask) {
  if (mask < 0 || mask > 7) return NDJIR_ERR_ARG;
  chain_pipeline(mask);
  return NDJIR_OK;
}
extern "C" int ndjir_mlp_get_chain_pipeline(void) { return chain_pipeline(-1); }

static int chain_impl(int bwd, long long P, const float* X, int ldx, int K0, int L,
                               const float* const* Wp, const float* const* bias, const int* Ks, const int* Ns,
                               const float* const* side_in, float* const* side_out, const int* ld_side,
                               float* const* bgrad, float* Y, int ldy, int accum_y, int has_output, float beta,
                               int skip_layer, float skip_scale, int skip_split, float* Xskip, int ld_xskip,
                               const float* const* side_in2, const float* const* side_add, float* const* side_out2,
                               const float* row_bias, int row_bias_div, float* in_bgrad, float* workspace,
                               unsigned* const* side_amax, unsigned* x_amax, hipStream_t stream) {
  if (P <= 0) {
    if (bwd != 0 && in_bgrad && K0 > 0 && hipMemsetAsync(in_bgrad, 0, (size_t)K0 * sizeof(float), stream) != hipSuccess)
      return NDJIR_ERR_LAUNCH;
    // nothing to process: bias gradients of an empty batch are zero
    if (bwd != 0 && bgrad && Ns && L >= 1 && L <= MAX_CHAIN_LAYERS)
      for (int i = 0; i < L; ++i)
        if (bgrad[i] && hipMemsetAsync(bgrad[i], 0, (size_t)Ns[i] * sizeof(float), stream) != hipSuccess) return NDJIR_ERR_LAUNCH;
    return NDJIR_OK;
  }
  if (!X || L < 1 || L > MAX_CHAIN_LAYERS || !Wp || !Ks || !Ns) return NDJIR_ERR_ARG;
  if (has_output && !Y) return NDJIR_ERR_ARG;
  ChainArgs a{};
  // 128-point tiles (mlp3w.hip) for large launches, else 64-point tiles; 32-point tiles when 64 would leave CUs without a
  // tile (small launches such as the sampler's 16-samples-per-ray rounds).  Results do not depend on the tile height.
  // NDJIR_MLP_TILE / ndjir_mlp_set_tile_rows force one.
  const int forced = tile_rows_setting();
  a.tile_rows = forced ? forced : ((P + 63) / 64 < 256 ? 32 : 64);
  a.forced_tile = forced;
  a.timeline = g_timeline;
  a.dry = g_dry;
  a.bg_partial = workspace;
  if (row_bias && (bwd != 0 || L < 2 || row_bias_div < 1)) return NDJIR_ERR_ARG;
  a.row_bias = row_bias; a.row_bias_div = row_bias_div;
  a.in_bgrad = (bwd != 0) ? in_bgrad : nullptr;
  a.x_amax = x_amax;
  a.P = P; a.X = X; a.ldx = ldx; a.K0 = K0; a.K0p = round_up(K0, 8); a.L = L;
  a.Y = Y; a.ldy = ldy; a.accum_y = accum_y & 1; a.bg_accum = (accum_y >> 1) & 1; a.defer_bg_reduce = (accum_y >> 2) & 1;
  a.side_blocked = (accum_y >> 3) & 1;
  if (a.side_blocked && (g_math != NDJIR_MATH_F16X3 || (P & 31) != 0)) return NDJIR_ERR_UNSUPPORTED;
  a.has_output = has_output; a.beta = beta;
  a.skip_layer = skip_layer; a.skip_scale = skip_scale; a.skip_split = skip_split; a.Xskip = Xskip; a.ld_xskip = ld_xskip;
  int kin = K0;
  for (int i = 0; i < L; ++i) {
    ChainLayer& ly = a.layers[i];
    if (!Wp[i] || Ks[i] <= 0 || Ns[i] <= 0) return NDJIR_ERR_ARG;
    ly.Wp = Wp[i];
    ly.bias = bias ? bias[i] : nullptr;
    ly.side_in = side_in ? side_in[i] : nullptr;
    ly.side_out = side_out ? side_out[i] : nullptr;
    ly.bgrad = bgrad ? bgrad[i] : nullptr;
    ly.side_in2 = side_in2 ? side_in2[i] : nullptr;
    ly.side_add = side_add ? side_add[i] : nullptr;
    ly.side_out2 = side_out2 ? side_out2[i] : nullptr;
    ly.side_amax = side_amax ? side_amax[i] : nullptr;
    ly.K = Ks[i]; ly.N = Ns[i]; ly.Kp = round_up(Ks[i], 8); ly.Np = round_up(Ns[i], 32);
    ly.ld_side = ld_side ? ld_side[i] : Ns[i];
    const bool last = has_output && (i == L - 1);
    // the input width of layer i must match what the previous epilogue leaves in LDS
    int expect = (i == 0) ? K0 : kin;
    if (ly.K != expect) return NDJIR_ERR_ARG;
    if (ly.Np == 32 && !last) return NDJIR_ERR_UNSUPPORTED;   // narrow layers only as the output layer
    if (bwd != 0 && !last && !ly.side_in) return NDJIR_ERR_ARG;
    kin = ly.N;
    if (bwd != 1 && i == skip_layer) kin = ly.N + K0;
    if (bwd == 1 && i == skip_layer) kin = skip_split;
  }
  if (g_group && !g_dry) {                 // an open group: the launch happens in ndjir_mlp_chain_group_end
    if (g_group_n >= GROUP_MAX_CALLS) return g_group_err = NDJIR_ERR_ARG;
    g_group[g_group_n].a = a;
    g_group[g_group_n].mode = bwd;
    ++g_group_n;
    return NDJIR_OK;
  }
  if (g_math == NDJIR_MATH_F16X3) return launch_chain3(a, bwd, stream);
  if (g_math == NDJIR_MATH_BF16X6) return launch_chain6(a, bwd, stream);
  return launch_chain(a, bwd, stream);
}

static int chain_dispatch(const ChainArgs& a, int mode, hipStream_t stream) {
  if (g_math == NDJIR_MATH_F16X3) return launch_chain3(a, mode, stream);
  if (g_math == NDJIR_MATH_BF16X6) return launch_chain6(a, mode, stream);
  return launch_chain(a, mode, stream);
}

// Chain groups.  The per-sample material nets of the reference (python/renderer.py:113-128: base colour, implicit
// illumination, roughness, specular reflectance, photogrammetric light) read the same packed row of every sample and their
// input gradients add up in one tensor.  A caller brackets their chain calls:
//     ndjir_mlp_chain_group_begin();  ndjir_mlp_chain(...) x n  (recorded, nothing is launched);  ndjir_mlp_chain_group_end(stream, &launches);
// _end launches the recorded calls IN ORDER -- consecutive calls of one mode that the 128-point-tile kernel can run as one
// launch (f16x3 arithmetic, same number of points, same tile shape, at most MAX_GROUP_NETS, the accumulators of all fit the
// LDS beside the planes) as ONE launch in which a workgroup takes its tile through the nets in turn, everything else exactly
// as the calls alone would have run.  Results are those of the separate launches, bit for bit (same kernel, same per-tile
// arithmetic; a backward group's `accum_y` calls add to the gradient tile in call order).  The pointer arrays passed to the
// recorded calls need not outlive them.  Thread-local; a group must be closed on the thread that opened it.
extern "C" int ndjir_mlp_chain_group_begin(void) {
  if (g_group) return NDJIR_ERR_ARG;
  g_group = static_cast<GroupCall*>(malloc(sizeof(GroupCall) * GROUP_MAX_CALLS));
  if (!g_group) return NDJIR_ERR_LAUNCH;
  g_group_n = 0;
  g_group_err = NDJIR_OK;
  return NDJIR_OK;
}

extern "C" int ndjir_mlp_chain_group_end(hipStream_t stream, int* launches) {
  if (!g_group) return NDJIR_ERR_ARG;
  GroupCall* calls = g_group;
  const int n = g_group_n;
  g_group = nullptr;                    // (closed whatever happens below)
  g_group_n = 0;
  int rc = g_group_err, issued = 0;
  ChainArgs run[MAX_GROUP_NETS];
  for (int i = 0; i < n && rc == NDJIR_OK;) {
    int m = 1;
    const ChainArgs& f = calls[i].a;
    const bool wide = g_math == NDJIR_MATH_F16X3 && (f.tile_rows == 128 || (f.tile_rows == 64 && f.forced_tile == 0));
    if (wide)
      while (i + m < n && m < MAX_GROUP_NETS && calls[i + m].mode == calls[i].mode && calls[i + m].a.P == f.P &&
             calls[i + m].a.tile_rows == f.tile_rows && calls[i + m].a.forced_tile == f.forced_tile) ++m;
    // the longest prefix of the run the kernel takes as one launch
    int done = 0;
    for (int k = m; k >= 2 && !done; --k) {
      for (int j = 0; j < k; ++j) run[j] = calls[i + j].a;
      const int r = launch_chainw_group(run, k, calls[i].mode, stream);
      if (r == NDJIR_OK) { done = k; ++issued; }
      else if (r != NDJIR_ERR_UNSUPPORTED) rc = r;
      if (rc != NDJIR_OK) break;
    }
    if (rc != NDJIR_OK) break;
    if (!done) { rc = chain_dispatch(f, calls[i].mode, stream); done = 1; ++issued; }
    i += done;
  }
  free(calls);
  if (launches) *launches = issued;
  return rc;
}

extern "C" int ndjir_mlp_chain(int bwd, long long P, const float* X, int ldx, int K0, int L,
                               const float* const* Wp, const float* const* bias, const int* Ks, const int* Ns,
                               const float* const* side_in, float* const* side_out, const int* ld_side,
                               float* const* bgrad, float* Y, int ldy, int accum_y, int has_output, float beta,
                               int skip_layer, float skip_scale, int skip_split, float* Xskip, int ld_xskip,
                               float* in_bgrad, float* workspace, unsigned* const* side_amax, unsigned* x_amax,
                               hipStream_t stream) {
  if (bwd != 0 && bwd != 1) return NDJIR_ERR_ARG;
  return chain_impl(bwd, P, X, ldx, K0, L, Wp, bias, Ks, Ns, side_in, side_out, ld_side, bgrad, Y, ldy, accum_y,
                    has_output, beta, skip_layer, skip_scale, skip_split, Xskip, ld_xskip, nullptr, nullptr, nullptr, nullptr, 0,
                    in_bgrad, workspace, side_amax, x_amax, stream);
}

// Extended form used by the geometric network's double backward (python/renderer.py:52 nn.grad):
// mode 1 with side_add = extra adjoints; mode 2 = tangent chain (side_in2 = s, side_out2 = extra adjoint).
extern "C" int ndjir_mlp_chain_ex(int mode, long long P, const float* X, int ldx, int K0, int L,
                                  const float* const* Wp, const float* const* bias, const int* Ks, const int* Ns,
                                  const float* const* side_in, float* const* side_out, const int* ld_side,
                                  float* const* bgrad, float* Y, int ldy, int accum_y, int has_output, float beta,
                                  int skip_layer, float skip_scale, int skip_split, float* Xskip, int ld_xskip,
                                  const float* const* side_in2, const float* const* side_add, float* const* side_out2,
                                  const float* row_bias, int row_bias_div, float* in_bgrad, float* workspace,
                                  unsigned* const* side_amax, unsigned* x_amax, hipStream_t stream) {
  if (mode < 0 || mode > 2) return NDJIR_ERR_ARG;
  return chain_impl(mode, P, X, ldx, K0, L, Wp, bias, Ks, Ns, side_in, side_out, ld_side, bgrad, Y, ldy, accum_y,
                    has_output, beta, skip_layer, skip_scale, skip_split, Xskip, ld_xskip, side_in2, side_add, side_out2,
                    row_bias, row_bias_div, in_bgrad, workspace, side_amax, x_amax, stream);
}

// Dry run of a chain launch of this shape under the current arithmetic / tile setting: the launchers' own decision code,
// nothing is launched.  bgrad_mask: bit i = layer i produces a bias gradient; in_bgrad: the chain input's column sums too.
static int chain_dry(int mode, long long P, int K0, int L, const int* Ks, const int* Ns, int has_output, int skip_layer,
                     int skip_split, unsigned bgrad_mask, int in_bgrad, ChainDry* out) {
  if (mode < 0 || mode > 2 || L < 1 || L > MAX_CHAIN_LAYERS || !Ks || !Ns || P <= 0) return NDJIR_ERR_ARG;
  static float dummy[4];          // (never dereferenced: the launchers return before any launch)
  const float* wp[MAX_CHAIN_LAYERS];
  const float* side[MAX_CHAIN_LAYERS];
  float* bg[MAX_CHAIN_LAYERS];
  for (int i = 0; i < L; ++i) { wp[i] = dummy; side[i] = dummy; bg[i] = (mode != 0 && ((bgrad_mask >> i) & 1)) ? dummy : nullptr; }
  out->name[0] = 0; out->blocks = 0; out->bg_total = 0;
  g_dry = out;
  const int rc = chain_impl(mode, P, dummy, K0, K0, L, wp, nullptr, Ks, Ns, mode != 0 ? side : nullptr, nullptr, nullptr,
                            mode != 0 ? bg : nullptr, dummy, Ns[L - 1], 0, has_output, 100.f, skip_layer, 1.f,
                            skip_split, (mode == 1 && skip_layer >= 0) ? dummy : nullptr, K0, nullptr, nullptr, nullptr, nullptr, 0,
                            (mode != 0 && in_bgrad) ? dummy : nullptr, dummy, nullptr, nullptr, nullptr);
  g_dry = nullptr;
  return rc;
}

// Diagnostics: the symbol (as rocprofv3 prints it) of the kernel a chain launch of this shape would run.  bench.py keys its
// per-kernel roofline table by it.
extern "C" int ndjir_mlp_chain_kernel(int mode, long long P, int K0, int L, const int* Ks, const int* Ns, int has_output,
                                      int skip_layer, int skip_split, int with_bias_gradients, char* name, int name_bytes) {
  if (!name || name_bytes < 64) return NDJIR_ERR_ARG;
  ChainDry d;
  const int rc = chain_dry(mode, P, K0, L, Ks, Ns, has_output, skip_layer, skip_split, with_bias_gradients ? ~0u : 0u, 0, &d);
  for (int i = 0; i < 64; ++i) name[i] = d.name[i];
  return rc;
}

// Layout of the bias-gradient partial rows a backward / tangent chain launch with the DEFER flag (accum_y bit 2) leaves in
// its workspace: *blocks rows of *row_floats floats -- the layers named by bgrad_mask in order (the output layer of a chain
// with has_output never has one), then the chain input's column sums when in_bgrad.  NDJIR_ERR_UNSUPPORTED: the current
// arithmetic's kernels do not defer.
extern "C" int ndjir_mlp_chain_bias_partials(int mode, long long P, int K0, int L, const int* Ks, const int* Ns, int has_output,
                                             int skip_layer, int skip_split, unsigned bgrad_mask, int in_bgrad, int* blocks,
                                             int* row_floats) {
  if (!blocks || !row_floats) return NDJIR_ERR_ARG;
  ChainDry d;
  const int rc = chain_dry(mode, P, K0, L, Ks, Ns, has_output, skip_layer, skip_split, bgrad_mask, in_bgrad, &d);
  if (rc != NDJIR_OK) return rc;
  if (d.bg_total < 0) return NDJIR_ERR_UNSUPPORTED;
  *blocks = d.blocks;
  *row_floats = d.bg_total;
  return NDJIR_OK;
}

extern "C" long long ndjir_mlp_chain_workspace(int bgrad_total) { return chain_workspace(bgrad_total); }

extern "C" long long ndjir_mlp_wgrad_workspace(int K, int N, long long P) { return wgrad_workspace(K, N, P); }

extern "C" int ndjir_mlp_wgrad(const float* A, int lda, const float* B, int ldb, int K, int N, long long P, float* out,
                               int accum, float* workspace, const unsigned* amax_a, const unsigned* amax_b,
                               hipStream_t stream) {
  if (P <= 0) return NDJIR_OK;
  if (!A || !B || !out || !workspace || lda < K || ldb < N) return NDJIR_ERR_ARG;
  return launch_wgrad(A, lda, B, ldb, K, N, P, out, accum, workspace, g_math, amax_a, amax_b, stream);
}

extern "C" long long ndjir_mlp_wgrad_group_workspace(int n_src, const float* const* A, const int* lda, const long long* P,
                                                     const int* out_id, int n_out, const int* K, const int* N, int target_items,
                                                     const int* layout) {
  if (n_src <= 0) return wgrad_group_workspace(0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, target_items, nullptr);   // (reduce-only calls)
  if (n_out <= 0 || !lda || !P || !out_id || !K || !N) return 0;
  for (int i = 0; i < n_src; ++i)
    if (out_id[i] < 0 || out_id[i] >= n_out) return 0;
  return wgrad_group_workspace(n_src, A, lda, P, out_id, n_out, K, N, target_items, layout);
}

extern "C" int ndjir_mlp_wgrad_group_launches(int n_src, const long long* P, const int* out_id, int n_out) {
  if (n_src <= 0 || n_out <= 0 || !P || !out_id) return 0;
  return wgrad_group_launches(n_src, P, out_id, n_out);
}

extern "C" int ndjir_mlp_wgrad_group(int n_src, const float* const* A, const int* lda, const float* const* B, const int* ldb,
                                     const long long* P, const unsigned* const* amax_a, const unsigned* const* amax_b,
                                     const int* out_id, int n_out, float* const* out, const int* ldo, const int* K, const int* N,
                                     const int* accum, float* workspace, int target_items, int n_extra, float* const* ex_out,
                                     const float* const* ex_partial, const int* ex_n, const int* ex_S, const int* ex_stride,
                                     const int* ex_accum, const int* layout, hipStream_t stream) {
  if ((n_src <= 0 || n_out <= 0) && n_extra <= 0) return NDJIR_OK;
  if (n_src > 0 && (!A || !lda || !B || !ldb || !P || !out_id || !out || !ldo || !K || !N)) return NDJIR_ERR_ARG;
  if (!workspace || (n_extra > 0 && (!ex_out || !ex_partial || !ex_n || !ex_S || !ex_stride))) return NDJIR_ERR_ARG;
  if (g_math != NDJIR_MATH_F16X3) return NDJIR_ERR_UNSUPPORTED;
  for (int o = 0; o < n_out; ++o)
    if (K[o] <= 0 || N[o] <= 0) return NDJIR_ERR_ARG;
  for (int i = 0; i < n_extra; ++i)
    if (!ex_out[i] || !ex_partial[i] || ex_n[i] <= 0 || ex_S[i] < 0 || ex_stride[i] < ex_n[i]) return NDJIR_ERR_ARG;
  for (int i = 0; layout && i < n_src; ++i)
    if ((layout[i] & ~3) != 0 || (layout[i] != 0 && (P[i] & 31) != 0)) return NDJIR_ERR_ARG;
  return launch_wgrad_group(n_src > 0 ? n_src : 0, A, lda, B, ldb, P, amax_a, amax_b, out_id, n_src > 0 ? n_out : 0, out, ldo, K, N, accum,
                            workspace, target_items, n_extra > 0 ? n_extra : 0, ex_out, ex_partial, ex_n, ex_S, ex_stride, ex_accum, layout, stream);
}

extern "C" long long ndjir_mlp_colsum_workspace(int N, long long P) { return colsum_workspace(N, P); }

extern "C" int ndjir_mlp_colsum(const float* X, int ldx, int N, long long P, float* out, int accum, float* workspace,
                                hipStream_t stream) {
  if (N <= 0) return NDJIR_OK;
  if (!out || (P > 0 && (!X || !workspace || ldx < N))) return NDJIR_ERR_ARG;
  if (N > 2048) return NDJIR_ERR_UNSUPPORTED;
  return launch_colsum(X, ldx, N, P, out, accum, workspace, stream);
}

extern "C" int ndjir_mlp_group_colsum(const float* X, int ldx, int N, long long G, int div, float* out, int blocked, hipStream_t stream) {
  if (G <= 0 || N <= 0) return NDJIR_OK;
  if (!X || !out || ldx < N || div < 1) return NDJIR_ERR_ARG;
  if (blocked && ((G * div) & 31) != 0) return NDJIR_ERR_ARG;
  return launch_group_colsum(X, ldx, N, G, div, out, blocked, stream);
}

// Diagnostics: when `buf` (device, MAX_CHAIN_LAYERS * 5 * 8 int64) is non-null, subsequent chain
// launches record shader-clock stamps of workgroup 0 (tools/chain_timeline.py).  Null switches it off.
extern "C" int ndjir_mlp_debug_timeline(long long* buf) {
  g_timeline = buf;
  return NDJIR_OK;
}
