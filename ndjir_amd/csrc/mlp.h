// mlp.h -- argument block of the fused MLP chain kernel (mlp.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ndjir {

constexpr int MAX_CHAIN_LAYERS = 10;

struct ChainLayer {
  const float* Wp;      // packed matrix [Np/32][Kp/8][64][4]
  const float* bias;    // forward only (may be null)
  const float* side_in; // backward / tangent: stored forward activation h of this width (P x N)
  const float* side_in2;// tangent: s (delta of the sdf chain) of this width
  const float* side_add;// backward: extra adjoint added after the softplus' product
  float* side_out2;     // tangent: extra adjoint beta * z * s * exp(-beta h)
  float* side_out;      // forward: activation store (P x N); backward: delta store (P x N)
  float* bgrad;         // backward: bias gradient (N): column sums of this layer's delta (overwritten)
  unsigned* side_amax;  // f16x3 engine: atomicMax of the largest finite |side_out| (bit pattern; caller zeroes), may be null
  int bg_off;           // filled by launch_chain: offset of this layer in the workgroup's LDS accumulator
  int K, N;             // logical dims of this GEMM (input width, output width)
  int Kp, Np;           // padded: Kp % 8 == 0, Np % 32 == 0
  int ld_side;
};

struct ChainDry {
  char name[64];        // kernel symbol as rocprofv3 prints it
  int blocks;           // workgroups of the launch ( = bias-gradient partial rows)
  int bg_total;         // floats per partial row
};

struct ChainArgs {
  long long P;          // points (rows)
  long long n_tiles;    // filled by launch_chain
  const float* X;       // chain input (P x K0), row stride ldx
  float* Y;             // chain output (P x N_last), row stride ldy
  float* Xskip;         // backward: stash of the skip-connection part of the input gradient
  int ldx, ldy, ld_xskip;
  int K0, K0p;
  int L;
  int accum_y;          // Y += result
  int bg_accum;         // bias gradients (bgrad of every layer, in_bgrad) += their sums instead of being overwritten
  int has_output;       // last layer writes Y (else the chain ends with a hidden epilogue)
  int skip_layer;       // -1: none.  fwd: layer whose output is concatenated with X and scaled;
                        // bwd: layer whose output is that concatenated gradient
  int skip_split;       // bwd: first column of the concatenated-input part
  int lds_split;        // filled by launch_chain
  int t64_prio;         // filled by launch_chainw_group (64-point tiles: static priority for one workgroup of a CU's pair)
  int tile_rows;        // 64 (default), 32 or 128 points per workgroup
  int forced_tile;      // tile_rows was forced by the caller (else 64 may be widened to 128 where mlp3w.hip supports the launch)
  float skip_scale;
  float beta;
  const float* row_bias;  // forward: (P / row_bias_div, N_0) term added to the first layer's pre-activation of each row group
  int row_bias_div;
  float* in_bgrad;        // backward: column sums of the chain INPUT (= bias gradient of the net's output layer), K0 floats
  int in_bg_off;          // filled by launch_chain
  float* bg_partial;    // bias gradients: per-workgroup partial sums [grid][bg_total] (workspace)
  int bg_total, bg_lds; // filled by launch_chain: accumulator floats / its offset in LDS
  unsigned* x_amax;     // f16x3 engine: atomicMax of the largest finite |X| (bit pattern; caller zeroes), may be null
  long long* timeline;  // diagnostics (tools/chain_timeline.py): per-layer phase stamps of workgroup 0, else null
  struct ChainDry* dry; // queries (ndjir_mlp_chain_kernel, ndjir_mlp_chain_bias_partials): non-null = report the kernel the
                        // launcher picks, its grid and its bias-gradient partial layout, and return WITHOUT launching
  int side_blocked;     // f16x3 engine: the hidden side tensors (side_in, side_in2, side_add, side_out, side_out2) are POINT-BLOCKED:
                        // element (p, f) of a tensor of row stride ld lives at ((p >> 5) * ld + f) * 32 + (p & 31) -- blocks of 32
                        // points, feature-major inside a block.  A lane of the 128-point-tile kernel then moves its accumulator
                        // registers as they are (register i of 32 lanes = 32 consecutive points of one feature = one 128-byte line;
                        // no quad transposes), and the weight-gradient kernel reads 4 consecutive points of a feature as one
                        // 16-byte load.  Needs P % 32 == 0.  X, Y, Xskip, row_bias stay row-major.
  int defer_bg_reduce;  // backward / tangent: leave the per-workgroup bias-gradient partial rows in `bg_partial` ([grid][bg_total];
                        // layers with a bias gradient in order, then in_bgrad) -- the caller sums them later (one reduction
                        // launch for a whole step: ndjir_mlp_wgrad_group's extra outputs)
  ChainLayer layers[MAX_CHAIN_LAYERS];
};

// Several nets on the same points in ONE launch of the 128-point-tile kernel (mlp3w.hip): a workgroup takes a tile through
// every net of the group in turn (the per-sample material nets of python/renderer.py:113-128 read the same packed row and
// accumulate into the same input gradient).  By value in the kernel arguments: 3 x sizeof(ChainArgs) stays under the 4 KB limit.
constexpr int MAX_GROUP_NETS = 3;
struct ChainGroup {
  int n, pad;
  ChainArgs net[MAX_GROUP_NETS];
};
static_assert(sizeof(ChainGroup) <= 4096, "kernel arguments are limited to 4 KB");

int launch_pack(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream);
int launch_chain(const ChainArgs& a, int mode, hipStream_t stream);
int launch_bgrad_reduce(const float* partial, int S, int total, float* const* ptr, const int* off, int n, int accum, hipStream_t stream);
// bf16 3-way-split engine (mlp6.hip): same ChainArgs, weights packed by launch_pack6
long long packed_size6(int K, int N, int transpose);   // in floats (for allocation through the same API)
int launch_pack6(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream);
int launch_chain6(const ChainArgs& a, int mode, hipStream_t stream);
// f16 2-way-split engine (mlp3.hip): same ChainArgs, weights packed by launch_pack3 (per-column-block scales inside)
long long packed_size3(int K, int N, int transpose);
int launch_pack3(const float* W, int ldw, float* dst, int K, int N, int transpose, hipStream_t stream);
// one entry of the device table of ndjir_mlp_pack_table (layout shared with ndjir_amd/mlp.py: 2 pointers + 8 ints = 48 bytes)
struct PackEntry {
  const float* W;       // (K, N) fp32, row stride ldw
  float* dst;           // packed_size3(K, N, transpose) floats
  int K, N, ldw, transpose;
  int Kp, Np;           // padded dims of the packed matrix (rows of 16, columns of 32)
  int first_block;      // first workgroup of this entry ( = sum of Np / 32 of the entries before it)
  int pad;
};
int launch_pack3_table(const PackEntry* table, int n, int total_blocks, hipStream_t stream);
int launch_chain3(const ChainArgs& a, int mode, hipStream_t stream);
// f16 2-way-split engine on 128-point tiles (mlp3w.hip); NDJIR_ERR_UNSUPPORTED = not a launch for this kernel
int launch_chainw(const ChainArgs& a, int mode, hipStream_t stream);
// n <= MAX_GROUP_NETS nets of one mode on the same number of points; NDJIR_ERR_UNSUPPORTED = launch them one by one
int launch_chainw_group(const ChainArgs* nets, int n, int mode, hipStream_t stream);
// the software-pipelined 128-point-tile kernel (mlp3p.hip): mode mask of the launches it takes (set < 0: query)
int chain_pipeline(int set);
constexpr int CHAIN_MAX_GRID_BG = 512;   // workgroups of a chain launch that produces bias gradients
inline long long chain_workspace(int bg_total) { return (long long)CHAIN_MAX_GRID_BG * ((bg_total + 3) & ~3); }      // (rows padded to 16 bytes)
long long wgrad_workspace(int K, int N, long long P);
// math: 0 fp32 MFMA, 1 bf16x6, 2 f16x3 (amax_a / amax_b: recorded maxima of A / B, null = computed by a pre-pass)
int launch_wgrad(const float* A, int lda, const float* B, int ldb, int K, int N, long long P, float* out, int accum,
                 float* workspace, int math, const unsigned* amax_a, const unsigned* amax_b, hipStream_t stream);

// many weight gradients in one launch + one split-reduction launch (f16x3 arithmetic; wgrad.hip "grouped weight gradients")
long long wgrad_group_workspace(int n_src, const float* const* A, const int* lda, const long long* P, const int* out_id, int n_out,
                                const int* K, const int* N, int target_items, const int* layout);
int wgrad_group_launches(int n_src, const long long* P, const int* out_id, int n_out);
// layout[i] (null = all row-major): bit 0 = A[i] is point-blocked (ChainArgs::side_blocked; lda = its row stride), bit 1 = B[i] is;
// a blocked operand needs P[i] % 32 == 0.
// extras: n_extra reduce-only outputs ex_out[i] (ex_n[i] floats) (+)= the sum of ex_S[i] partial rows ex_partial[i] + s * ex_stride[i]
// (the deferred bias gradients of chain launches: ChainArgs::defer_bg_reduce), summed by the same reduction launch
int launch_wgrad_group(int n_src, const float* const* A, const int* lda, const float* const* B, const int* ldb, const long long* P,
                       const unsigned* const* amax_a, const unsigned* const* amax_b, const int* out_id, int n_out,
                       float* const* out, const int* ldo, const int* K, const int* N, const int* accum, float* workspace,
                       int target_items, int n_extra, float* const* ex_out, const float* const* ex_partial, const int* ex_n,
                       const int* ex_S, const int* ex_stride, const int* ex_accum, const int* layout, hipStream_t stream);

long long colsum_workspace(int N, long long P);
int launch_group_colsum(const float* X, int ldx, int N, long long G, int div, float* out, int blocked, hipStream_t stream);
int launch_colsum(const float* X, int ldx, int N, long long P, float* out, int accum, float* workspace,
                  hipStream_t stream);

}  // namespace ndjir
