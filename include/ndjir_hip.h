/*
 * ndjir_hip.h -- C ABI of libndjir_hip.so, the MI355X (gfx950) drop-in for the native
 * extension modules of sony/NDJIR (reference: /root/reference/csrc, 19 pybind11 modules built
 * one-per-.cu by Makefile:16-23 and called from python/grid_feature/ *.py,
 * python/intersection/ *.py, python/sampler.py).
 *
 * Conventions (identical to the reference's pybind11 functions unless stated):
 *   - symbol name = ndjir_<reference module without "_cuda">_<reference function>
 *   - same argument order; raw DEVICE pointers to contiguous row-major fp32 (the reference passes
 *     them as int64); std::vector<float/int> min/max/grid_sizes become `const float[3]` /
 *     `const int[3]` HOST pointers; bool becomes int
 *   - N is the reference's thread count (P*D, P*D*3 or L*P), kept for signature parity
 *   - one trailing `hipStream_t stream` (the reference always uses the default stream)
 *   - returns 0 on success, NDJIR_ERR_* otherwise (the reference returns void and only printf's
 *     launch errors, csrc/cuda_common.cuh:24-32)
 *   - caller owns every buffer; kernels never allocate.  `accum == 0` => destination is
 *     zero-filled first, for exactly the entry points where the reference does so
 *     (grad_query, grad_feature); grad_query_grad_query, grad_query_grad_feature,
 *     grad_feature_grad_query and every TV backward ALWAYS accumulate, as in the reference
 *   - `boundary_check` is accepted and ignored, as in every reference kernel
 *   - launches are asynchronous on `stream`; every scratch buffer is the caller's (`workspace` arguments), so launches on
 *     different streams share nothing.  The library keeps no state beyond two PROCESS-WIDE TUNING KNOBS of the MLP engine --
 *     the matrix arithmetic (ndjir_mlp_set_math) and the forced tile height (ndjir_mlp_set_tile_rows), both also readable
 *     from the environment at first use -- which select between kernels that compute the same values and must not be
 *     changed while launches that read them are being issued from another thread (packed weights are specific to the
 *     arithmetic they were packed under), and the diagnostic hooks (ndjir_mlp_debug_timeline, ndjir_mlp_chain_kernel)
 *
 * No torch types appear here; the library links only libamdhip64.
 */
#ifndef NDJIR_HIP_H
#define NDJIR_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDJIR_OK 0
#define NDJIR_ERR_LAUNCH 1
#define NDJIR_ERR_UNSUPPORTED 2
#define NDJIR_ERR_ARG 3

const char* ndjir_version(void);
int ndjir_zero(float* p, long long n, hipStream_t stream);

/* ---- hash-grid host helpers: csrc/grid_feature/common_voxel_hash.cuh:24-55 -------------------- */
int ndjir_hash_force_align(int size, int mod);                      /* size + size % mod (sic) */
int ndjir_hash_grid_size(int G0, float growth_factor, int level);
int ndjir_hash_table_size(int G, int T0);
long long ndjir_hash_num_params(int G0, float growth_factor, int T0, int L, int D);

/* ---- dense voxel grids (G0,G1,G2,D) -----------------------------------------------------------
 * voxel_feature          <- csrc/grid_feature/voxel_feature_cuda.cu:101-115,207-228,289-311,414-438,522-546,616-638,709-731,816-838 (exports :844-863)
 * cosine_voxel_feature   <- csrc/grid_feature/cosine_voxel_feature_cuda.cu (same export list)
 * lanczos_voxel_feature  <- csrc/grid_feature/lanczos_voxel_feature_cuda.cu (same export list)
 * Python callers: python/grid_feature/{,cosine_,lanczos_}voxel_feature.py:85,119,213,242,257      */
#define NDJIR_DECL_VOXEL_FAMILY(P)                                                                                          \
  int ndjir_##P##_query_on_voxel(int N, float* output, const float* query, const float* feature, const int* grid_sizes,     \
                                 int D, const float* min, const float* max, int boundary_check, hipStream_t stream);       \
  int ndjir_##P##_grad_query(int N, float* grad_query, const float* grad_output, const float* query, const float* feature,  \
                             const int* grid_sizes, int D, const float* min, const float* max, int boundary_check,         \
                             int accum, hipStream_t stream);                                                                \
  int ndjir_##P##_grad_feature(int N, float* grad_feature, const float* grad_output, const float* query,                    \
                               const int* grid_sizes, int D, const float* min, const float* max, int boundary_check,       \
                               int accum, hipStream_t stream);                                                              \
  int ndjir_##P##_grad_query_grad_grad_output(int N, float* grad_grad_output, const float* grad_grad_query,                 \
                                              const float* query, const float* feature, const int* grid_sizes, int D,      \
                                              const float* min, const float* max, int boundary_check, int accum,           \
                                              hipStream_t stream);                                                          \
  int ndjir_##P##_grad_query_grad_feature(int N, float* grad_feature, const float* grad_grad_query,                         \
                                          const float* grad_output, const float* query, const int* grid_sizes, int D,      \
                                          const float* min, const float* max, int boundary_check, int accum,               \
                                          hipStream_t stream);
NDJIR_DECL_VOXEL_FAMILY(voxel_feature)
/* No reference counterpart (nnabla zero-fills gradients densely): zero only the cells that the N query
 * points (N x 3) touch in an accumulate-in-place gradient buffer of the linear dense voxel grid -- the
 * 8 corners, which also cover the TV backward's cells -- instead of rewriting the whole buffer. */
int ndjir_voxel_feature_zero_touched(int N, float* grad_feature, const float* query, const int* grid_sizes, int D,
                                     const float* min, const float* max, hipStream_t stream);
/* ... for any interpolation of the dense voxel family: interp 0 linear (8 corners), 1 cosine, 2 Lanczos (4 x 4 x 4 taps) */
int ndjir_voxel_feature_zero_touched_interp(int N, float* grad_feature, const float* query, const int* grid_sizes, int D,
                                            const float* min, const float* max, int interp, hipStream_t stream);
/* *flag |= 1 (device int) when one of those cells holds an inf or nan: the grid half of
 * `check_inf_or_nan_grad` (python/solver.py:67-69) without reading the 2 GiB buffer. */
int ndjir_voxel_feature_check_touched(int N, const float* grad_feature, const float* query, const int* grid_sizes, int D,
                                      const float* min, const float* max, int* flag, hipStream_t stream);
/* ---- sparse exchange of a dense grid gradient between ranks (no reference counterpart: the reference is single-GPU).
 * A rank touches < 1 % of the 512^3 cells per step.  pack_rows appends the NON-ZERO rows (the D = 4 or 8 floats of one
 * cell) of the cells in the interpolation stencils of the N query points to a packed list -- ids (capacity) int32, rows
 * (capacity, D) -- each cell once (bitmap: 1 bit per cell, all zero before the first call, cleared again by
 * ndjir_sparse_rows_clear_bitmap); *count (device int) keeps counting past capacity.  After all-gathers of the counts
 * and of the first `limit` rows of every rank -- ids (world, limit), rows (world, limit, D) packed with row stride
 * `capacity` (ndjir_sparse_rows_apply; pass capacity = limit for the layout one all_gather_into_tensor leaves behind) or
 * `*limit` (ndjir_sparse_rows_zero), counts (world), all device memory -- ndjir_sparse_rows_apply adds every other rank's
 * rows into the local
 * buffer, ndjir_sparse_rows_overflow raises a device flag when some rank listed more than `limit` rows (the caller
 * vetoes that optimizer step and grows `limit`; no host synchronisation per step), and ndjir_sparse_rows_zero clears
 * all listed rows (own_ids / own_count non-null: this rank's own rows from its local list, which may exceed `limit`),
 * re-arming the accumulate-in-place buffer for the next step without touching the whole tensor. */
int ndjir_voxel_feature_pack_rows(int N, const float* grad_feature, const float* query, const int* grid_sizes, int D,
                                  const float* min, const float* max, unsigned* bitmap, int* ids, float* rows, int* count,
                                  int capacity, hipStream_t stream);
/* topo 0 dense voxel (grid_sizes[3]) / 1 tri-plane / 2 tri-line (grid_sizes[0] = G); interp 0 linear / 1 cosine / 2 Lanczos */
int ndjir_grid_pack_rows(int topo, int interp, int N, const float* grad_feature, const float* query, const int* grid_sizes, int D,
                         const float* min, const float* max, unsigned* bitmap, int* ids, float* rows, int* count, int capacity,
                         hipStream_t stream);
int ndjir_sparse_rows_clear_bitmap(const int* ids, const int* count, int capacity, unsigned* bitmap, hipStream_t stream);
int ndjir_sparse_rows_apply(const int* ids, const float* rows, const int* counts, int world, int capacity, int limit,
                            int skip_rank, float* grad_feature, int D, hipStream_t stream);
/* stats (device, 2 ints, may be null): [0] running maximum of the counts over all exchanges (the host sizes `limit` from it
 * at its next look, whichever exchange overflowed), [1] number of exchanges that overflowed */
int ndjir_sparse_rows_overflow(const int* counts, int world, int limit, int* flag, int* stats, int counts_stride, hipStream_t stream);
/* Packed wire: with counts == NULL, ndjir_sparse_rows_apply / _zero read every rank's count from a header of
 * ndjir_sparse_rows_header() = 4 ints in front of its ids -- [count, list capacity the rank needs, 0, 0] --, the id lists being
 * (world, limit + 4) ints (rows unchanged): the counts then travel with the ids instead of in a collective of their own
 * (ndjir_sparse_rows_overflow reads them with counts_stride = limit + 4; dense counts: counts_stride = 1). */
int ndjir_sparse_rows_header(void);
/* (`limit`: device int here -- the call may be replayed from a captured HIP graph after the limit has grown; it is also the
 * row stride of the communicated lists; `capacity`: that of this rank's own list) */
int ndjir_sparse_rows_zero(const int* ids, const int* counts, int world, int capacity, const int* limit, int own_rank,
                           const int* own_ids, const int* own_count, float* grad_feature, int D, hipStream_t stream);
/* dense fall-back of the re-arm: clears all `n` floats of the buffer iff *own_count > capacity, i.e. iff ndjir_grid_pack_rows had
 * to drop cells its list had no room for (those hold gradient no list names; the step that dropped them is vetoed through
 * ndjir_sparse_rows_overflow).  One int read otherwise. */
int ndjir_sparse_rows_zero_if_dropped(const int* own_count, int capacity, float* grad_feature, long long n, hipStream_t stream);
/* round 6: the dense-voxel query of python/network.py:120-151 fused with the positional encoding of :96-117 -- rows
 * e[p] = [x, cos(x_d 2^k), sin(x_d 2^k), feature(p)] (row stride lde >= 3 + 6 M + D) of N points in one launch; interp 0 linear /
 * 1 cosine / 2 Lanczos; values bit-identical to <family>_query followed by ndjir_geo_encode */
int ndjir_voxel_feature_query_encode(int N, int M, const float* query, const float* feature, const int* grid_sizes, int D,
                                     const float* min, const float* max, int interp, float* e, int lde, hipStream_t stream);
/* ... the same for the tri-plane + tri-line pair of `triplaneline` grids (python/network.py:137-147): rows [x | cos | sin | tri-plane
 * feature (Dp, 3) | tri-line feature (Dl, 3)], bit-identical to the two <family>_query_on_* launches followed by ndjir_geo_encode */
int ndjir_triplaneline_query_encode(int N, int M, const float* query, const float* plane_feature, int Gp, int Dp,
                                    const float* line_feature, int Gl, int Dl, const float* min, const float* max, int interp,
                                    float* e, int lde, hipStream_t stream);
NDJIR_DECL_VOXEL_FAMILY(cosine_voxel_feature)
NDJIR_DECL_VOXEL_FAMILY(lanczos_voxel_feature)

/* linear dense voxel only: voxel_feature_cuda.cu:522-546 (1-2), :709-731 (2-1), :816-838 (2-2) */
int ndjir_voxel_feature_grad_query_grad_query(int N, float* grad_query, const float* grad_grad_query,
                                              const float* grad_output, const float* query, const float* feature,
                                              const int* grid_sizes, int D, const float* min, const float* max,
                                              int boundary_check, int accum, hipStream_t stream);
int ndjir_voxel_feature_grad_feature_grad_grad_output(int N, float* grad_grad_output, const float* grad_grad_feature,
                                                      const float* query, const int* grid_sizes, int D, const float* min,
                                                      const float* max, int boundary_check, int accum, hipStream_t stream);
int ndjir_voxel_feature_grad_feature_grad_query(int N, float* grad_query, const float* grad_grad_feature,
                                                const float* grad_output, const float* query, const int* grid_sizes, int D,
                                                const float* min, const float* max, int boundary_check, int accum,
                                                hipStream_t stream);

/* ---- tri-plane (3,G,G,D) and tri-line (3,G,D); output (P, D, 3) plane-fastest ----------------------
 * triplane_feature <- csrc/grid_feature/triplane_feature_cuda.cu:91-112,179-200,257-279,363-385,558-580 (exports :783-805)
 * triline_feature  <- csrc/grid_feature/triline_feature_cuda.cu (same list); cosine_ / lanczos_ variants alike.
 * Python callers: python/grid_feature/{,cosine_,lanczos_}tri{plane,line}_feature.py:85,119,213,240,256 */
#define NDJIR_DECL_PLANE_FAMILY(P, FWD)                                                                                      \
  int ndjir_##P##_##FWD(int N, float* output, const float* query, const float* feature, int G, int D, const float* min,      \
                        const float* max, int boundary_check, hipStream_t stream);                                          \
  int ndjir_##P##_grad_query(int N, float* grad_query, const float* grad_output, const float* query, const float* feature,  \
                             int G, int D, const float* min, const float* max, int boundary_check, int accum,              \
                             hipStream_t stream);                                                                           \
  int ndjir_##P##_grad_feature(int N, float* grad_feature, const float* grad_output, const float* query, int G, int D,      \
                               const float* min, const float* max, int boundary_check, int accum, hipStream_t stream);     \
  int ndjir_##P##_grad_query_grad_grad_output(int N, float* grad_grad_output, const float* grad_grad_query,                 \
                                              const float* query, const float* feature, int G, int D, const float* min,    \
                                              const float* max, int boundary_check, int accum, hipStream_t stream);        \
  int ndjir_##P##_grad_query_grad_feature(int N, float* grad_feature, const float* grad_grad_query,                         \
                                          const float* grad_output, const float* query, int G, int D, const float* min,    \
                                          const float* max, int boundary_check, int accum, hipStream_t stream);
NDJIR_DECL_PLANE_FAMILY(triplane_feature, query_on_triplane)
NDJIR_DECL_PLANE_FAMILY(cosine_triplane_feature, query_on_triplane)
NDJIR_DECL_PLANE_FAMILY(lanczos_triplane_feature, query_on_triplane)
NDJIR_DECL_PLANE_FAMILY(triline_feature, query_on_triline)
NDJIR_DECL_PLANE_FAMILY(cosine_triline_feature, query_on_triline)
NDJIR_DECL_PLANE_FAMILY(lanczos_triline_feature, query_on_triline)

/* ---- multi-resolution hash grid; N = L*P; feature outputs / grad_output in layout (D, L, P) --------
 * voxel_hash_feature         <- csrc/grid_feature/voxel_hash_feature_cuda.cu:102-119,197-217,312-332,406-427,540-563,751-773 (exports :976-1000)
 * lanczos_voxel_hash_feature <- csrc/grid_feature/lanczos_voxel_hash_feature_cuda.cu (same list)
 * Python callers: python/grid_feature/{,lanczos_}voxel_hash_feature.py:145,192,301,338,357           */
#define NDJIR_DECL_HASH_FAMILY(P)                                                                                            \
  int ndjir_##P##_hash_index(int N, float* output, const float* query, int G, int T, const float* min, const float* max,     \
                             int boundary_check, hipStream_t stream);                                                       \
  int ndjir_##P##_voxel_hash_feature(int N, float* output, const float* query, const float* feature, int G0,                \
                                     float growth_factor, int T0, int L, int D, const float* min, const float* max,        \
                                     int boundary_check, hipStream_t stream);                                               \
  int ndjir_##P##_grad_query(int N, float* grad_query, const float* grad_output, const float* query, const float* feature,  \
                             int G0, float growth_factor, int T0, int L, int D, const float* min, const float* max,        \
                             int boundary_check, int accum, hipStream_t stream);                                            \
  int ndjir_##P##_grad_feature(int N, float* grad_feature, const float* grad_output, const float* query, int G0,            \
                               float growth_factor, int T0, int L, int D, const float* min, const float* max,              \
                               int boundary_check, int accum, hipStream_t stream);                                          \
  int ndjir_##P##_grad_query_grad_grad_output(int N, float* grad_grad_output, const float* grad_grad_query,                 \
                                              const float* query, const float* feature, int G0, float growth_factor,       \
                                              int T0, int L, int D, const float* min, const float* max,                    \
                                              int boundary_check, int accum, hipStream_t stream);                           \
  int ndjir_##P##_grad_query_grad_feature(int N, float* grad_feature, const float* grad_grad_query,                         \
                                          const float* grad_output, const float* query, int G0, float growth_factor,       \
                                          int T0, int L, int D, const float* min, const float* max, int boundary_check,    \
                                          int accum, hipStream_t stream);
NDJIR_DECL_HASH_FAMILY(voxel_hash_feature)
NDJIR_DECL_HASH_FAMILY(lanczos_voxel_hash_feature)

/* ---- sampled total-variation loss ----------------------------------------------------------------
 * csrc/grid_feature/total_variation_loss_cuda.cu:86-105,177-199; ..._on_triplane_cuda.cu; ..._on_triline_cuda.cu;
 * ..._on_voxel_hash_cuda.cu.  Python callers: python/grid_feature/total_variation_loss*.py:75,100 */
int ndjir_total_variation_loss_tv_loss_on_voxel(int N, float* output, const float* query, const float* feature,
                                                const int* grid_sizes, int D, const float* min, const float* max,
                                                int boundary_check, hipStream_t stream);
int ndjir_total_variation_loss_tv_loss_on_voxel_backward(int N, float* grad_feature, const float* grad_output,
                                                         const float* query, const float* feature, const int* grid_sizes,
                                                         int D, const float* min, const float* max, int sym_backward,
                                                         int boundary_check, int accum, hipStream_t stream);
int ndjir_total_variation_loss_on_triplane_tv_loss_on_triplane(int N, float* output, const float* query, const float* feature,
                                                               int G, int D, const float* min, const float* max,
                                                               int boundary_check, hipStream_t stream);
int ndjir_total_variation_loss_on_triplane_tv_loss_on_triplane_backward(int N, float* grad_feature, const float* grad_output,
                                                                        const float* query, const float* feature, int G, int D,
                                                                        const float* min, const float* max, int sym_backward,
                                                                        int boundary_check, int accum, hipStream_t stream);
int ndjir_total_variation_loss_on_triline_tv_loss_on_triline(int N, float* output, const float* query, const float* feature,
                                                             int G, int D, const float* min, const float* max,
                                                             int boundary_check, hipStream_t stream);
int ndjir_total_variation_loss_on_triline_tv_loss_on_triline_backward(int N, float* grad_feature, const float* grad_output,
                                                                      const float* query, const float* feature, int G, int D,
                                                                      const float* min, const float* max, int sym_backward,
                                                                      int boundary_check, int accum, hipStream_t stream);
int ndjir_total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash(int N, float* output, const float* query,
                                                                   const float* feature, int G0, float growth_factor, int T0,
                                                                   int L, int D, const float* min, const float* max,
                                                                   int boundary_check, hipStream_t stream);
int ndjir_total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash_backward(int N, float* grad_feature,
                                                                            const float* grad_output, const float* query,
                                                                            const float* feature, int G0, float growth_factor,
                                                                            int T0, int L, int D, const float* min,
                                                                            const float* max, int sym_backward,
                                                                            int boundary_check, int accum, hipStream_t stream);

/* ---- intersection: csrc/intersection/ray_aabb_intersection_cuda.cu:145-162, ray_sphere_intersection_cuda.cu:81-96;
 * callers python/intersection/ray_aabb_intersection.py:82-101, ray_sphere_intersection.py:80-97.
 * n_hits is stored as float, as in the reference. */
int ndjir_ray_aabb_intersection(int N, float* t_near, float* t_far, float* n_hits, const float* camloc,
                                const float* raydir, int B, int R, const float* min, const float* max, hipStream_t stream);
int ndjir_ray_sphere_intersection(int N, float* t_near, float* t_far, float* n_hits, const float* camloc,
                                  const float* raydir, int B, int R, float radius, hipStream_t stream);

/* ---- light-direction sampling: csrc/sampling/inverse_transform_cuda.cu:72-90,139-160;
 * caller python/sampler.py:376-389 */
int ndjir_inverse_transform_sample_uniform_directions(int size, float* light_dirs, const float* normal,
                                                      const float* cdf_the, const float* cdf_phi, int batch_size,
                                                      int n_lights, int n_thes, int n_phis, float eps, hipStream_t stream);
int ndjir_inverse_transform_sample_importance_directions(int size, float* light_dirs, const float* normal,
                                                         const float* cdf_the, const float* cdf_phi, const float* alpha,
                                                         int batch_size, int n_lights, int n_thes, int n_phis, float eps,
                                                         hipStream_t stream);

/* Diagnostics (no reference counterpart): output[i] = ndjir_expf(input[i]) (sigmoid != 0: ndjir_sigmoidf) -- the shared
 * definitions of include/ndjir_math.h on which the sampler's bin decisions are built, evaluated on the device so that a
 * test can compare their bits with the host's and their values with float64 exp. */
int ndjir_math_expf(int size, float* output, const float* input, int sigmoid, hipStream_t stream);
/* ---- csrc/activation/squareplus_cuda.cu:62-93 (built by the reference, unused by its model) */
int ndjir_squareplus_forward(int size, float* output, const float* input, float b, hipStream_t stream);
int ndjir_squareplus_backward(int size, float* dinput, const float* doutput, const float* input, float b, int accum,
                              hipStream_t stream);

/* ---- hierarchical sampler: one up-sampling round --------------------------------------------------
 * Replaces the ~40 nnabla launches of one iteration of SamplePoints.sample_importance_dists
 * (python/sampler.py:194-240) given the SDF at the current samples: robust slope, sigmoid CDF,
 * alpha, transmittance weights, normalisation, inverse-transform sampling at the deterministic
 * u_m = m / (M - 1 + 1/M), clip to [t_near, t_far], merge-sort.  t, sdf: (R, N); t_near, t_far: (R);
 * t_out: (R, N + M) sorted; idx_out: (R, M) int32 bin indices.  N + M <= 256, M <= 32.
 * Optional (may be null): src_out (R, N + M) int32 = for every merged position the slot it came
 * from (i < N: old sample i; N + m: new sample m) and tnew_out (R, M) the new distances, so that a
 * caller can evaluate the SDF at the M new samples only and merge it into the values it already has.
 * exp and the scan / reduction orders are DEFINED in include/ndjir_math.h, so the integer indices
 * are reproducible bit for bit on the host (oracle) and the device. */
int ndjir_sampler_importance_round(int R, int N, int M, float gain, const float* t, const float* sdf,
                                   const float* t_near, const float* t_far, float* t_out, int* idx_out,
                                   int* src_out, float* tnew_out, hipStream_t stream);
/* The rest of SamplePoints around that round as kernels instead of ~60 small launches (same arithmetic, expression by
 * expression, as the stock-op formulation; floating-point contraction off):
 * ndjir_sampler_begin (python/sampler.py:71-165, 265-273): mask = n_hits > 1 (n_hits null: 1), stratified distances
 *   t_i = t_near + (t_far - t_near) / N0 (i + u_i) and their points x_i = c + t_i d.  camloc (B,3), rays_per_batch = R / B.
 * ndjir_sampler_round_fused: ndjir_sampler_importance_round that also (a) gathers the SDF of the current samples from the
 *   previous round's values sdf_prev (R, n_prev) and the values at the samples that round added, sdf_new (R, N - n_prev), by
 *   its source map src_prev (R, N) -- sdf_new null: sdf_prev is the (R, N) array -- writing the result to sdf_out (may be
 *   null), and (b) emits the new samples' points xnew_out (R, M, 3) = camloc + t_new raydir (null: skipped).
 * ndjir_sampler_finish (python/sampler.py:275-299, 244-254): x_fg = c + t d, t_fg = [t, t_far]; background (x_bg null:
 *   skipped): t_base = t_far mask + (|c| - radius)(1 - mask), t_bg = sort(t_base / background_sample) (Nb + 1 <= 64 values),
 *   x_bg (R, Nb, 4) = inverted-sphere coordinates (x / |x|, 1 / |x|) of the first Nb. */
int ndjir_sampler_begin(long long R, int N0, int rays_per_batch, const float* camloc, const float* raydir, const float* t_near,
                        const float* t_far, const float* n_hits, const float* stratified_sample, float* mask, float* t0,
                        float* x0, hipStream_t stream);
int ndjir_sampler_round_fused(int R, int N, int M, float gain, const float* t, const float* sdf_prev, int n_prev,
                              const float* sdf_new, const int* src_prev, float* sdf_out, const float* t_near, const float* t_far,
                              const float* camloc, const float* raydir, int rays_per_batch, float* t_out, int* idx_out,
                              int* src_out, float* tnew_out, float* xnew_out, hipStream_t stream);
int ndjir_sampler_finish(long long R, int N, int Nb, int rays_per_batch, float radius, const float* camloc, const float* raydir,
                         const float* t, const float* t_far, const float* mask, const float* background_sample, float* x_fg,
                         float* t_fg, float* x_bg, float* t_bg, hipStream_t stream);

/* ---- volume-rendering stage (ndjir_amd/csrc/render.hip) ----------------------------------------------
 * python/renderer.py:55-67 (foreground alpha from sdf, n = d sdf/dx and the ray direction, with the
 * cos-anneal blend), :79-87 (alpha = [alpha_fg * mask, alpha_bg]; transmittance = exclusive cumprod of
 * 1 - alpha; weights = alpha * transmittance) as ONE launch, and its backward as one launch
 * (reverse scan without division; the reference relies on nnabla's per-function backward).
 * Rays r < R; sdf (R,N); n (R,N,3); raydir (R,3); t (R,N+1); gain, cos_anneal_ratio: 1-element device
 * arrays; mask (R); alpha_bg (R,Nb); alpha_fg (R,N); trans, weights (R,N+Nb).  N + Nb <= 256.
 * Backward: g_alpha_fg (R,N) / g_trans / g_weights (R,N+Nb) may be null (= 0); writes g_sdf (R,N),
 * g_n (R,N,3), g_gain_ray (R) (per-ray partials of dL/dgain; sum them) and, if non-null, g_alpha_bg. */
int ndjir_render_alpha_weights(int R, int N, int Nb, const float* sdf, const float* n, const float* raydir,
                               const float* t, const float* gain, const float* cos_anneal_ratio, const float* mask,
                               const float* alpha_bg, float* alpha_fg, float* trans, float* weights, hipStream_t stream);
int ndjir_render_alpha_weights_backward(int R, int N, int Nb, const float* sdf, const float* n, const float* raydir,
                                        const float* t, const float* gain, const float* cos_anneal_ratio,
                                        const float* mask, const float* alpha_bg, const float* trans,
                                        const float* g_alpha_fg, const float* g_trans, const float* g_weights,
                                        float* g_sdf, float* g_n, float* g_gain_ray, float* g_alpha_bg,
                                        hipStream_t stream);
/* VR integral (renderer.py:84-87): out (R,C) = sum_i w[r][i] x[r][i][c], w rows of stride ldw >= S,
 * x (R*S rows of C values, row stride ldx >= C).  Backward: gx (R,S,C) = w g, gw[r][i] (row stride ldgw) =
 * sum_c x g; either may be null. */
int ndjir_render_integrate(int R, int S, int C, const float* w, int ldw, const float* x, int ldx, float* out,
                           hipStream_t stream);
int ndjir_render_integrate_backward(int R, int S, int C, const float* w, int ldw, const float* x, int ldx, const float* g,
                                    float* gx, float* gw, int ldgw, hipStream_t stream);

/* Direct-light integrals over the M sampled light directions of each ray, one launch each way.
 * diffuse (python/renderer.py:117-118):   out (R,C) = mean_m soft_vis * env * clamp(n.l, eps_dot)
 * specular (python/renderer.py:136-161 with python/specular_brdf.py:40-118: filament model, importance
 * sampling, no split sum):                  out (R,3) = weight * mean_m sBRDF * soft_vis * env * clamp(n.l, eps_dot)
 * normal, view_dir (R,3) unit; light_dir (R,M,3); roughness (R); specular_color (R,3); soft_vis (R,M);
 * env (R,M,C), C in {1..3} (specular: 1 or 3).  Light and view directions carry no gradient
 * (python/sampler.py:391-392).  Backward writes g_normal (R,3), g_roughness (R), g_specular_color (R,3),
 * g_soft_vis (R,M), g_env (R,M,C). */
int ndjir_render_diffuse_light(int R, int M, int C, const float* normal, const float* light_dir, const float* soft_vis,
                               const float* env, float eps_dot, float* out, hipStream_t stream);
int ndjir_render_diffuse_light_backward(int R, int M, int C, const float* normal, const float* light_dir, const float* soft_vis,
                                        const float* env, float eps_dot, const float* g, float* g_normal, float* g_soft_vis,
                                        float* g_env, hipStream_t stream);
int ndjir_render_specular_light_filament(int R, int M, int C, const float* normal, const float* view_dir, const float* light_dir,
                                         const float* roughness, const float* specular_color, const float* soft_vis,
                                         const float* env, float eps_dot, float weight, float* out, hipStream_t stream);
int ndjir_render_specular_light_filament_backward(int R, int M, int C, const float* normal, const float* view_dir,
                                                  const float* light_dir, const float* roughness, const float* specular_color,
                                                  const float* soft_vis, const float* env, float eps_dot, float weight,
                                                  const float* g, float* g_normal, float* g_roughness, float* g_specular_color,
                                                  float* g_soft_vis, float* g_env, hipStream_t stream);
/* Every other branch of the specular BRDF in one launch each way (python/specular_brdf.py:40-199, python/renderer.py:141-161):
 * model 0 filament | 1 ue4; sampling 0 importance | 1 uniform; split != 0: use_split_sum --
 * weight * mean_m(soft_vis env) * mean_m(sBRDF cos) instead of weight * mean_m(sBRDF soft_vis env cos).  Shapes as for
 * ndjir_render_specular_light_filament (which stays the default branch's kernel).  The half vector has no gradient. */
int ndjir_render_specular_light(int R, int M, int C, int model, int sampling, int split, const float* normal, const float* view_dir,
                                const float* light_dir, const float* roughness, const float* specular_color, const float* soft_vis,
                                const float* env, float eps_dot, float weight, float* out, hipStream_t stream);
int ndjir_render_specular_light_backward(int R, int M, int C, int model, int sampling, int split, const float* normal,
                                         const float* view_dir, const float* light_dir, const float* roughness,
                                         const float* specular_color, const float* soft_vis, const float* env, float eps_dot,
                                         float weight, const float* g, float* g_normal, float* g_roughness, float* g_specular_color,
                                         float* g_soft_vis, float* g_env, hipStream_t stream);

/* Background head (python/network.py:543-556): h (P, 1 + F) = output of the background model's geometric net, x (P, nx) its
 * inverted-sphere sample coordinates, delta (P) the sample spacing.  alpha (P) = 1 - exp(-softplus_100(h_0) delta);
 * inp (P, nx + F) = [x | h_1..F], the per-sample input of the background lighting net (its per-ray inputs enter the fused
 * chain as a row term).  Backward: g_h (P, 1 + F) from g_alpha (P) and g_inp (P, nx + F) (either may be null = zero). */
int ndjir_render_background_head(long long P, int nx, int F, const float* h, const float* x, const float* delta, float* alpha,
                                 float* inp, hipStream_t stream);
int ndjir_render_background_head_backward(long long P, int nx, int F, const float* h, const float* delta, const float* g_alpha,
                                          const float* g_inp, float* g_h, hipStream_t stream);

/* The geometric network's SDF-to-density gain  clamp(exp(scale p), lo, hi)  of its scalar parameter(s) p (n values):
 * python/network.py:229-231 (`F.clip_by_value(F.exp(gain * 10), 1e-6, 5e4)`), one launch each way; the backward passes the
 * gradient where lo <= exp(scale p) <= hi. */
int ndjir_render_gain(int n, const float* p, float scale, float lo, float hi, float* out, hipStream_t stream);
int ndjir_render_gain_backward(int n, const float* p, float scale, float lo, float hi, const float* g, float* gp,
                               hipStream_t stream);

/* Both light integrals AND the pixel composition of a ray in one launch each way (the default branch of
 * python/renderer.py:105-178: filament BRDF, importance sampling, no split sum).  The environment-light and soft-visibility
 * nets run once over the 2 M directions [diffuse (M) | specular (M)] of a ray and hand over their RAW outputs: light_dirs
 * (R,2M,3), raw_soft_vis (R,2M), raw_env (R,2M,C), C = 1 | 3.  acts[2] = output activation of (soft_vis, env): 0 identity,
 * 1 softplus(beta) (python/network.py:288-296 `act_last` with `inverse_black_degree`), 2 sigmoid, 3 relu; params[5] =
 * (beta_soft_vis, beta_env, env upper bound (> 0: clamp(., 0, ub), python/network.py:295-296), eps_dot, specular weight).
 * pix (R,9) = VR of the material head's V [implicit, roughness, specular x3, photo, base x3]; bg (R,3) or null.
 * Writes color (R,3) (= ndjir_render_pixel_compose of the two integrals) and keeps env_pixel (R,C), spec_pixel (R,3) for the
 * backward, which writes g_normal (R,3), the raw outputs' gradients over all 2 M directions, the full g_pix row (R,9)
 * and g_bg (R,3; may be null). */
int ndjir_render_direct_light(int R, int M, int C, const int* acts, const float* params, int entangle, const float* normal,
                              const float* view_dir, const float* light_dirs, const float* raw_soft_vis, const float* raw_env,
                              const float* pix, const float* bg, float* color, float* env_pixel, float* spec_pixel,
                              hipStream_t stream);
int ndjir_render_direct_light_backward(int R, int M, int C, const int* acts, const float* params, int entangle, const float* normal,
                                       const float* view_dir, const float* light_dirs, const float* raw_soft_vis,
                                       const float* raw_env, const float* pix, const float* env_pixel, const float* spec_pixel,
                                       const float* g_color, float* g_normal, float* g_raw_soft_vis, float* g_raw_env, float* g_pix,
                                       float* g_bg, hipStream_t stream);

/* Material head: the output activations of the per-sample nets and the integrands of the prior terms in one
 * launch each way (python/network.py:262, 335, 423, 456-463, 498-508; python/loss.py:117-166).
 * Inputs are the nets' raw outputs for the R x N foreground samples: base colour (P,3), base colour of the
 * perturbed pass (P,3), implicit illumination (P), photogrammetric light (P) with its 1-element gain,
 * roughness (P,2) = [value, std], specular reflectance (P,6) = [value x3, std x3].
 * Outputs: V (P,9) = [implicit, roughness, specular x3, photo, base colour (x photo when entangle) x3], the
 * integrands of ONE VR integral; aux (P,10) = [base x3, base_ptb x3, std_roughness, std_specular x3];
 * prior (R,5) = per-ray sums over the samples of |base - base_ptb| (3 ch), |r - prior_r| / std_r,
 * clamp(log std_r, 1e-5, 1e5), sum_c |s_c - prior_s| / std_s_c, sum_c clamp(log std_s_c, 1e-5, 1e5).
 * remap: roughness^2 / 0.16 s^2 (filament, python/network.py:459, 502) else specular_scale * s.
 * Backward: gV (P,9), g_prior (R,5, may be null) -> gradients of the six raw inputs (aux carries none). */
int ndjir_render_material_head(int R, int N, const float* raw_base_color, const float* raw_base_color_ptb,
                               const float* raw_implicit, const float* raw_photo, const float* photo_gain,
                               const float* raw_roughness, const float* raw_specular, int remap, int entangle,
                               int sym_backward, float roughness_lower_bound, float specular_scale, float roughness_prior,
                               float specular_prior, float* V, float* aux, float* prior, hipStream_t stream);
int ndjir_render_material_head_backward(int R, int N, const float* raw_base_color, const float* raw_base_color_ptb,
                                        const float* raw_implicit, const float* raw_photo, const float* photo_gain,
                                        const float* raw_roughness, const float* raw_specular, int remap, int entangle,
                                        int sym_backward, float roughness_lower_bound, float specular_scale,
                                        float roughness_prior, float specular_prior, const float* gV, const float* g_prior,
                                        float* g_base_color, float* g_base_color_ptb, float* g_implicit, float* g_photo,
                                        float* g_roughness, float* g_specular, hipStream_t stream);

/* Several VR integrals of the same weights (python/renderer.py:84-87 as called at :90-176) in one launch.  Segment k:
 * x[k] (R, S[k], C[k]) with row stride ld[k], weighted by w[:, off[k] : off[k] + S[k]] of w (R, S_all); out[k] (R, C[k]).
 * Backward: g[k] (R, C[k]) or null; gx[k] (R, S[k], C[k]) or null; gw (R, S_all) = the sum over all segments (or null). */
int ndjir_render_integrate_many(int R, int S_all, const float* w, int nseg, const float* const* x, const int* ld, const int* C,
                                const int* S, const int* off, float* const* out, hipStream_t stream);
int ndjir_render_integrate_many_backward(int R, int S_all, const float* w, int nseg, const float* const* x, const int* ld,
                                         const int* C, const int* S, const int* off, const float* const* g, float* const* gx,
                                         float* gw, hipStream_t stream);

/* Pixel normal (python/renderer.py:90-91): normal = (grad_pixel + eps) / |grad_pixel + eps| per ray, and its backward. */
int ndjir_render_pixel_normal(int R, float eps, const float* grad_pixel, float* normal, hipStream_t stream);
int ndjir_render_pixel_normal_backward(int R, float eps, const float* grad_pixel, const float* g_normal, float* g_grad_pixel,
                                       hipStream_t stream);

/* ---- the per-ray tail of the step (ndjir_amd/csrc/loss.hip) -------------------------------------------------------
 * ndjir_render_pixel_compose: python/renderer.py:163-178 for the fused material head -- pix (R,9) = VR of
 *   [implicit, roughness, specular x3, photo, base term x3], env (R,Ce) the diffuse light integral (Ce = 1 or 3), spec (R,3),
 *   bg (R,3, may be null): diffuse = env + implicit; color = base * diffuse + photo * spec (entangle) or
 *   photo * (base * diffuse + spec); color += bg.  The reference: ~6 nnabla functions, twice as many in backward.
 * ndjir_loss_terms: python/loss.py:59-178 without the mask term (train.mask_weight = 0) -- RGB error (l1 / l2), eikonal
 *   term, sampled TV term(s), the five prior / regulariser sums of the material head (prior (R,5), may be null) and the
 *   weighted total, normalised by sum(mask) N + 1e-5 (mask_sum_global: device scalar holding the sum over all ray shards,
 *   or null = this call's own sum); the five prior sums by sum(mask) N_prior + 1e-5 (python/loss.py:36, 72, 118: N_prior is
 *   the sample count when the eikonal term is on, renderer.n_samples0 when it is off).  A term whose weight is 0 is
 *   reported as 0 and left out of the total (the reference does not evaluate it).  weights5 (host) = eikonal, tv, base colour prior, roughness prior, specular prior.
 *   workspace: ndjir_loss_terms_workspace(R) floats; terms (device, 12 floats): total, rgb, eikonal, tv, prior base colour,
 *   prior roughness, reg std roughness, prior specular, reg std specular, 1 / denorm, sum(mask), 1 / denorm of the priors.  Per-ray partial sums
 *   and a fixed-order final reduction: the loss is bit-reproducible.  The backward takes the gradient of terms[0]. */
int ndjir_render_pixel_compose(int R, int Ce, int entangle, const float* pix, const float* env, const float* spec, const float* bg,
                               float* color, hipStream_t stream);
int ndjir_render_pixel_compose_backward(int R, int Ce, int entangle, const float* pix, const float* env, const float* spec,
                                        const float* g, float* g_pix, float* g_env, float* g_spec, float* g_bg, hipStream_t stream);
int ndjir_loss_terms_workspace(int R);
int ndjir_loss_terms(int R, int N, int N_prior, const float* color, const float* color_gt, const float* mask, const float* grad_x,
                     const float* tv0, int D0, const float* tv1, int D1, const float* prior, const float* mask_sum_global,
                     float inv_rays, const float* weights5, int l2, float* workspace, float* terms, hipStream_t stream);
int ndjir_loss_terms_backward(int R, int N, const float* color, const float* color_gt, const float* mask, const float* grad_x,
                              int D0, int D1, const float* terms, const float* g_loss, float inv_rays, const float* weights5,
                              int l2, float* g_color, float* g_grad_x, float* g_tv0, float* g_tv1, float* g_prior,
                              hipStream_t stream);

/* Positional encoding (python/network.py:96-117): out (P, [C +] 2 C M) = [x, cos(x_i 2^k), sin(x_i 2^k)],
 * band index k fastest; backward gx (P,C) from g (P, [C +] 2 C M). */
int ndjir_positional_encoding(long long P, int C, int M, int include_input, const float* x, float* out, hipStream_t stream);
int ndjir_positional_encoding_backward(long long P, int C, int M, int include_input, const float* x, const float* g,
                                       float* gx, hipStream_t stream);

/* Element-wise stages around the geometric network's chains (ndjir_amd/csrc/geo.hip).  The reference spells them as
 * nnabla functions inside `geometric_network` and `nn.grad([sdf], [x])` (python/network.py:96-117, 154-170,
 * python/renderer.py:51-52); here each is one launch.
 *   geo_encode            e (P, lde) = [x | cos(x_d 2^k) | sin(x_d 2^k) | seg_0 | seg_1 ...], seg_i (P, segC[i]) contiguous
 *   geo_normal            n = g0[:, :3] + J_pe(x)^T g0[:, 3:3+6M] + sum_i gq_i   (cos / sin taken from e);
 *                         with Z: sdf_out[p] = Z[p][2], then Z[p] = [x | feature (kept) | n | 0 ...]  (row stride ldz)
 *   geo_backward_begin    gy (P, 1+D) = [g_sdf | g_feat + gZ[:, 3:3+D]], nbar (P, 3) = g_n + gZ[:, 3+D:6+D]; null = zero
 *   geo_gbar0             gb0 (P, 3+6M+sum segC) = [nbar | -sin nbar_d 2^k | cos nbar_d 2^k | seg_0 | ...]
 *   copy_columns          dst[p][0:C] = src[p][0:C] for row strides lds / ldd
 *   inverse_squared_distance  out[p * ldo] = 1 / (|x_p - camloc_b|^2 + 1e-5)   (python/network.py:405-409) */
int ndjir_geo_encode(long long P, int M, const float* x, int nseg, const float* const* seg, const int* segC, float* e, int lde,
                     hipStream_t stream);
int ndjir_geo_normal(long long P, int M, const float* e, int lde, const float* g0, int ldg, int nseg, const float* const* gq,
                     float* n_out, float* Z, int ldz, int D, float* sdf_out, hipStream_t stream);
int ndjir_geo_backward_begin(long long P, int D, const float* g_sdf, const float* g_feat, int ldf, const float* g_n,
                             const float* gZ, int ldz, float* gy, float* nbar, hipStream_t stream);
int ndjir_geo_gbar0(long long P, int M, const float* e, int lde, const float* nbar, int nseg, const float* const* seg,
                    const int* segC, float* gb0, hipStream_t stream);
int ndjir_copy_columns(long long P, int C, const float* src, int lds, float* dst, int ldd, hipStream_t stream);
int ndjir_inverse_squared_distance(long long P, long long rows_per_batch, const float* x, int ldx, const float* camloc, float* out,
                                   int ldo, hipStream_t stream);

/* Chain groups (reference: python/renderer.py:113-128 -- the per-sample material nets read the same row of every sample;
 * BASELINE.json north_star: "the small MLPs fused per-sample").  Between _begin and _end the chain calls of the calling thread
 * are RECORDED (validated, nothing launched; the pointer arrays need not outlive the call).  _end launches them in call
 * order: consecutive calls of one mode on the same number of points that the 128-point-tile kernel can take together (f16x3
 * arithmetic; at most 3; bias-gradient accumulators of all fit the LDS) as ONE launch in which a workgroup takes its tile
 * through the nets in turn -- the tile's input row is re-staged from L2, a backward group's accumulating calls (accum_y bit 0)
 * add to the gradient tile the first call wrote while it is still in cache -- and everything else as the calls alone would
 * have run.  Results equal those of the separate launches bit for bit.  *launches (may be NULL) = chain kernel launches issued.
 * NDJIR_ERR_ARG: _begin with a group open, _end without one, more than 16 calls in a group. */
int ndjir_mlp_chain_group_begin(void);
int ndjir_mlp_chain_group_end(hipStream_t stream, int* launches);

/* ---- fused MLP engine ------------------------------------------------------------------------------
 * Replaces the reference's per-layer nnabla launches (PF.affine -> cuBLAS GEMM, F.softplus(beta=100),
 * python/network.py:88-93,165 and every network function :154-561) by ONE launch per net and
 * direction: a tile of 64 points runs through all layers with activations kept in LDS
 * (ndjir_amd/csrc/mlp.hip).  Weights W are (in, out) row-major as in nnabla (y = x W + b).
 *
 * ndjir_mlp_pack: W (K x N) -> MFMA-fragment order, or W^T when transpose != 0 (for backward).
 *                 dst needs ndjir_mlp_packed_size(K, N, transpose) floats.
 * ndjir_mlp_chain (bwd == 0), forward:  h_l = softplus_beta(h_{l-1} W_l + b_l), Y = h_{L-1} W_L + b_L;
 *                 side_out[l] (P x N_l) receives h_l when non-null (needed by backward).
 * ndjir_mlp_chain (bwd != 0), backward of the data path, chain input X = dL/dY:
 *                 step i uses packed W_{L-i}^T; side_in[i] = stored forward activation of the layer
 *                 below (softplus' = 1 - exp(-beta h)); side_out[i] receives delta of that layer
 *                 (P x N), bgrad[i] (N) its column sums = the bias gradient (overwritten; summed per
 *                 workgroup in LDS, then across workgroups from `workspace`, which must hold
 *                 ndjir_mlp_chain_workspace(sum of the N_i that have a bgrad) floats -- may be null
 *                 when no bgrad is requested); in_bgrad (K0 floats, may be null) receives the column sums of
 *                 the chain input dL/dY itself = the bias gradient of the net's output layer (its N counts
 *                 towards the workspace size); with has_output the last step writes dL/dX to Y.
 *                 Weight gradients are plain GEMMs H^T delta (ndjir_mlp_wgrad).
 * Per-layer arrays are HOST arrays of length L; Ks/Ns are each step's logical input/output width.
 * skip_layer (-1 = none): forward, the output of that layer is scaled by skip_scale and the scaled
 * chain input is appended (python/network.py:221-224); backward, the step whose output is the
 * gradient of that concatenation: columns >= skip_split go (scaled) to Xskip. */
#define NDJIR_MATH_FP32 0     /* v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 vector rate */
#define NDJIR_MATH_BF16X6 1   /* x = hi + mid + lo in bf16 (exact), six bf16 MFMA partial products accumulated in
                                 fp32: ~2^-24 relative error like an fp32 FMA chain, 6/16 of the matrix time */
#define NDJIR_MATH_F16X3 2    /* x s = hi + lo 2^-11 in f16 (s: power of two per scaling group), three f16 MFMA partial
                                 products in two fp32 accumulators: 22 significant bits per operand, error below an fp32
                                 FMA chain's, 3/16 of the matrix time (default) */
int ndjir_mlp_set_math(int math);   /* selects the arithmetic of pack / chain; packed weights are mode specific */
int ndjir_mlp_get_math(void);
/* Points per workgroup tile of the f16x3 chain kernels: 0 = per launch (128 for launches of >= 32768 points that the
 * wide-tile kernel supports, else 64; 32 for small launches), 32 / 64 / 128 = forced (128: where supported).  A point's
 * forward result does not depend on the tile height. */
int ndjir_mlp_set_tile_rows(int rows);
int ndjir_mlp_get_tile_rows(void);
/* Which TRAINING-pass chain launches (every hidden layer stores its point-blocked side tensor) of nets wider than 128 columns
 * run on the software-pipelined 128-point-tile kernel (csrc/mlp3p.hip): bit 0 forward, bit 1 backward, bit 2 tangent; 0 = none.
 * That kernel keeps ONE fp32 accumulator per block: its results agree with the other f16x3 kernels to round-off (2e-6), not bit
 * for bit -- passes without side tensors (the sampler's SDF rounds, the SDF volume, render_image) never run on it.  NDJIR_CHAINP
 * sets the initial value. */
/* y (P, N; row stride ldy) = x (P, K; ldx) W (K, N; ldw) + bias (may be null); transpose != 0: y = x W^T with W (N, K; ldw).  For
 * FEW rows (the per-ray terms of the first layers of python/network.py:438, 528, 619 and their input gradients): exact fp32
 * products (v_mfma_f32_32x32x2_f32), one wave per 32 x 32 output tile, operands read where they lie -- no packed weights. */
int ndjir_mlp_small_affine(long long P, const float* x, int ldx, int K, const float* W, int ldw, int N, int transpose,
                           const float* bias, float* y, int ldy, hipStream_t stream);
int ndjir_mlp_set_chain_pipeline(int mask);
int ndjir_mlp_get_chain_pipeline(void);
long long ndjir_mlp_packed_size(int K, int N, int transpose);
int ndjir_mlp_pack(const float* W, float* dst, int K, int N, int transpose, hipStream_t stream);
/* f16x3 arithmetic: W (K, N) with row stride ldw (a column slice of a wider matrix packs without a copy) */
int ndjir_mlp_pack_strided(const float* W, int ldw, float* dst, int K, int N, int transpose, hipStream_t stream);
/* f16x3 arithmetic: re-pack many matrices in ONE launch (what the optimizer step issues after its update instead of one
 * pack per weight and orientation).  table: DEVICE array of n entries of ndjir_mlp_pack_entry_bytes() = 48 bytes:
 * { const float* W; float* dst; int K, N, ldw, transpose, Kp, Np, first_block, pad } -- Kp = round_up(transpose ? N : K, 16),
 * Np = round_up(transpose ? K : N, 32), first_block = running sum of Np / 32; total_blocks = that sum over all entries. */
int ndjir_mlp_pack_table(const void* table, int n, int total_blocks, hipStream_t stream);
int ndjir_mlp_pack_entry_bytes(void);
/* ndjir_mlp_chain / _ex: `accum_y` bit 0 = Y += result; bit 1 = the bias gradients (bgrad of every layer, in_bgrad) are
 * ADDED to their destinations instead of overwriting them -- nnabla's `accum` protocol for parameters whose gradient
 * buffer several operators share (python/train.py:136-140 zeroes the gradients once per iteration, every backward adds);
 * bit 2 (NDJIR_MATH_F16X3) = DEFER the bias-gradient reduction: the launch leaves its per-workgroup partial rows in
 * `workspace` (layout: ndjir_mlp_chain_bias_partials) and writes no bgrad / in_bgrad -- the caller sums them later, for
 * all launches of a step at once (ndjir_mlp_wgrad_group's extra outputs);
 * bit 3 (NDJIR_MATH_F16X3, P % 32 == 0; NDJIR_ERR_UNSUPPORTED otherwise) = POINT-BLOCKED HIDDEN TENSORS: side_in / side_in2 /
 * side_add / side_out / side_out2 hold element (p, f) at float offset ((p >> 5) * ld_side + f) * 32 + (p & 31) -- blocks of 32
 * points, feature-major inside a block, same P * ld_side floats as the row-major tensor.  These tensors never leave the
 * engine (forward writes them, backward / tangent / the weight gradients read them; the reference keeps nnabla's
 * intermediate Variables of python/network.py:154-232 the same way), so their layout is free: in this one a lane of the
 * 128-point-tile kernel moves its MFMA accumulator registers as they are -- register i of 32 lanes = 32 consecutive points of
 * one feature = one 128-byte line -- and ndjir_mlp_wgrad_group reads 4 consecutive points of a feature (an MFMA operand
 * group of its reduction over the points) as one 16-byte load.  X, Y, Xskip, row_bias and the bias gradients stay row-major. */
int ndjir_mlp_chain(int bwd, long long P, const float* X, int ldx, int K0, int L,
                    const float* const* Wp, const float* const* bias, const int* Ks, const int* Ns,
                    const float* const* side_in, float* const* side_out, const int* ld_side,
                    float* const* bgrad, float* Y, int ldy, int accum_y, int has_output, float beta,
                    int skip_layer, float skip_scale, int skip_split, float* Xskip, int ld_xskip,
                    float* in_bgrad, float* workspace, unsigned* const* side_amax, unsigned* x_amax, hipStream_t stream);
/* side_amax (host array of L device pointers, or null; entries may be null) / x_amax (device pointer or null): slots
 * that receive, by atomic max, the bit pattern of the largest finite |value| written to side_out[i] / read from X.
 * The caller zeroes them.  NDJIR_MATH_F16X3 only (ignored otherwise): ndjir_mlp_wgrad takes them as the scales of its
 * operands, which saves it a pass over the tensors. */
long long ndjir_mlp_chain_workspace(int bgrad_total);   /* floats */
/* Extended chain used by the geometric network, whose output gradient d(sdf)/dx itself enters the
 * loss (nn.grad, python/renderer.py:52; eikonal term python/loss.py:68-76):
 *   mode 0 / 1: as ndjir_mlp_chain, plus side_add[i] (P x N): extra adjoint added to delta after the
 *               softplus' product (mode 1);
 *   mode 2:     tangent chain of the double backward -- forward-direction weights, no bias; step l
 *               computes s-bar_l = g-bar_l W_l, writes g-bar_{l+1} = s-bar_l * softplus'(z_l) to
 *               side_out[l] and the extra adjoint beta * s-bar_l * s_l * exp(-beta h_l) to
 *               side_out2[l], with h_l = side_in[l] and s_l = side_in2[l] (delta of the sdf chain). */
int ndjir_mlp_chain_ex(int mode, long long P, const float* X, int ldx, int K0, int L,
                       const float* const* Wp, const float* const* bias, const int* Ks, const int* Ns,
                       const float* const* side_in, float* const* side_out, const int* ld_side,
                       float* const* bgrad, float* Y, int ldy, int accum_y, int has_output, float beta,
                       int skip_layer, float skip_scale, int skip_split, float* Xskip, int ld_xskip,
                       const float* const* side_in2, const float* const* side_add, float* const* side_out2,
                       const float* row_bias, int row_bias_div, float* in_bgrad, float* workspace,
                       unsigned* const* side_amax, unsigned* x_amax, hipStream_t stream);
/* row_bias (mode 0, may be null): (P / row_bias_div, N_0) term added to the first layer's pre-activation of every
 * group of row_bias_div consecutive rows -- the part of x W_0 that is constant over a group (e.g. the per-ray inputs
 * of the soft-visibility net, python/network.py:339-377, whose other inputs vary per light direction): it is
 * computed once per group instead of once per row, and the broadcast inputs are never materialised.  Its gradient
 * is the group-wise column sum of the first layer's delta: */
int ndjir_mlp_group_colsum(const float* X, int ldx, int N, long long G, int div, float* out, int blocked, hipStream_t stream);
/* blocked != 0: X is point-blocked (ndjir_mlp_chain accum_y bit 3; G * div % 32 == 0). */
/* Weight gradient of one layer: out (K x N) (+)= A^T B with A (P x K, row stride lda) the layer's
 * input activations and B (P x N, row stride ldb) its deltas (ndjir_amd/csrc/wgrad.hip; the
 * reference gets this from nnabla's affine backward, a cuBLAS GEMM).  `workspace` needs
 * ndjir_mlp_wgrad_workspace(K, N, P) floats (split-P partial sums).  amax_a / amax_b (device, may be null):
 * recorded maxima of A / B (see ndjir_mlp_chain; any upper bound of the largest finite magnitude will do) -- the
 * operand scales of NDJIR_MATH_F16X3; null = found by one extra pass over the tensor. */
long long ndjir_mlp_wgrad_workspace(int K, int N, long long P);
int ndjir_mlp_wgrad(const float* A, int lda, const float* B, int ldb, int K, int N, long long P, float* out,
                    int accum, float* workspace, const unsigned* amax_a, const unsigned* amax_b, hipStream_t stream);
/* Many weight gradients in ONE launch (+ one split-reduction launch): out[o] (K[o] x N[o], row stride ldo[o]) (+)= the sum
 * over the sources i with out_id[i] == o of A[i]^T B[i] (A[i]: P[i] x K[o], row stride lda[i]; B[i]: P[i] x N[o], row stride
 * ldb[i]).  A training step's ~47 per-layer GEMMs (nnabla's affine backward, python/network.py:88-93) share the machine:
 * splits of thousands of points instead of hundreds, partial slabs smaller by the number of layers grouped, one reduction
 * launch for all.  Two sources of one output: the geometric network's dW_j = A_j^T delta_j + gbar_j^T s_j (its double
 * backward through nn.grad, python/renderer.py:52).  NDJIR_MATH_F16X3 only (NDJIR_ERR_UNSUPPORTED otherwise); amax_a[i] /
 * amax_b[i]: device, recorded maxima as for ndjir_mlp_wgrad; a null entry makes every work item find the maximum of the
 * values it multiplies itself (one extra pass over them: meant for the few-hundred-row per-ray layers); outputs <= 8 wide
 * take a streaming path that needs none.  `workspace`: ndjir_mlp_wgrad_group_workspace(...) floats for the same sources / outputs / target_items
 * (work items the launch aims for: the point axis of every source is split accordingly; 0 = default, 4 per CU). */
long long ndjir_mlp_wgrad_group_workspace(int n_src, const float* const* A, const int* lda, const long long* P, const int* out_id,
                                          int n_out, const int* K, const int* N, int target_items, const int* layout);
/* (diagnostics) launches of the grouped kernel -- and of its reduction -- that the call issues: one per 24 operand pairs */
int ndjir_mlp_wgrad_group_launches(int n_src, const long long* P, const int* out_id, int n_out);
/* n_extra reduce-only outputs ride in the same reduction launch: ex_out[i] (ex_n[i] floats) (+)= the sum of ex_S[i] partial
 * rows at ex_partial[i] + s * ex_stride[i] -- the bias gradients that chain launches with the DEFER flag left in their
 * workspaces (ndjir_mlp_chain accum_y bit 2, ndjir_mlp_chain_bias_partials), so that a training step's ~13 per-launch
 * bias reductions become none. */
int ndjir_mlp_wgrad_group(int n_src, const float* const* A, const int* lda, const float* const* B, const int* ldb,
                          const long long* P, const unsigned* const* amax_a, const unsigned* const* amax_b, const int* out_id,
                          int n_out, float* const* out, const int* ldo, const int* K, const int* N, const int* accum,
                          float* workspace, int target_items, int n_extra, float* const* ex_out, const float* const* ex_partial,
                          const int* ex_n, const int* ex_S, const int* ex_stride, const int* ex_accum, const int* layout,
                          hipStream_t stream);
/* layout (host array of n_src ints, or NULL = every operand row-major): bit 0 = A[i] is point-blocked (ndjir_mlp_chain accum_y
 * bit 3; lda[i] = its ld_side), bit 1 = B[i] is; a blocked operand needs P[i] % 32 == 0.  The same array goes to
 * ndjir_mlp_wgrad_group_workspace (it decides which outputs <= 8 wide take the streaming path). */
/* Bias gradient of a layer: out (N) (+)= column sums of its deltas X (P x N, row stride ldx); the
 * reference gets it from nnabla's affine backward (a reduction kernel per layer). */
long long ndjir_mlp_colsum_workspace(int N, long long P);   /* floats */
int ndjir_mlp_colsum(const float* X, int ldx, int N, long long P, float* out, int accum, float* workspace,
                     hipStream_t stream);
/* Diagnostics (no reference counterpart): the symbol -- as rocprofv3 prints it, at most 63 characters -- of the kernel that
 * ndjir_mlp_chain / ndjir_mlp_chain_ex would launch for a chain of this shape under the current arithmetic and tile setting
 * (the launchers' own decision code; nothing is launched).  name_bytes >= 64. */
int ndjir_mlp_chain_kernel(int mode, long long P, int K0, int L, const int* Ks, const int* Ns, int has_output, int skip_layer,
                           int skip_split, int with_bias_gradients, char* name, int name_bytes);
/* Layout of the bias-gradient partial rows that a backward / tangent chain launch with the DEFER flag (accum_y bit 2) leaves
 * in its workspace instead of reducing them itself: *blocks rows of *row_floats floats -- the layers named by bgrad_mask
 * (bit i = layer i) in order, then the chain input's column sums when in_bgrad.  Same decision code as the launch; nothing
 * is launched.  NDJIR_ERR_UNSUPPORTED: the current arithmetic's kernels do not defer. */
int ndjir_mlp_chain_bias_partials(int mode, long long P, int K0, int L, const int* Ks, const int* Ns, int has_output,
                                  int skip_layer, int skip_split, unsigned bgrad_mask, int in_bgrad, int* blocks, int* row_floats);
/* Diagnostics (no reference counterpart): with a non-null device buffer of 10 * 5 * 8 int64, later
 * chain launches record shader-clock stamps of workgroup 0, [layer][phase][wave], phases = layer
 * start / k-loop done / accumulators staged / epilogue done / barrier passed.  Null switches it off. */
int ndjir_mlp_debug_timeline(long long* buf);


/* ---- camera rays -----------------------------------------------------------------------------------
 * Replaces the per-iteration host computation `generate_raydir_camloc` (python/helper.py:44-73, called at
 * python/train.py:131-133 and python/renderer.py:238) and its two host->device copies: x_w = R_c2w K^-1 (x, y, 1)^T,
 * normalised; camloc = pose[:3, 3].  pose (B,4,4) and intrinsic (B,3,3) are double (the reference keeps them in float64
 * numpy), arithmetic in double, outputs fp32.  Pixels: flat indices pixel_index (B,R) int32 into a W-wide image
 * (python/dataset.py:96-101) or coordinates xy (B,R,2) fp32 -- exactly one of the two non-null. */
int ndjir_generate_raydir_camloc(int B, int R, const double* pose, const double* intrinsic, const int* pixel_index,
                                 const float* xy, int W, float* raydir, float* camloc, hipStream_t stream);

/* ---- optimizer step ------------------------------------------------------------------------------
 * Replaces nnabla's `S.Adam` update + `weight_decay` + `zero_grad` + `check_inf_or_nan_grad` as the reference
 * calls them (python/solver.py:29-30, 48-50, 60-69; python/train.py:136-148).  nnabla 1.29.0's update rule:
 *   m <- beta1 m + (1-beta1) g;  v <- beta2 v + (1-beta2) g^2;  w <- w - alpha_t m / (sqrt(v) + eps),
 *   alpha_t = alpha sqrt(1-beta2^t) / (1-beta1^t)  (formed by the caller from its step counter t),
 * with g = dL/dw + decay * w (the reference adds decay * w to the zeroed buffer before backward accumulates).
 * ndjir_solver_adam: one dense parameter of n floats (16-byte aligned); zero_grad != 0 clears g in the same pass.
 * ndjir_adam_state (device memory, 16 bytes): keeps the learning rate, nnabla's step counter t, the step size of
 * the current step and the guard's verdict on the device, so that a captured HIP graph replays the step under a
 * changing learning rate and a vetoed update (python/train.py:141-143) costs no host round trip.
 * ndjir_solver_adam_begin: skipped = flag_a && flag_b (the reference combines the two solvers' guards with `and`,
 * python/solver.py:67-69; one flag given: that flag; none: never); if not skipped t += 1 and alpha_t is formed.
 * The update functions take `state` (may be null): non-null = use state->alpha_t instead of the argument and, when
 * state->skipped, leave w, m, v untouched (g is still cleared if zero_grad). */
typedef struct ndjir_adam_state { float alpha; int t; float alpha_t; int skipped; } ndjir_adam_state;
int ndjir_solver_adam_begin(void* state, float beta1, float beta2, const int* flag_a, const int* flag_b,
                            hipStream_t stream);
int ndjir_solver_adam(long long n, float* w, float* g, float* m, float* v, float alpha_t, float beta1, float beta2,
                      float eps, float decay, int zero_grad, const void* state, hipStream_t stream);
/* Sparse-gradient variant for the voxel grid (zero_grad implied): `touched` holds one bit per float4 of g, set where g
 * may be non-zero (ndjir_voxel_feature_mark_touched with the step's query points); g is read and cleared only there
 * (24 instead of 32 bytes per parameter) and the bitmap comes back all zero.  n % 128 == 0. */
int ndjir_solver_adam_touched(long long n, float* w, float* g, float* m, float* v, float alpha_t, float beta1, float beta2,
                              float eps, float decay, unsigned* touched, const void* state, hipStream_t stream);
int ndjir_voxel_feature_mark_touched(int N, const float* query, const int* grid_sizes, int D, const float* min,
                                     const float* max, unsigned* bitmap, hipStream_t stream);
/* the same over `count` small tensors given as host arrays of device pointers; g[k] null = zero gradient */
int ndjir_solver_adam_multi(int count, float* const* w, const float* const* g, float* const* m, float* const* v,
                            const long long* numel, float alpha_t, float beta1, float beta2, float eps, float decay,
                            const void* state, hipStream_t stream);
/* *flag |= 1 when g holds an inf or nan (flag zeroed by the caller) */
int ndjir_solver_check_inf_or_nan(long long n, const float* g, int* flag, hipStream_t stream);
int ndjir_solver_check_inf_or_nan_multi(int count, const float* const* g, const long long* numel, int* flag,
                                        hipStream_t stream);
/* python/train.py:144-146 (`if np.any(np.isnan(loss.d)): continue`): raises both guard flags when x (n floats, the
 * loss) holds a NaN, so that the `and` in ndjir_solver_adam_begin skips the update */
int ndjir_solver_veto_if_nan(int n, const float* x, int* flag_a, int* flag_b, hipStream_t stream);
/* *out += sum x^2 (device double): the norm of `clip_grad_by_norm` (python/solver.py:53-58) */
int ndjir_solver_sum_squares(long long n, const float* x, double* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NDJIR_HIP_H */
