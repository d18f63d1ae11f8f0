// capi_grid.hip -- C-ABI entry points for the grid-feature / TV / intersection / sampling ops.
// Declarations + reference citations: include/ndjir_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/ndjir_hip.h"
#include "common.h"
#include "grid.h"

using namespace ndjir;

// --- hash-grid level table: bit-for-bit the host arithmetic of common_voxel_hash.cuh:24-55 ------
extern "C" int ndjir_hash_force_align(int size, int mod) { return size + size % mod; }  // sic

extern "C" int ndjir_hash_grid_size(int G0, float growth_factor, int level) {
  double Gf = floor(G0 * pow((double)growth_factor, (double)level));
  return (int)Gf;
}

extern "C" int ndjir_hash_table_size(int G, int T0) {
  float Gf = (float)G;
  float T = fminf(Gf * Gf * Gf, (float)T0);
  int Ti = (int)T;
  return Ti < T0 ? Ti : T0;
}

extern "C" long long ndjir_hash_num_params(int G0, float growth_factor, int T0, int L, int D) {
  long long n = 0;
  for (int l = 0; l < L; ++l) {
    int G = ndjir_hash_grid_size(G0, growth_factor, l);
    int T = ndjir_hash_table_size(G, T0);
    n += ndjir_hash_force_align(T * D, 8);
  }
  return n;
}

static void set_box(GridDesc& g, const float* mn, const float* mx) {
  for (int i = 0; i < 3; ++i) { g.mn[i] = mn[i]; g.mx[i] = mx[i]; }
}

static GridDesc voxel_desc(const int* gs, int D, const float* mn, const float* mx) {
  GridDesc g{};
  g.topo = VOXEL; g.G[0] = gs[0]; g.G[1] = gs[1]; g.G[2] = gs[2]; g.D = D; g.S = 1;
  set_box(g, mn, mx);
  return g;
}

static GridDesc plane_desc(int topo, int G, int D, const float* mn, const float* mx) {
  GridDesc g{};
  g.topo = topo; g.G[0] = g.G[1] = g.G[2] = G; g.D = D; g.S = 3;
  set_box(g, mn, mx);
  return g;
}

static int hash_desc(GridDesc& g, int G0, float gf, int T0, int L, int D, const float* mn, const float* mx) {
  if (L < 1 || L > MAX_LEVELS) return NDJIR_ERR_UNSUPPORTED;
  g = GridDesc{};
  g.topo = HASH; g.D = D; g.S = L;
  set_box(g, mn, mx);
  long long off = 0;
  for (int l = 0; l < L; ++l) {
    g.lvlG[l] = ndjir_hash_grid_size(G0, gf, l);
    g.lvlT[l] = ndjir_hash_table_size(g.lvlG[l], T0);
    g.lvlOff[l] = off;
    off += ndjir_hash_force_align(g.lvlT[l] * D, 8);
  }
  return NDJIR_OK;
}

static long long desc_numel(const GridDesc& g, int G0 = 0, float gf = 0, int T0 = 0) {
  switch (g.topo) {
    case VOXEL: return (long long)g.G[0] * g.G[1] * g.G[2] * g.D;
    case TRIPLANE: return 3LL * g.G[0] * g.G[0] * g.D;
    case TRILINE: return 3LL * g.G[0] * g.D;
    default: return ndjir_hash_num_params(G0, gf, T0, g.S, g.D);
  }
}

// shared bodies ---------------------------------------------------------------------------------
#define CHECK_PTRS_N(N_, ...) do { if ((N_) <= 0) return NDJIR_OK; CHECK_PTRS(__VA_ARGS__); } while (0)
#define CHECK_PTRS(...) do { const void* _p[] = {__VA_ARGS__}; for (auto q : _p) if (!q) return NDJIR_ERR_ARG; } while (0)

static int do_query(int interp, const GridDesc& g, long long P, float* out, const float* q, const float* f, hipStream_t s) {
  if (P <= 0) return NDJIR_OK;
  CHECK_PTRS(out, q, f);
  return launch_query(interp, g, P, out, q, f, false, s);
}
static int do_grad_query(int interp, const GridDesc& g, long long P, float* gq, const float* go, const float* q, const float* f, int accum, hipStream_t s) {
  if (P <= 0) return NDJIR_OK;
  CHECK_PTRS(gq, go, q, f);
  return launch_dquery(interp, g, P, 0, gq, go, q, f, accum != 0, s);
}
static int do_grad_feature(int interp, const GridDesc& g, long long numel, long long P, float* gf, const float* go, const float* q, int accum, hipStream_t s) {
  if (!gf) return NDJIR_ERR_ARG;
  if (!accum) zero_fill(gf, numel, s);
  if (P <= 0) return ndjir_check_launch();
  CHECK_PTRS(go, q);
  return launch_scatter(interp, g, P, 0, gf, nullptr, go, q, s);
}
static int do_ggo(int interp, const GridDesc& g, long long P, float* ggo, const float* ggq, const float* q, const float* f, int accum, hipStream_t s) {
  if (P <= 0) return NDJIR_OK;
  CHECK_PTRS(ggo, ggq, q, f);
  return launch_dquery(interp, g, P, 1, ggo, ggq, q, f, accum != 0, s);
}
static int do_gq_gf(int interp, const GridDesc& g, long long P, float* gf, const float* ggq, const float* go, const float* q, hipStream_t s) {
  if (P <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, ggq, go, q);
  return launch_scatter(interp, g, P, 1, gf, ggq, go, q, s);   // never zeroes (reference behaviour)
}

// dense voxel families -----------------------------------------------------------------------------
#define VOXEL_FAMILY(PREFIX, INTERP)                                                                                   \
  extern "C" int ndjir_##PREFIX##_query_on_voxel(int N, float* output, const float* query, const float* feature,      \
      const int* gs, int D, const float* mn, const float* mx, int bc, hipStream_t st) {                                \
    (void)bc; return do_query(INTERP, voxel_desc(gs, D, mn, mx), N / D, output, query, feature, st); }                 \
  extern "C" int ndjir_##PREFIX##_grad_query(int N, float* gq, const float* go, const float* query, const float* feature, \
      const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {                     \
    (void)bc; return do_grad_query(INTERP, voxel_desc(gs, D, mn, mx), N / D, gq, go, query, feature, accum, st); }     \
  extern "C" int ndjir_##PREFIX##_grad_feature(int N, float* gf, const float* go, const float* query,                  \
      const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {                     \
    (void)bc; GridDesc g = voxel_desc(gs, D, mn, mx);                                                                  \
    return do_grad_feature(INTERP, g, desc_numel(g), N / D, gf, go, query, accum, st); }                               \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_grad_output(int N, float* ggo, const float* ggq, const float* query, \
      const float* feature, const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) { \
    (void)bc; return do_ggo(INTERP, voxel_desc(gs, D, mn, mx), N / D, ggo, ggq, query, feature, accum, st); }          \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_feature(int N, float* gf, const float* ggq, const float* go,         \
      const float* query, const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) { \
    (void)bc; (void)accum; return do_gq_gf(INTERP, voxel_desc(gs, D, mn, mx), N / D, gf, ggq, go, query, st); }

VOXEL_FAMILY(voxel_feature, LINEAR)
VOXEL_FAMILY(cosine_voxel_feature, COSINE)
VOXEL_FAMILY(lanczos_voxel_feature, LANCZOS)

// linear dense voxel only: the remaining second-order entry points (voxel_feature_cuda.cu:440-814)
extern "C" int ndjir_voxel_feature_grad_query_grad_query(int N, float* gq, const float* ggq, const float* go,
    const float* query, const float* feature, const int* gs, int D, const float* mn, const float* mx, int bc, int accum,
    hipStream_t st) {
  (void)bc; (void)accum;
  CHECK_PTRS(gq, ggq, go, query, feature);
  return launch_voxel_gq_gq(voxel_desc(gs, D, mn, mx), N / D, gq, ggq, go, query, feature, st);
}
// No reference counterpart (the reference zero-fills gradients densely): zero only the cells the N query
// points touch in an accumulate-in-place gradient buffer of the linear dense voxel grid.
// [x | cos | sin | voxel feature] rows of the geometric net's input in one launch (interp 0 linear / 1 cosine / 2 Lanczos)
extern "C" int ndjir_voxel_feature_query_encode(int N, int M, const float* query, const float* feature, const int* gs, int D,
                                                const float* mn, const float* mx, int interp, float* e, int lde, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  if (!query || !feature || !gs || !mn || !mx || !e || M < 0 || M > 30 || D < 1 || lde < 3 + 6 * M + D) return NDJIR_ERR_ARG;
  return launch_voxel_query_encode(interp, voxel_desc(gs, D, mn, mx), N, M, query, feature, e, lde, st);
}

// ... and for the tri-plane + tri-line pair: rows [x | cos | sin | tri-plane feature (Dp, 3) | tri-line feature (Dl, 3)]
extern "C" int ndjir_triplaneline_query_encode(int N, int M, const float* query, const float* plane, int Gp, int Dp, const float* line,
                                               int Gl, int Dl, const float* mn, const float* mx, int interp, float* e, int lde,
                                               hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  if (!query || !plane || !line || !mn || !mx || !e || M < 0 || M > 30 || Dp < 1 || Dl < 1 || Gp < 1 || Gl < 1 ||
      lde < 3 + 6 * M + 3 * Dp + 3 * Dl)
    return NDJIR_ERR_ARG;
  return launch_tri_query_encode(interp, plane_desc(TRIPLANE, Gp, Dp, mn, mx), plane_desc(TRILINE, Gl, Dl, mn, mx), N, M, query, plane,
                                 line, e, lde, st);
}

extern "C" int ndjir_voxel_feature_zero_touched(int N, float* gf, const float* query, const int* gs, int D, const float* mn,
                                                const float* mx, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, query);
  return launch_zero_touched(LINEAR, voxel_desc(gs, D, mn, mx), N, gf, query, nullptr, st);
}

// bitmap |= the cells (one bit per cell, D = 4) the N query points touch -- input of ndjir_solver_adam_touched
extern "C" int ndjir_voxel_feature_mark_touched(int N, const float* query, const int* gs, int D, const float* mn, const float* mx,
                                                unsigned* bitmap, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(query, bitmap);
  return launch_mark_touched(voxel_desc(gs, D, mn, mx), N, query, bitmap, st);
}

// Sparse gradient exchange: append the non-zero rows of the cells the N query points touch (each cell once).
extern "C" int ndjir_voxel_feature_pack_rows(int N, const float* gf, const float* query, const int* gs, int D, const float* mn,
                                             const float* mx, unsigned* bitmap, int* ids, float* rows, int* count, int capacity,
                                             hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, query, bitmap, ids, rows, count);
  return launch_pack_rows(LINEAR, voxel_desc(gs, D, mn, mx), N, gf, query, bitmap, ids, rows, count, capacity, st);
}

// ... for any dense family: topo 0 voxel (grid_sizes[3]) / 1 tri-plane / 2 tri-line (grid_sizes[0] = G), interp 0 linear /
// 1 cosine / 2 Lanczos, D = 4 or 8.  A row = the D floats of one cell; cell id = float offset / D.
extern "C" int ndjir_grid_pack_rows(int topo, int interp, int N, const float* gf, const float* query, const int* gs, int D,
                                    const float* mn, const float* mx, unsigned* bitmap, int* ids, float* rows, int* count,
                                    int capacity, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, query, gs, mn, mx, bitmap, ids, rows, count);
  if (topo < VOXEL || topo > TRILINE || interp < LINEAR || interp > LANCZOS) return NDJIR_ERR_ARG;
  const GridDesc g = topo == VOXEL ? voxel_desc(gs, D, mn, mx) : plane_desc(topo, gs[0], D, mn, mx);
  return launch_pack_rows(interp, g, N, gf, query, bitmap, ids, rows, count, capacity, st);
}

// the same for the cosine (interp = 1) and Lanczos (interp = 2: 4 x 4 x 4 taps) dense voxel families
extern "C" int ndjir_voxel_feature_zero_touched_interp(int N, float* gf, const float* query, const int* gs, int D, const float* mn,
                                                       const float* mx, int interp, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, query);
  if (interp != LINEAR && interp != COSINE && interp != LANCZOS) return NDJIR_ERR_ARG;
  return launch_zero_touched(interp, voxel_desc(gs, D, mn, mx), N, gf, query, nullptr, st);
}

// *flag |= 1 when a cell of grad_feature that the N query points touch holds an inf or nan.
extern "C" int ndjir_voxel_feature_check_touched(int N, const float* gf, const float* query, const int* gs, int D,
                                                 const float* mn, const float* mx, int* flag, hipStream_t st) {
  if (N <= 0) return NDJIR_OK;
  CHECK_PTRS(gf, query, flag);
  return launch_zero_touched(LINEAR, voxel_desc(gs, D, mn, mx), N, const_cast<float*>(gf), query, flag, st);
}

extern "C" int ndjir_voxel_feature_grad_feature_grad_grad_output(int N, float* ggo, const float* ggf, const float* query,
    const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {
  (void)bc;
  CHECK_PTRS(ggo, ggf, query);
  return launch_query(LINEAR, voxel_desc(gs, D, mn, mx), N / D, ggo, query, ggf, accum != 0, st);
}
extern "C" int ndjir_voxel_feature_grad_feature_grad_query(int N, float* gq, const float* ggf, const float* go,
    const float* query, const int* gs, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {
  (void)bc; (void)accum;
  CHECK_PTRS(gq, ggf, go, query);
  return launch_dquery(LINEAR, voxel_desc(gs, D, mn, mx), N / D, 0, gq, go, query, ggf, true, st);  // never zeroes
}

// tri-plane / tri-line families --------------------------------------------------------------------
#define PLANE_FAMILY(PREFIX, FWD, TOPO, INTERP)                                                                        \
  extern "C" int ndjir_##PREFIX##_##FWD(int N, float* output, const float* query, const float* feature, int G, int D,  \
      const float* mn, const float* mx, int bc, hipStream_t st) {                                                      \
    (void)bc; return do_query(INTERP, plane_desc(TOPO, G, D, mn, mx), N / (D * 3), output, query, feature, st); }      \
  extern "C" int ndjir_##PREFIX##_grad_query(int N, float* gq, const float* go, const float* query, const float* feature, \
      int G, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {                             \
    (void)bc; return do_grad_query(INTERP, plane_desc(TOPO, G, D, mn, mx), N / (D * 3), gq, go, query, feature, accum, st); } \
  extern "C" int ndjir_##PREFIX##_grad_feature(int N, float* gf, const float* go, const float* query, int G, int D,    \
      const float* mn, const float* mx, int bc, int accum, hipStream_t st) {                                           \
    (void)bc; GridDesc g = plane_desc(TOPO, G, D, mn, mx);                                                             \
    return do_grad_feature(INTERP, g, desc_numel(g), N / (D * 3), gf, go, query, accum, st); }                         \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_grad_output(int N, float* ggo, const float* ggq, const float* query, \
      const float* feature, int G, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {       \
    (void)bc; return do_ggo(INTERP, plane_desc(TOPO, G, D, mn, mx), N / (D * 3), ggo, ggq, query, feature, accum, st); } \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_feature(int N, float* gf, const float* ggq, const float* go,         \
      const float* query, int G, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {         \
    (void)bc; (void)accum; return do_gq_gf(INTERP, plane_desc(TOPO, G, D, mn, mx), N / (D * 3), gf, ggq, go, query, st); }

PLANE_FAMILY(triplane_feature, query_on_triplane, TRIPLANE, LINEAR)
PLANE_FAMILY(cosine_triplane_feature, query_on_triplane, TRIPLANE, COSINE)
PLANE_FAMILY(lanczos_triplane_feature, query_on_triplane, TRIPLANE, LANCZOS)
PLANE_FAMILY(triline_feature, query_on_triline, TRILINE, LINEAR)
PLANE_FAMILY(cosine_triline_feature, query_on_triline, TRILINE, COSINE)
PLANE_FAMILY(lanczos_triline_feature, query_on_triline, TRILINE, LANCZOS)

// hash-grid families (N = L * P; feature outputs in the reference's (D, L, P) layout) ----------------
#define HASH_FAMILY(PREFIX, INTERP)                                                                                    \
  extern "C" int ndjir_##PREFIX##_hash_index(int N, float* output, const float* query, int G, int T,                   \
      const float* mn, const float* mx, int bc, hipStream_t st) {                                                      \
    (void)bc; CHECK_PTRS(output, query);                                                                               \
    GridDesc g{}; g.topo = HASH; g.D = 1; g.S = 1; set_box(g, mn, mx); g.lvlG[0] = G; g.lvlT[0] = T; g.lvlOff[0] = 0;  \
    return launch_hash_index(g, N, output, query, st); }                                                               \
  extern "C" int ndjir_##PREFIX##_voxel_hash_feature(int N, float* output, const float* query, const float* feature,   \
      int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc, hipStream_t st) {             \
    (void)bc; GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;                         \
    return do_query(INTERP, g, N / L, output, query, feature, st); }                                                   \
  extern "C" int ndjir_##PREFIX##_grad_query(int N, float* gq, const float* go, const float* query, const float* feature, \
      int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {  \
    (void)bc; GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;                         \
    return do_grad_query(INTERP, g, N / L, gq, go, query, feature, accum, st); }                                       \
  extern "C" int ndjir_##PREFIX##_grad_feature(int N, float* gf, const float* go, const float* query,                  \
      int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc, int accum, hipStream_t st) {  \
    (void)bc; GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;                         \
    return do_grad_feature(INTERP, g, desc_numel(g, G0, gf_, T0), N / L, gf, go, query, accum, st); }                  \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_grad_output(int N, float* ggo, const float* ggq, const float* query, \
      const float* feature, int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc,         \
      int accum, hipStream_t st) {                                                                                     \
    (void)bc; GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;                         \
    return do_ggo(INTERP, g, N / L, ggo, ggq, query, feature, accum, st); }                                            \
  extern "C" int ndjir_##PREFIX##_grad_query_grad_feature(int N, float* gf, const float* ggq, const float* go,         \
      const float* query, int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc,           \
      int accum, hipStream_t st) {                                                                                     \
    (void)bc; (void)accum; GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;            \
    return do_gq_gf(INTERP, g, N / L, gf, ggq, go, query, st); }

HASH_FAMILY(voxel_hash_feature, LINEAR)
HASH_FAMILY(lanczos_voxel_hash_feature, LANCZOS)

// sampled TV loss -----------------------------------------------------------------------------------
extern "C" int ndjir_total_variation_loss_tv_loss_on_voxel(int N, float* output, const float* query, const float* feature,
    const int* gs, int D, const float* mn, const float* mx, int bc, hipStream_t st) {
  (void)bc; CHECK_PTRS(output, query, feature);
  return launch_tv(voxel_desc(gs, D, mn, mx), N / D, false, output, nullptr, query, feature, 0, st);
}
extern "C" int ndjir_total_variation_loss_tv_loss_on_voxel_backward(int N, float* gf, const float* go, const float* query,
    const float* feature, const int* gs, int D, const float* mn, const float* mx, int sym_backward, int bc, int accum,
    hipStream_t st) {
  (void)bc; (void)accum; CHECK_PTRS(gf, go, query, feature);
  return launch_tv(voxel_desc(gs, D, mn, mx), N / D, true, gf, go, query, feature, sym_backward, st);
}
#define TV_PLANE(NAME, TOPO)                                                                                           \
  extern "C" int ndjir_total_variation_loss_on_##NAME##_tv_loss_on_##NAME(int N, float* output, const float* query,    \
      const float* feature, int G, int D, const float* mn, const float* mx, int bc, hipStream_t st) {                  \
    (void)bc; CHECK_PTRS(output, query, feature);                                                                      \
    return launch_tv(plane_desc(TOPO, G, D, mn, mx), N / (D * 3), false, output, nullptr, query, feature, 0, st); }    \
  extern "C" int ndjir_total_variation_loss_on_##NAME##_tv_loss_on_##NAME##_backward(int N, float* gf, const float* go, \
      const float* query, const float* feature, int G, int D, const float* mn, const float* mx, int sym_backward,      \
      int bc, int accum, hipStream_t st) {                                                                             \
    (void)bc; (void)accum; CHECK_PTRS(gf, go, query, feature);                                                         \
    return launch_tv(plane_desc(TOPO, G, D, mn, mx), N / (D * 3), true, gf, go, query, feature, sym_backward, st); }
TV_PLANE(triplane, TRIPLANE)
TV_PLANE(triline, TRILINE)

extern "C" int ndjir_total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash(int N, float* output, const float* query,
    const float* feature, int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx, int bc, hipStream_t st) {
  (void)bc; CHECK_PTRS(output, query, feature);
  GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;
  return launch_tv(g, N / L, false, output, nullptr, query, feature, 0, st);
}
extern "C" int ndjir_total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash_backward(int N, float* gf, const float* go,
    const float* query, const float* feature, int G0, float gf_, int T0, int L, int D, const float* mn, const float* mx,
    int sym_backward, int bc, int accum, hipStream_t st) {
  (void)bc; (void)accum; CHECK_PTRS(gf, go, query, feature);
  GridDesc g; int rc = hash_desc(g, G0, gf_, T0, L, D, mn, mx); if (rc) return rc;
  return launch_tv(g, N / L, true, gf, go, query, feature, sym_backward, st);
}

// intersection / sampling / activation ------------------------------------------------------------------
extern "C" int ndjir_ray_aabb_intersection(int N, float* t_near, float* t_far, float* n_hits, const float* camloc,
    const float* raydir, int B, int R, const float* mn, const float* mx, hipStream_t st) {
  (void)B; CHECK_PTRS(t_near, t_far, n_hits, camloc, raydir);
  return launch_ray_aabb(N, t_near, t_far, n_hits, camloc, raydir, R, mn, mx, st);
}
extern "C" int ndjir_ray_sphere_intersection(int N, float* t_near, float* t_far, float* n_hits, const float* camloc,
    const float* raydir, int B, int R, float radius, hipStream_t st) {
  (void)B; CHECK_PTRS(t_near, t_far, n_hits, camloc, raydir);
  return launch_ray_sphere(N, t_near, t_far, n_hits, camloc, raydir, R, radius, st);
}
extern "C" int ndjir_inverse_transform_sample_uniform_directions(int size, float* light_dirs, const float* normal,
    const float* cdf_the, const float* cdf_phi, int batch_size, int n_lights, int n_thes, int n_phis, float eps,
    hipStream_t st) {
  (void)batch_size; CHECK_PTRS(light_dirs, normal, cdf_the, cdf_phi);
  return launch_sample_dirs(size, light_dirs, normal, cdf_the, cdf_phi, nullptr, n_lights, n_thes, n_phis, eps, st);
}
extern "C" int ndjir_inverse_transform_sample_importance_directions(int size, float* light_dirs, const float* normal,
    const float* cdf_the, const float* cdf_phi, const float* alpha, int batch_size, int n_lights, int n_thes, int n_phis,
    float eps, hipStream_t st) {
  (void)batch_size; CHECK_PTRS(light_dirs, normal, cdf_the, cdf_phi, alpha);
  return launch_sample_dirs(size, light_dirs, normal, cdf_the, cdf_phi, alpha, n_lights, n_thes, n_phis, eps, st);
}
extern "C" int ndjir_math_expf(int size, float* output, const float* input, int sigmoid, hipStream_t st) {
  if (size > 0 && (!output || !input)) return NDJIR_ERR_ARG;
  return launch_math_expf(size, output, input, sigmoid, st);
}
extern "C" int ndjir_squareplus_forward(int size, float* output, const float* input, float b, hipStream_t st) {
  CHECK_PTRS(output, input);
  return launch_squareplus(size, false, output, nullptr, input, b, false, st);
}
extern "C" int ndjir_squareplus_backward(int size, float* dinput, const float* doutput, const float* input, float b,
    int accum, hipStream_t st) {
  CHECK_PTRS(dinput, doutput, input);
  return launch_squareplus(size, true, dinput, doutput, input, b, accum != 0, st);
}

extern "C" int ndjir_zero(float* p, long long n, hipStream_t st) {
  if (!p && n > 0) return NDJIR_ERR_ARG;
  zero_fill(p, n, st);
  return ndjir_check_launch();
}

extern "C" const char* ndjir_version(void) { return "ndjir_amd 0.1 (gfx950)"; }

// point count from which the tri-plane scatters bin their (point, plane) pairs by tile (csrc/grid.hip); < 0 restores the default
