// grid.h -- shared between grid.hip (kernels) and capi_grid.hip (C-ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace ndjir {

enum Topo { VOXEL = 0, TRIPLANE = 1, TRILINE = 2, HASH = 3 };
enum Interp { LINEAR = 0, COSINE = 1, LANCZOS = 2 };

constexpr int MAX_LEVELS = 32;

struct GridDesc {
  int topo;
  int G[3];      // dense: grid sizes (voxel Gx,Gy,Gz ; plane/line G,G,G)
  int D;         // channels per cell
  int S;         // sub-grids: 1 (voxel) / 3 (planes, lines) / L (hash levels)
  float mn[3], mx[3];
  // hash levels (host-computed, bit-identical to common_voxel_hash.cuh:24-55)
  int lvlG[MAX_LEVELS], lvlT[MAX_LEVELS];
  long long lvlOff[MAX_LEVELS];
};

int launch_query(int interp, const GridDesc& g, long long P, float* out, const float* query, const float* feature, bool accum, hipStream_t stream);
int launch_voxel_query_encode(int interp, const GridDesc& g, long long P, int M, const float* query, const float* feature, float* e,
                              int lde, hipStream_t stream);
int launch_tri_query_encode(int interp, const GridDesc& gp, const GridDesc& gl, long long P, int M, const float* query,
                            const float* plane, const float* line, float* e, int lde, hipStream_t stream);
int launch_dquery(int interp, const GridDesc& g, long long P, int mode, float* dst, const float* src, const float* query, const float* feature, bool accum, hipStream_t stream);
int launch_mark_touched(const GridDesc& g, long long P, const float* query, unsigned* bitmap, hipStream_t stream);
int launch_pack_rows(int interp, const GridDesc& g, long long P, const float* gf, const float* query, unsigned* bitmap, int* ids,
                     float* rows, int* count, int capacity, hipStream_t stream);
int launch_zero_touched(int interp, const GridDesc& g, long long P, float* gf, const float* query, int* nonfinite_flag,
                        hipStream_t stream);
int launch_scatter(int interp, const GridDesc& g, long long P, int mode, float* gf, const float* gg_query, const float* grad_output, const float* query, hipStream_t stream);
int launch_voxel_gq_gq(const GridDesc& g, long long P, float* gq, const float* gg_query, const float* grad_output, const float* query, const float* feature, hipStream_t stream);
int launch_tv(const GridDesc& g, long long P, bool bwd, float* dst, const float* grad_output, const float* query, const float* feature, int sym_backward, hipStream_t stream);
int launch_hash_index(const GridDesc& g, long long P, float* out, const float* query, hipStream_t stream);
int launch_ray_aabb(int N, float* tn, float* tf, float* nh, const float* camloc, const float* raydir, int R, const float* mn, const float* mx, hipStream_t stream);
int launch_ray_sphere(int N, float* tn, float* tf, float* nh, const float* camloc, const float* raydir, int R, float radius, hipStream_t stream);
int launch_sample_dirs(int size, float* light_dirs, const float* normal, const float* cdf_the, const float* cdf_phi, const float* alpha, int n_lights, int n_thes, int n_phis, float eps, hipStream_t stream);
int launch_math_expf(int n, float* y, const float* x, int sigmoid, hipStream_t stream);
int launch_squareplus(int n, bool bwd, float* out, const float* dy, const float* x, float b, bool accum, hipStream_t stream);

}  // namespace ndjir
