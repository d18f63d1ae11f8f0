// atomics_shape.hip -- does the fp32 atomic rate depend on how a wave's lanes are laid over the cells?  n random cells of D
// consecutive floats in a 2 GiB buffer receive one atomic add per float.
//   per-lane cells: lane = cell, D instructions (every instruction touches 64 different cells)      -- k_scatter_agg's flush until now
//   shared cells:   D consecutive lanes = the D floats of one cell, 1 instruction per 64 / D cells
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomics_shape.hip -o atomics_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int D>
__global__ void k_lane_cells(float* buf, const unsigned* idx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* p = buf + (size_t)idx[i] * D;
#pragma unroll
  for (int q = 0; q < D; ++q) atomicAdd(p + q, 1.0f);
}

template <int D>
__global__ void k_shared_cells(float* buf, const unsigned* idx, long long n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = t / D;
  if (i >= n) return;
  atomicAdd(buf + (size_t)idx[i] * D + (int)(t % D), 1.0f);
}

template <int D>
void run(float* buf, const unsigned* idx, long long n, long long cells) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_lane_cells<D>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, buf, idx, n);
      else hipLaunchKernelGGL(k_shared_cells<D>, dim3((unsigned)((n * D + 255) / 256)), dim3(256), 0, 0, buf, idx, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("D = %d, %s: %.1f us for %lld atomics = %.1f G atomics/s = %.2f G cells/s\n", D,
           mode == 0 ? "lane = cell          " : "D lanes share a cell ", ms * 1e3, n * D, n * D / (ms * 1e-3) / 1e9, n / (ms * 1e-3) / 1e9);
  }
}

int main() {
  const long long floats = 1LL << 29, n = 1LL << 22;
  float* buf; unsigned* idx;
  (void)hipMalloc(&buf, floats * 4); (void)hipMemset(buf, 0, floats * 4);
  unsigned* h = (unsigned*)malloc(n * 4);
  srand(412);
  for (int D : {1, 2, 4, 8, 16, 32, 64}) {
    const long long cells = floats / D;
    for (long long i = 0; i < n; ++i) h[i] = ((unsigned)rand() * 32768u + (unsigned)rand()) & (unsigned)(cells - 1);
    (void)hipMalloc(&idx, n * 4); (void)hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice);
    if (D == 1) run<1>(buf, idx, n, cells); else if (D == 2) run<2>(buf, idx, n, cells); else if (D == 64) run<64>(buf, idx, n, cells); else if (D == 4) run<4>(buf, idx, n, cells); else if (D == 8) run<8>(buf, idx, n, cells); else if (D == 16) run<16>(buf, idx, n, cells); else run<32>(buf, idx, n, cells);
    (void)hipFree(idx);
  }
  return 0;
}
