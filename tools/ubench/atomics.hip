// fp32 atomic-add rate on MI355X by memory scope: device (agent) scope resolves behind the non-coherent per-XCD L2s,
// workgroup scope runs in the issuing XCD's L2.  n random float4-cell updates (4 atomics each) into a 2 GiB buffer.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench/atomics.hip -o tools/ubench/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCOPE>
__global__ void k(float* buf, const unsigned* idx, long long n, int xcd_local) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned c = idx[i];
  if (xcd_local) {   // keep a cell on the XCD that owns it (blockIdx % 8 round-robin): cell id low bits := block's XCD
    c = (c & ~7u) | (blockIdx.x & 7u);
  }
  float* p = buf + (size_t)c * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (SCOPE == 0) atomicAdd(p + q, 1.0f);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p + q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p + q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void kp(float* buf, const unsigned* idx, long long n, long long cells) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float* p = buf + ((size_t)(xcc & 7u) * cells + idx[i]) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) __hip_atomic_fetch_add(p + q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

int main() {
  const long long cells = 1LL << 27, n = 1LL << 22;
  float* buf; unsigned* idx;
  hipMalloc(&buf, cells * 16); hipMemset(buf, 0, cells * 16);
  unsigned* h = (unsigned*)malloc(n * 4);
  srand(412);
  for (long long i = 0; i < n; ++i) h[i] = ((unsigned)rand() * 32768u + (unsigned)rand()) & (cells - 1);
  hipMalloc(&idx, n * 4); hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int local = 0; local < 2; ++local)
    for (int s = 0; s < 3; ++s) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (s == 0) k<0><<<(n + 255) / 256, 256>>>(buf, idx, n, local);
        else if (s == 1) k<1><<<(n + 255) / 256, 256>>>(buf, idx, n, local);
        else k<2><<<(n + 255) / 256, 256>>>(buf, idx, n, local);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("xcd_local=%d scope=%s: %.1f us for %lld atomics = %.1f G atomics/s\n", local,
             s == 0 ? "atomicAdd(default)" : s == 1 ? "workgroup" : "agent", ms * 1e3, n * 4, n * 4 / (ms * 1e-3) / 1e9);
    }
  // small tables (a hash level: 2^19 entries x 2 floats = 4 MB): do L2-resident targets change the picture?
  // XCD-private copies: block b works on copy (b & 7), so no cell is shared between XCDs if blocks are dealt round-robin
  for (int lg : {20, 18, 16, 14}) {
    unsigned* idx2; hipMalloc(&idx2, n * 4);
    for (long long i = 0; i < n; ++i) h[i] &= ((1u << lg) - 1);
    hipMemcpy(idx2, h, n * 4, hipMemcpyHostToDevice);
    for (int s = 0; s < 3; ++s) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (s == 0) k<0><<<(n + 255) / 256, 256>>>(buf, idx2, n, 0);
        else if (s == 1) k<1><<<(n + 255) / 256, 256>>>(buf, idx2, n, 0);
        else kp<<<(n + 255) / 256, 256>>>(buf, idx2, n, 1LL << lg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("table 2^%d cells (%lld KB) %s: %.1f us = %.1f G atomics/s\n", lg, (16LL << lg) >> 10,
             s == 0 ? "device scope, one table" : s == 1 ? "workgroup scope, one table" : "workgroup scope, 8 XCD-private copies",
             ms * 1e3, n * 4 / (ms * 1e-3) / 1e9);
    }
    hipFree(idx2);
  }
  // correctness of the XCD-local workgroup-scope variant: total must equal the number of atomics issued
  hipMemset(buf, 0, cells * 16);
  k<1><<<(n + 255) / 256, 256>>>(buf, idx, n, 1);
  hipDeviceSynchronize();
  float* hb = (float*)malloc(cells * 16);
  hipMemcpy(hb, buf, cells * 16, hipMemcpyDeviceToHost);
  double tot = 0; for (long long i = 0; i < cells * 4; ++i) tot += hb[i];
  printf("workgroup-scope, xcd-local: sum = %.0f, expected %lld\n", tot, n * 4);
  hipMemset(buf, 0, cells * 16);
  k<1><<<(n + 255) / 256, 256>>>(buf, idx, n, 0);
  hipDeviceSynchronize();
  hipMemcpy(hb, buf, cells * 16, hipMemcpyDeviceToHost);
  tot = 0; for (long long i = 0; i < cells * 4; ++i) tot += hb[i];
  printf("workgroup-scope, NOT xcd-local (cells shared between XCDs): sum = %.0f, expected %lld\n", tot, n * 4);
  return 0;
}
