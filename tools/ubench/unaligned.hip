// unaligned.hip -- are 16-byte vector accesses at 4-byte alignment correct (and how fast) on this GPU?  mlp3w.hip uses them where a
// row starts at a 4- or 8-byte offset (the geometric net's output inside Z, 257-wide gradients).
// build: hipcc -O3 --offload-arch=gfx950 unaligned.hip -o unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));
__global__ void k_copy(const float* __restrict__ in, float* __restrict__ out, int off, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4u*>(in + off + 4 * i);
    *reinterpret_cast<f32x4u*>(out + off + 4 * i) = v * 2.f;
  }
}
int main() {
  const long long n4 = 1 << 24;               // 256 MB per array
  const size_t n = (size_t)n4 * 4 + 16;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000);
  float *in, *out;
  (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4);
  (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int off = 0; off < 4; ++off) {
    (void)hipMemset(out, 0, n * 4);
    hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, in, out, off, n4);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, in, out, off, n4);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> r(n);
    (void)hipMemcpy(r.data(), out, n * 4, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < (size_t)n4 * 4; ++i) if (r[off + i] != 2.f * h[off + i]) ++bad;
    printf("offset %d floats: %lld wrong of %lld, %.0f GB/s (%s)\n", off, bad, n4 * 4, 2.0 * n4 * 16 / (ms / 5 * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
