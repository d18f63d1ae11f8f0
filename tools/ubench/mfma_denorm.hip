// mfma_denorm.hip -- does v_mfma_f32_32x32x16_f16 on gfx950 keep f16 denormal INPUTS, or flush them to zero?
// A = 2^-20 (an f16 denormal: smallest normal is 2^-14) everywhere, B = 2^10 everywhere: every output = 16 * 2^-10 = 2^-6 if
// denormals are kept, 0 if they are flushed.          build: hipcc -O3 --offload-arch=gfx950 mfma_denorm.hip -o mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = (_Float16)1024.f; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
}
int main() {
  float* d; (void)hipMalloc(&d, 256);
  float h[64];
  const unsigned short bits[3] = {0x0010 /* 2^-20 */, 0x0001 /* 2^-24 */, 0x0400 /* 2^-14, normal */};
  const char* nm[3] = {"2^-20 (denormal)", "2^-24 (smallest denormal)", "2^-14 (smallest normal)"};
  const double expect[3] = {16 * 1024.0 / 1048576.0, 16 * 1024.0 / 16777216.0, 16 * 1024.0 / 16384.0};
  for (int t = 0; t < 3; ++t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, bits[t]);
    (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("A = %-26s x B = 2^10, k = 16: got %.9g, exact %.9g\n", nm[t], h[0], expect[t]);
  }
  return 0;
}
