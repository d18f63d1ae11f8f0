// shadow.hip -- what fits in the shadow of a 32x32x16 f16 MFMA when it comes from the SAME wave: one wave per SIMD issues
// MFMA, then N copies of one instruction kind, repeated; prints the time per MFMA (ns) for N = 0, 2, 4, 6, 8 per kind.
// build: hipcc -O3 --offload-arch=gfx950 shadow.hip -o shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OPS(N, STR, ...) for (int i_ = 0; i_ < N; ++i_) asm volatile(STR __VA_ARGS__);

template <int KIND, int N>
__global__ void __launch_bounds__(256) k(int iters, float* out) {
  __shared__ f32x4 lds[512];
  lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x[8]; unsigned hx[8]; f32x2 p[4]; f32x4 lv[8] = {};
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.01f + i; hx[i] = 0x3c003c00u + i; }
  for (int i = 0; i < 4; ++i) p[i] = f32x2{1.f + i, 2.f};
  const float m = 1.0001f, c = 0.0003f;
  const int la = (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#define SIDE                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < N; ++i) {                                                              \
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(m), "v"(c));                 \
      if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hx[i & 7]) : "v"(x[i & 7]), "v"(x[(i + 1) & 7])); \
      if (KIND == 2) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i & 7]) : "v"(hx[i & 7]));                     \
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));             \
      if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(x[i & 7]) : "v"(x[(i + 1) & 7]), "v"(m), "v"(hx[i & 7])); \
      if (KIND == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(hx[i & 7]) : "v"(x[i & 7]), "v"(m));    \
      if (KIND == 6) asm volatile("ds_write_b64 %0, %1" : : "v"(la / 2), "v"(p[i & 3]) : "memory");              \
      if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[i & 7]) : "v"(la) : "memory");                 \
      if (KIND == 10) asm volatile("ds_write_b128 %0, %1" : : "v"(la), "v"(lv[i & 7]) : "memory");              \
      if (KIND == 11) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:4" : : "v"(la / 2), "v"(p[i & 3]), "v"(p[(i + 1) & 3]) : "memory"); \
      if (KIND == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(c));                             \
      if (KIND == 9) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[i & 7]) : "v"(hx[i & 7])); \
    }
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
    SIDE
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
    SIDE
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
    SIDE
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
    SIDE
    if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 8; ++i) s += x[i] + (float)hx[i] + lv[i][0];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND, int N>
static float run(float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, 100, out);
  (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, iters, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6f / (iters * 4);
}
#define ROW(KIND, NAME) printf("%-22s %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", NAME, run<KIND, 0>(out, e0, e1), run<KIND, 2>(out, e0, e1), run<KIND, 4>(out, e0, e1), run<KIND, 6>(out, e0, e1), run<KIND, 8>(out, e0, e1), run<KIND, 12>(out, e0, e1));
int main() {
  float* out; (void)hipMalloc(&out, 1 << 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("ns per MFMA with N instructions behind each (one wave per SIMD)\n%-22s %6s %6s %6s %6s %6s %6s\n", "kind", "N=0", "2", "4", "6", "8", "12");
  ROW(0, "v_fma_f32") ROW(8, "v_sub_f32") ROW(1, "v_cvt_pk_f16_f32") ROW(2, "v_cvt_f32_f16") ROW(9, "v_cvt_f32_f16 sdwa") ROW(3, "v_pk_mul_f32")
  ROW(4, "v_fma_mix_f32") ROW(5, "v_fma_mixlo_f16") ROW(6, "ds_write_b64") ROW(10, "ds_write_b128") ROW(11, "ds_write2st64_b64") ROW(7, "ds_read_b128")
  return 0;
}
