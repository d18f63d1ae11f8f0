// coissue.hip -- do the matrix pipe and the vector ALU of one SIMD run at the same time when the work comes from DIFFERENT
// waves?  One workgroup of 8 waves per CU (wave w -> SIMD w % 4): role A waves run a chain of MFMA 32x32x16 f16 over four
// independent accumulators, role B waves run independent v_fma_f32 chains (or ds_read_b128 / ds_write_b64 traffic).
// Modes: all-MFMA, all-VALU, 4 MFMA + 4 idle, 4 VALU + 4 idle, 4 MFMA + 4 VALU (one of each per SIMD).  If the mixed time is
// max(the two solo times) the pipes overlap; if it is their sum they do not.
// build: hipcc -O3 --offload-arch=gfx950 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_work(int iters, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

__device__ __forceinline__ void valu_work(int iters, float* out) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
  const float m = 1.0001f, c = 0.0003f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

__device__ __forceinline__ void lds_work(int iters, float* out, f32x4* lds) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      f32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lane * 16), "n"(0));
      asm volatile("s_waitcnt lgkmcnt(0)");
      acc += v;
    }
  }
  if (acc[0] == 123.456f) out[threadIdx.x] = acc[0] + lds[0][0];
}

// MFMA chain with the wave idling NOPS x 16 cycles after each MFMA (does a wave that WAITS on the busy matrix pipe hold the
// SIMD's issue port against the other waves?)
template <int NOPS>
__device__ __forceinline__ void mfma_nop_work(int iters, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
#define NOPQ for (int i = 0; i < NOPS; ++i) asm volatile("s_nop 15");
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
    NOPQ
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
    NOPQ
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
#pragma unroll
    NOPQ
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
#pragma unroll
    NOPQ
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// one wave doing both: after every MFMA, `NV` independent FMAs (NV = 8: the same 4 MFMAs + 32 FMAs per iteration as above)
template <int NV>
__device__ __forceinline__ void both_work(int iters, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
  const float m = 1.0001f, c = 0.0003f;
  for (int it = 0; it < iters; ++it) {
#define VAL for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 7]) : "v"(m), "v"(c));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
    VAL
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
    VAL
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
#pragma unroll
    VAL
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
#pragma unroll
    VAL
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// mode bits: role of waves 0-3 (low nibble) and of waves 4-7 (high nibble): 0 idle, 1 MFMA, 2 VALU, 3 LDS reads
__global__ void __launch_bounds__(512) k_co(int mode, int it_mfma, int it_valu, float* out) {
  __shared__ f32x4 lds[1024];
  if (threadIdx.x < 1024) lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? (mode & 15) : ((mode >> 4) & 15);
  const int prio = mode >> 8;                    // 1: MFMA waves high, 2: the other role high
  if (prio == 1 && role == 1) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && role != 1) __builtin_amdgcn_s_setprio(3);
  if (role == 1) mfma_work(it_mfma, out);
  else if (role == 2) valu_work(it_valu, out);
  else if (role == 3) lds_work(it_valu, out, lds);
  else if (role == 7) mfma_nop_work<1>(it_mfma, out);
  else if (role == 8) mfma_nop_work<2>(it_mfma, out);
  else if (role == 4) both_work<8>(it_mfma, out);
  else if (role == 5) both_work<4>(it_mfma, out);
  else if (role == 6) both_work<6>(it_mfma, out);
}

int main() {
  float* out; (void)hipMalloc(&out, 1 << 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int it_m = 20000, it_v = 20000;      // 4 MFMAs (128 matrix cycles) / 32 FMAs (128 issue cycles) / 8 LDS reads per iteration
  struct { int mode; const char* name; } runs[] = {
      {0x01, "4 waves MFMA (one per SIMD)"},       {0x11, "8 waves MFMA (two per SIMD)"},
      {0x02, "4 waves VALU"},                      {0x22, "8 waves VALU"},
      {0x21, "4 MFMA + 4 VALU (one of each per SIMD)"},
      {0x121, "4 MFMA (prio 3) + 4 VALU"},  {0x221, "4 MFMA + 4 VALU (prio 3)"},
      {0x07, "4 waves MFMA + s_nop 15"},   {0x27, "4 (MFMA + s_nop 15) + 4 VALU"},  {0x37, "4 (MFMA + s_nop 15) + 4 LDS read"},
      {0x08, "4 waves MFMA + 2 x s_nop 15"},   {0x28, "4 (MFMA + 2 x s_nop 15) + 4 VALU"},  {0x38, "4 (MFMA + 2 x s_nop 15) + 4 LDS read"},
      {0x04, "4 waves, each MFMA + 8 FMAs interleaved"},   {0x44, "8 waves, each MFMA + 8 FMAs interleaved"},
      {0x05, "4 waves, each MFMA + 4 FMAs interleaved"},   {0x55, "8 waves, each MFMA + 4 FMAs interleaved"},
      {0x06, "4 waves, each MFMA + 6 FMAs interleaved"},   {0x66, "8 waves, each MFMA + 6 FMAs interleaved"},
      {0x03, "4 waves LDS read"},                  {0x31, "4 MFMA + 4 LDS read"},  {0x32, "4 VALU + 4 LDS read"},
      {0x131, "4 MFMA (prio 3) + 4 LDS read"},     {0x231, "4 MFMA + 4 LDS read (prio 3)"}};
  for (auto& r : runs) {
    hipLaunchKernelGGL(k_co, dim3(256), dim3(512), 0, 0, r.mode, 100, 100, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_co, dim3(256), dim3(512), 0, 0, r.mode, it_m, it_v, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us\n", r.name, ms * 1e3);
  }
  return 0;
}
