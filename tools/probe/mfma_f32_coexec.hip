// Probe: do v_mfma_f32_16x16x4_f32 and f32 VALU FMAs overlap on one SIMD?  512-thread blocks (2 waves per
// SIMD): waves 0-3 issue MFMAs (mode & 1), waves 4-7 issue independent v_fma_f32 (mode & 2).
// Prints ms for mode 1 (f32 MFMA only), 2 (VALU only), 3 (both), 4 (bf16 16x16x32 MFMA only), 6 (bf16 MFMA + VALU).  If t3 ~ max(t1, t2) the pipes overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 4) {     // bf16 16x16x32
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      bf16x8 x, y;
      for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + e * 0.25f); }
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (mode & 1) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else if (mode & 2) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float m = 1.0001f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], m, c);   // 64 independent-ish FMAs per iteration
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) r += v[j];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode : {1, 2, 3, 4, 6}) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 MFMAs per iteration (mode & 1), 64 v_fma per iteration (mode & 2)
    printf("mode %d: %.3f ms  (%.1f ns/iter: MFMA %.1f cyc each @2.4GHz if alone, VALU %.2f cyc each)\n", mode, ms,
           ms * 1e6 / iters, ms * 1e6 / iters * 2.4 / 4, ms * 1e6 / iters * 2.4 / 64);
  }
  return 0;
}
