// probe: VALU throughput of fp32 ops whose operands are ALL VGPRs (valu_rate.hip used an SGPR and a
// literal): add a+b, fma a*b+c, and a radix-2 butterfly pattern, at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, int iters, float s) {
  float a[16], b[16], c[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.0f + 1e-6f * (threadIdx.x + i); c[i] = 1e-3f * i; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 1 VGPR source
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = a[i] + s;
    } else if (MODE == 1) {   // 2 VGPR sources
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = a[i] + b[i];
    } else if (MODE == 2) {   // 3 VGPR sources
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], b[i], c[i]);
    } else {                  // butterflies: (a, b) <- (a + b, a - b), scaled to stay finite: 2 instr per pair + 2 mul
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float u = a[i] + b[i], v = a[i] - b[i];
          a[i] = u; b[i] = v;
        }
    }
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { a[i] *= 0.5f; b[i] *= 0.5f; }   // 32 more (1 VGPR source)
    }
  }
  float acc = 0;
  for (int i = 0; i < 16; ++i) acc += a[i] + b[i] + c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* o; (void)hipMalloc(&o, 256 * 1024 * 4 * 4);
  const int iters = 20000;
  dim3 grid(256), block(256 * waves_per_simd);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(o, 10, 1.0001f);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, block>>>(o, iters, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr = (MODE == 3 ? 96.0 : 64.0) * iters;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-12s waves/SIMD %d: %.3f ms  cycles per VALU instr per SIMD (at 2.4 GHz): %.2f\n", name, waves_per_simd, ms,
         cycles / (instr * waves_per_simd));
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("add v,s", w); run<1>("add v,v", w); run<2>("fma v,v,v", w); run<3>("butterfly", w); }
  return 0;
}
