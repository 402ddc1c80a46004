// probe: VALU throughput of scalar vs packed fp32 FMA / ADD on gfx950 at 1, 2, 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float s) {
  float a[16];
  v2f p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
  v2f sv = {s, s * 1.0001f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], s, 0.5f);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], sv, (v2f){0.5f, 0.25f});
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = a[i] + s;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = p[i] + sv;
    }
  }
  float acc = 0;
  for (int i = 0; i < 16; ++i) acc += a[i];
  for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* o; (void)hipMalloc(&o, 256 * 1024 * 4 * 4);
  const int iters = 20000;
  dim3 grid(256), block(256 * waves_per_simd);   // 1 block per CU, 4*w waves per CU
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(o, 10, 1.0001f);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, block>>>(o, iters, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop_instr = (MODE == 0 || MODE == 2) ? 64.0 * iters : 32.0 * iters;   // VALU instrs per wave
  const double lane_ops = 64.0 * iters;   // float ops per lane (fma or add counted as 1)
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-10s waves/SIMD %d: %.3f ms  cycles per VALU instr per SIMD: %.2f  (float-ops per clk per SIMD: %.1f)\n", name,
         waves_per_simd, ms, cycles / (flop_instr * waves_per_simd), lane_ops * 64 * waves_per_simd / cycles);
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("fma", w); run<1>("pk_fma", w); run<2>("add", w); run<3>("pk_add", w);
  }
  return 0;
}
