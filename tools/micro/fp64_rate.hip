// dev micro-benchmark (GPU box): issue rate of dependent / independent float64 FMAs per wave, at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_rate.hip -o /tmp/fp64_rate && /tmp/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, typename T>
__global__ void chains(T* out, T a, T b, int iters) {
  T z[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) z[c] = (T)(threadIdx.x + c);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) z[c] = __builtin_fma(a, z[c], b);
  }
  T s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += z[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH, typename T>
static void run(const char* name, int threads) {
  T* d; hipMalloc(&d, sizeof(T) * 256 * 1024);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chains<CH, T>), dim3(256), dim3(threads), 0, 0, d, (T)0.999, (T)0.001, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chains<CH, T>), dim3(256), dim3(threads), 0, 0, d, (T)0.999, (T)0.001, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 64.0 / 4.0;
  const double ops_per_simd = (double)iters * CH * waves_per_simd;          // wave-instructions per SIMD
  printf("%s chains %2d, %4d threads/CU (%.2f waves/SIMD): %.3f ms, %.2f ns per wave-FMA per SIMD, %.1f TFLOP/s chip\n", name, CH,
         threads, waves_per_simd, ms, ms * 1e6 / ops_per_simd, 2.0 * iters * CH * threads * 256 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int t : {256, 512, 1024}) {
    run<1, double>("f64", t); run<2, double>("f64", t); run<4, double>("f64", t); run<8, double>("f64", t); run<16, double>("f64", t);
  }
  for (int t : {256, 1024}) { run<1, float>("f32", t); run<4, float>("f32", t); run<16, float>("f32", t); }
  return 0;
}
