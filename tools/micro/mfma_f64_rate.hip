// dev micro-benchmark (GPU box): issue interval of v_mfma_f64_16x16x4_f64 per wave at 1..4 waves per SIMD, 1..4 independent
// accumulator tiles -- what the float64 weighted sums of the segmented IIR kernels would cost on the matrix pipe.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int ACC>
__global__ void mfma_loop(double* out, double a, double b, int iters) {
  v4d c[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) c[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  const double av = a + threadIdx.x * 1e-9, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ACC>
static void run(int threads) {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 1024);
  const int iters = 5000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_loop<ACC>), dim3(256), dim3(threads), 0, 0, d, 1.0, 1e-3, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_loop<ACC>), dim3(256), dim3(threads), 0, 0, d, 1.0, 1e-3, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wps = threads / 64.0 / 4.0;
  const double per_simd = (double)iters * ACC * wps;
  printf("f64 MFMA 16x16x4, %d accumulators, %.0f waves/SIMD: %.3f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s chip\n", ACC, wps, ms,
         ms * 1e6 / per_simd, 2048.0 * iters * ACC * (threads / 64.0) * 256 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int t : {256, 512, 1024}) { run<1>(t); run<2>(t); run<4>(t); }
  return 0;
}
