// probe: cost of ds_read_b128 / ds_read_b32 by address pattern on gfx950: all lanes one address
// (broadcast), 16 distinct 16-byte chunks replicated over the 4 rows of a wave (per-lane constant
// records of the fused kernels), 64 distinct conflict-free chunks; 1 and 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT, int WIDE>
__global__ void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base;
  if (PAT == 0) base = wave * 64;                                  // broadcast: one address per wave
  else if (PAT == 1) base = wave * 64 + (lane & 15) * 4 * 21;      // 16 chunks (pitch 84 floats), x4 rows
  else base = lane * 4 + wave * 256;                               // 64 distinct chunks, contiguous
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = (base + u * 4 + (it & 7) * 32) & 16380;
      if (WIDE) {
        const float4 v = *reinterpret_cast<const float4*>(&lds[idx]);
        acc += v.x + v.y + v.z + v.w;
      } else {
        acc += lds[idx];
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int PAT, int WIDE>
void run(const char* name, int waves) {
  float* o; (void)hipMalloc(&o, 256 * 1024 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<PAT, WIDE><<<256, 64 * waves>>>(o, 10);
  (void)hipEventRecord(e0);
  k<PAT, WIDE><<<256, 64 * waves>>>(o, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * iters * waves);
  printf("%-28s waves/CU %2d: %.3f ms  cycles of CU time per wave-instruction (at 2.4 GHz): %.1f\n", name, waves, ms, cyc);
  (void)hipFree(o);
}
int main() {
  for (int w : {1, 4, 16}) {
    run<0, 1>("b128 broadcast", w); run<1, 1>("b128 16 chunks x 4 rows", w); run<2, 1>("b128 64 distinct", w);
    run<0, 0>("b32 broadcast", w); run<2, 0>("b32 64 distinct", w);
  }
  return 0;
}
