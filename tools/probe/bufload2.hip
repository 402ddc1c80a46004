#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v2i __attribute__((__vector_size__(8)));
__global__ void k(const float* base, int n, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n * 4, 0x00020000);
  int lane = threadIdx.x;
  v2i d = __builtin_amdgcn_raw_buffer_load_b64(r, (n - 8 + lane) * 4, 0, 0);   // pairs (n-8+lane, n-7+lane)
  out[2 * lane] = __builtin_bit_cast(float, d[0]);
  out[2 * lane + 1] = __builtin_bit_cast(float, d[1]);
  v2i e = __builtin_amdgcn_raw_buffer_load_b64(r, (lane - 4) * 8, 0, 0);       // pairs from index 2*(lane-4)
  out[128 + 2 * lane] = __builtin_bit_cast(float, e[0]);
  out[128 + 2 * lane + 1] = __builtin_bit_cast(float, e[1]);
}
int main() {
  const int n = 101, pad = 64;
  float h[pad + n + pad];
  for (int i = 0; i < pad + n + pad; ++i) h[i] = 1000.0f + i - pad;
  float *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 256 * 4);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d + pad, n, o);
  float r[256]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("END:"); for (int l = 0; l < 12; ++l) printf(" (%g,%g)", r[2 * l], r[2 * l + 1]); printf("\n");
  printf("BEG:"); for (int l = 0; l < 8; ++l) printf(" (%g,%g)", r[128 + 2 * l], r[128 + 2 * l + 1]); printf("\n");
  return 0;
}
