// probe: raw buffer load range checking on gfx950 (negative voffset + immediate offset, 8-byte
// loads straddling num_records)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* base, int n, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n * 4, 0x00020000);
  int lane = threadIdx.x;
  // case A: voffset negative, no immediate
  int offA = (lane - 8) * 4;
  float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, offA, 0, 0));
  // case B: voffset negative, immediate +64 bytes added via pointer math the compiler may fold
  int offB = (lane - 32) * 4;
  float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, offB + 64, 0, 0));
  // case C: soffset carries the negative part
  float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, -32, 0));
  // case D: 8-byte load straddling the end: elements n-1 (valid) and n (invalid)
  typedef int v2i __attribute__((ext_vector_type(2)));
  v2i d = __builtin_amdgcn_raw_buffer_load_b64(r, (n - 1 - (lane & 1)) * 4, 0, 0);
  out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
  out[192 + lane] = __builtin_bit_cast(float, d.x); out[256 + lane] = __builtin_bit_cast(float, d.y);
}
int main() {
  const int n = 101, pad = 64;
  float h[pad + n + pad];
  for (int i = 0; i < pad + n + pad; ++i) h[i] = 1000.0f + i - pad;   // value = 1000 + index
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 320 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d + pad, n, o);
  float r[320]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("A (idx lane-8):"); for (int i = 0; i < 12; ++i) printf(" %g", r[i]); printf("\n");
  printf("B (idx lane-32+16):"); for (int i = 12; i < 22; ++i) printf(" %g", r[64 + i]); printf("\n");
  printf("C (soffset -32B: idx lane-8):"); for (int i = 0; i < 12; ++i) printf(" %g", r[128 + i]); printf("\n");
  printf("D lane0 (idx n-1,n): %g %g ; lane1 (idx n-2,n-1): %g %g\n", r[192], r[256], r[193], r[257]);
  return 0;
}
