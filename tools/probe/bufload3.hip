// probe: 16 x buffer_load_dwordx2 at immediate offsets 512 n1 from one per-lane byte offset (the wpf kernel's frame loads)
// against plain global loads of the same pairs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, int n, int base0, float* out, float* ref) {
  const int ll = threadIdx.x & 63;
  const int base = base0 + 2 * ll;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, n * 4, 0x00020000);
  const int vo = 4 * base;
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo + 512 * n1, 0, 0);
    out[(n1 * 64 + ll) * 2] = __uint_as_float(v.x);
    out[(n1 * 64 + ll) * 2 + 1] = __uint_as_float(v.y);
    ref[(n1 * 64 + ll) * 2] = a[base + 128 * n1];
    ref[(n1 * 64 + ll) * 2 + 1] = a[base + 128 * n1 + 1];
  }
}
int main() {
  const int n = 24000;
  float* h = new float[n];
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o, *r;
  hipMalloc(&d, n * 4); hipMalloc(&o, 2048 * 4); hipMalloc(&r, 2048 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  for (int base0 : {0, 480 - 1024 + 2048, 4800, 21000}) {
    k<<<1, 64>>>(d, n, base0, o, r);
    float ho[2048], hr[2048];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost); hipMemcpy(hr, r, sizeof(hr), hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < 2048; ++i) if (ho[i] != hr[i]) { if (first < 0) first = i; ++bad; }
    printf("base0 %d: %d mismatches", base0, bad);
    if (first >= 0) printf(" first at %d: buf %g ref %g", first, ho[first], hr[first]);
    printf("\n");
  }
  return 0;
}
