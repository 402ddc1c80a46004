// probe: DFT-16 + 15 twiddle multiplies per lane, written with scalar fp32 ops (as f16::dft16) vs with
// packed v2f ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers), at 1, 2, 4
// waves per SIMD.  Same flops; the packed form issues about half the instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ constexpr int P16(int k) { return 4 * (k & 3) + (k >> 2); }

// ---- scalar ----
__device__ __forceinline__ float2 cmulf(float2 a, float2 w) {
  return make_float2(fmaf(-a.y, w.y, a.x * w.x), fmaf(a.y, w.x, a.x * w.y));
}
__device__ __forceinline__ void radix4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 t0 = make_float2(v0.x + v2.x, v0.y + v2.y);
  const float2 t1 = make_float2(v0.x - v2.x, v0.y - v2.y);
  const float2 t2 = make_float2(v1.x + v3.x, v1.y + v3.y);
  const float2 t3 = make_float2(v1.x - v3.x, v1.y - v3.y);
  v0 = make_float2(t0.x + t2.x, t0.y + t2.y);
  v2 = make_float2(t0.x - t2.x, t0.y - t2.y);
  v1 = make_float2(t1.x + t3.y, t1.y - t3.x);
  v3 = make_float2(t1.x - t3.y, t1.y + t3.x);
}
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) radix4(x[b], x[4 + b], x[8 + b], x[12 + b]);
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  x[5] = cmulf(x[5], make_float2(C1, -S1));
  x[6] = make_float2((x[6].x + x[6].y) * H, (x[6].y - x[6].x) * H);
  x[7] = cmulf(x[7], make_float2(S1, -C1));
  x[9] = make_float2((x[9].x + x[9].y) * H, (x[9].y - x[9].x) * H);
  x[10] = make_float2(x[10].y, -x[10].x);
  x[11] = make_float2((x[11].y - x[11].x) * H, -(x[11].x + x[11].y) * H);
  x[13] = cmulf(x[13], make_float2(S1, -C1));
  x[14] = make_float2((x[14].y - x[14].x) * H, -(x[14].x + x[14].y) * H);
  x[15] = cmulf(x[15], make_float2(-C1, S1));
#pragma unroll
  for (int c = 0; c < 4; ++c) radix4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
}

// ---- packed ----
__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2f mi(v2f a) { return (v2f){a.y, -a.x}; }            // a * (-i)
__device__ __forceinline__ v2f pcmul(v2f a, v2f w) {
  // (a.x w.x - a.y w.y, a.x w.y + a.y w.x): mul then fma, as cmulf
  const v2f t = (v2f){a.x, a.x} * w;
  return __builtin_elementwise_fma((v2f){-a.y, a.y}, swp(w), t);
}
__device__ __forceinline__ void pradix4(v2f& v0, v2f& v1, v2f& v2, v2f& v3) {
  const v2f t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, t3 = v1 - v3;
  v0 = t0 + t2;
  v2 = t0 - t2;
  v1 = t1 + mi(t3);
  v3 = t1 - mi(t3);
}
__device__ __forceinline__ void pdft16(v2f (&x)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) pradix4(x[b], x[4 + b], x[8 + b], x[12 + b]);
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
  x[5] = pcmul(x[5], (v2f){C1, -S1});
  x[6] = (x[6] + mi(x[6])) * (v2f){H, H};
  x[7] = pcmul(x[7], (v2f){S1, -C1});
  x[9] = (x[9] + mi(x[9])) * (v2f){H, H};
  x[10] = mi(x[10]);
  x[11] = (mi(x[11]) - x[11]) * (v2f){H, H};
  x[13] = pcmul(x[13], (v2f){S1, -C1});
  x[14] = (mi(x[14]) - x[14]) * (v2f){H, H};
  x[15] = pcmul(x[15], (v2f){-C1, S1});
#pragma unroll
  for (int c = 0; c < 4; ++c) pradix4(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
}

template <int MODE>
__global__ void k(float* out, int iters) {
  const float s = 1.0f / 16.0f;
  float acc = 0.0f;
  if (MODE == 0) {
    float2 x[16], w[16];
    for (int i = 0; i < 16; ++i) { x[i] = make_float2(0.01f * (threadIdx.x + i), 0.5f - 0.01f * i); w[i] = make_float2(__cosf(0.1f * i * threadIdx.x), -__sinf(0.1f * i * threadIdx.x)); }
    for (int it = 0; it < iters; ++it) {
      dft16(x);
#pragma unroll
      for (int j = 1; j < 16; ++j) x[P16(j)] = cmulf(x[P16(j)], w[j]);
#pragma unroll
      for (int j = 0; j < 16; ++j) { x[j].x *= s; x[j].y *= s; }
    }
    for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
  } else {
    v2f x[16], w[16];
    for (int i = 0; i < 16; ++i) { x[i] = (v2f){0.01f * (threadIdx.x + i), 0.5f - 0.01f * i}; w[i] = (v2f){__cosf(0.1f * i * threadIdx.x), -__sinf(0.1f * i * threadIdx.x)}; }
    for (int it = 0; it < iters; ++it) {
      pdft16(x);
#pragma unroll
      for (int j = 1; j < 16; ++j) x[P16(j)] = pcmul(x[P16(j)], w[j]);
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] *= (v2f){s, s};
    }
    for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
float run(const char* name, int waves_per_simd, float* check) {
  float* o; (void)hipMalloc(&o, 256 * 1024 * 4);
  const int iters = 4000;
  dim3 grid(256), block(256 * waves_per_simd);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<grid, block>>>(o, 3);
  (void)hipMemcpy(check, o, 4 * 64, hipMemcpyDeviceToHost);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, block>>>(o, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-8s waves/SIMD %d: %.3f ms  -> %.1f cycles (2.4 GHz) per DFT-16+twiddle per SIMD\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd));
  (void)hipFree(o);
  return ms;
}
int main() {
  float a[64], b[64];
  for (int w : {1, 2, 4}) { run<0>("scalar", w, a); run<1>("packed", w, b); }
  int same = 1;
  for (int i = 0; i < 64; ++i) same &= (a[i] == b[i]);
  printf("bit-identical results: %s\n", same ? "yes" : "no");
  return 0;
}
