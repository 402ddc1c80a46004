// Probe: issue rate of v_fma_f32 vs v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (with op_sel swizzles) on gfx950,
// at 1, 2 and 4 waves per SIMD.  Prints SIMD cycles per instruction (wave64), from s_memtime-free wall time and the
// clock the same launch measures with a known s_sleep-free loop (reported both at 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  f32x2 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { v[j].x = threadIdx.x * 1e-3f + j; v[j].y = threadIdx.x * 2e-3f - j; }
  f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (MODE == 0) {        // 2 scalar FMAs
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j].x) : "v"(m.x), "v"(c.x));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j].y) : "v"(m.y), "v"(c.y));
        } else if (MODE == 1) { // 1 packed FMA
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        } else if (MODE == 2) { // packed add with swizzle + negate (x + i*y style)
          asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(v[j]) : "v"(c));
        } else if (MODE == 3) { // packed mul, broadcast low half of src1
          asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(v[j]) : "v"(m));
        } else if (MODE == 4) { // scalar add x2
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j].x) : "v"(c.x));
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j].y) : "v"(c.y));
        } else if (MODE == 5) { // v_mov_b32 x2 (does a plain move cost a full slot?)
          asm volatile("v_mov_b32 %0, %1" : "=v"(v[j].x) : "v"(v[(j + 1) & 15].y));
          asm volatile("v_mov_b32 %0, %1" : "=v"(v[j].y) : "v"(v[(j + 3) & 15].x));
        } else if (MODE == 6) { // v_pk_mov_b32
          asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(v[j]) : "v"(v[(j + 1) & 15]), "v"(v[(j + 3) & 15]));
        } else if (MODE == 7) { // DPP mov x2
          asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[j].x) : "v"(v[(j + 1) & 15].y));
          asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v[j].y) : "v"(v[(j + 3) & 15].x));
        }
      }
  }
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) r += v[j].x + v[j].y;
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MODE>
void run(const char* name, int per_iter, float* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * per_iter * (threads / 256);
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f cycles per instruction per SIMD @2.4 GHz\n", name, threads / 256, ms,
           ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  run<0>("v_fma_f32 (x2 per pair)", 128, d);
  run<1>("v_pk_fma_f32", 64, d);
  run<2>("v_pk_add_f32 op_sel+neg", 64, d);
  run<3>("v_pk_mul_f32 bcast", 64, d);
  run<4>("v_add/v_sub_f32 (x2)", 128, d);
  run<5>("v_mov_b32 (x2)", 128, d);
  run<6>("v_pk_mov_b32", 64, d);
  run<7>("v_mov_b32_dpp (x2)", 128, d);
  return 0;
}
