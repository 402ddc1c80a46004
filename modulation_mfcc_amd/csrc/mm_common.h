// Shared by every translation unit of libmodmfcc.so: includes, the error string, HIP_TRY, per-device attribute flags,
// small device helpers.  gfx950 (MI355X / CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <new>

#include "mm_internal.h"
#include "mm_dev.h"

#define MM_TW_N 8192  // master twiddle table: exp(-2 pi i k / 8192), k < 8192 (full circle)

extern thread_local std::string g_hip_err;      // defined in mm_api.hip; mm_last_hip_error() returns it

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) {                                                   \
      g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);          \
      return MM_ERR_HIP;                                                      \
    }                                                                         \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the device function object, i.e. it is kept PER DEVICE: the
// entry points that have no plan to set it in (mm_mfcc_change_f64, mm_sosfiltfilt_*, mm_resample_banded_f32 take any
// stream of any device) set it once per device, behind a flag indexed by hipGetDevice() -- atomic, so that two host
// threads making their first calls at once are fine (both may set the attribute: idempotent).
#define MM_MAX_DEV 64
struct PerDeviceOnce { std::atomic<int> done[MM_MAX_DEV]; };
template <class F>
static int per_device_once(PerDeviceOnce& o, const char* what, F set) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { g_hip_err = "hipGetDevice failed"; return MM_ERR_HIP; }
  const bool flagged = dev >= 0 && dev < MM_MAX_DEV;
  if (flagged && o.done[dev].load(std::memory_order_acquire)) return MM_OK;
  if (!set()) { g_hip_err = std::string("hipFuncSetAttribute(") + what + ") failed"; return MM_ERR_HIP; }
  if (flagged) o.done[dev].store(1, std::memory_order_release);
  return MM_OK;
}

// sample i of a clip, zero outside [0, n) (the centred frames are zero-padded), with optional pre-emphasis
__device__ __forceinline__ float load_sample(const float* __restrict__ a, int64_t i, int64_t n,
                                             float pre) {
  if (i < 0 || i >= n) return 0.0f;
  float v = a[i];
  if (pre != 0.0f && i > 0) v -= __fmul_rn(pre, a[i - 1]);   // rounded product, then subtract: no fma
  return v;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of ONE wave execute in order; this only stops the compiler from moving
  // accesses across the point where lanes exchange data through the wave-private buffer.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that waits for LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every
// global load / store / atomic in flight: in the persistent fused kernels that exposes a full
// memory round trip per tile (the next tile's prefetch before phase B, the log-mel stores after it).
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// order-preserving float -> int key for atomicMax
__device__ __forceinline__ int float_key(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key_float(int k) {
  return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

typedef float mm_f32x4 __attribute__((ext_vector_type(4)));   // accumulator of v_mfma_f32_16x16x4_f32

// In-place radix-2 DIT over a wave-private LDS buffer; input already bit-reversed.
__device__ __forceinline__ void wave_cfft_lds(float2* z, int log2nc, const float2* __restrict__ tw,
                                              int lane) {
  const int nc = 1 << log2nc;
  for (int s = 1; s <= log2nc; ++s) {
    const int half = 1 << (s - 1);
    const int tw_stride = MM_TW_N >> s;
    for (int j = lane; j < (nc >> 1); j += 64) {
      const int pos = j & (half - 1);
      const int i0 = ((j >> (s - 1)) << s) + pos;
      const int i1 = i0 + half;
      const float2 w = tw[pos * tw_stride];
      const float2 a = z[i0];
      const float2 t = cmul(w, z[i1]);
      z[i0] = make_float2(a.x + t.x, a.y + t.y);
      z[i1] = make_float2(a.x - t.x, a.y - t.y);
    }
    wave_lds_sync();
  }
}

// Split the half-length complex FFT Z (nc points, in LDS) of a packed real row into the real
// FFT bins k and nc-k.  Returns X[k] in xa and X[nc-k] in xb.
__device__ __forceinline__ void real_split(const float2* z, int k, int nc, const float2* __restrict__ tw,
                                           int tw_stride, float2& xa, float2& xb) {
  const float2 a = z[k];
  const float2 b = z[(nc - k) & (nc - 1)];
  const float2 E = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
  const float2 D = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y + b.y));
  const float2 O = make_float2(D.y, -D.x);  // -i * D
  const float2 t = cmul(tw[k * tw_stride], O);
  xa = make_float2(E.x + t.x, E.y + t.y);
  xb = make_float2(E.x - t.x, -(E.y - t.y));
}

// native vector types and explicit address spaces (1 = global, 3 = LDS) for code that must not fall back to flat accesses
typedef float mm_v2f __attribute__((ext_vector_type(2)));
typedef float mm_v4f __attribute__((ext_vector_type(4)));
#define MM_GLOBAL __attribute__((address_space(1)))
#define MM_LDS __attribute__((address_space(3)))

typedef unsigned mm_v2u __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) MmFloat4U { float x, y, z, w; };

// Global memory through buffer descriptors (wave-uniform base in SGPRs, 32-bit per-lane byte offset, immediate
// offsets folded into the instruction, reads past num_records return 0): the 64-bit address pairs of ~60 plain
// loads do not fit beside the transform in the 128 VGPRs of a 16-wave workgroup.
__device__ __forceinline__ float mm_buf_f32(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ float2 mm_buf_f32x2(__amdgpu_buffer_rsrc_t r, int byte_off) {
  const mm_v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}

// fmaxf / fminf as ONE instruction.  The compiler quiets possible signalling NaNs first (a v_max_f32 x, x "canonicalise" in
// front of every operand whose origin it cannot see: three extra instructions per max / min pair in the mel walk); the
// hardware instruction already implements IEEE maxNum / minNum (a NaN operand loses, sNaN is quieted), i.e. fmaxf's result.
__device__ __forceinline__ float mm_max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// in-place forms (the accumulator is a tied operand: no copies where the update sits in one arm of a branch)
__device__ __forceinline__ void mm_max_acc(float& acc, float b) { asm("v_max_f32 %0, %0, %1" : "+v"(acc) : "v"(b)); }
__device__ __forceinline__ void mm_min_acc(float& acc, float b) { asm("v_min_f32 %0, %0, %1" : "+v"(acc) : "v"(b)); }
__device__ __forceinline__ float mm_max_raw_s(float a, float s_uniform) {      // second operand wave-uniform (an SGPR)
  float r;
  asm("v_max_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(s_uniform));
  return r;
}
__device__ __forceinline__ float mm_min_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The value a filter without weights takes in the fused kernels' phase B: 10 log10(max(amin, 0)) - db_offset exactly as
// the kernels evaluate it (one fused multiply-add of the hardware log2).  An explicit fma: written as a product and a
// difference the compiler contracts it into whatever consumes it (thr - L0 became fma(-c, log, thr): one ulp off).
__device__ __forceinline__ float mm_empty_level(float amin, float db_offset) {
  return __builtin_fmaf(3.0102999566398120f, __builtin_amdgcn_logf(amin), -db_offset);
}

#define MM_LM_LDS_MAX 163840     // the CU's whole LDS: what a single workgroup may declare (160 KiB)
#define MM_DCT_KB 16            // DCT coefficients per accumulator block of the lane <-> frame clamp + DCT kernels

// Clamp fix-up of the fused DCT (dct_fixup_kernel in tile mode, s16_fix_clip in clip mode -- the same arithmetic in the
// same order, so the two modes agree to the bit).  The fused kernel stored DCT(unclamped rows); the clamped result is
//   sum_m D[k][m] max(x_m, thr) = DCT(unclamped)[k] + sum_m D[k][m] max(thr - x_m, 0):
// the correction of ONE frame's coefficients k0 .. k0 + 15 over the stored filters, ascending, into acc.  Only values under
// the threshold contribute, so a wave whose 64 frames have none for a filter skips its sixteen multiply-adds (adding
// D * 0 would leave acc as it is: the skip never changes a result) -- a clip with a few dips costs one pass over its
// log-mel rows instead of a full n_mels x n_mfcc product per frame.  x_of(m): the frame's log-mel value of filter m;
// ew_of(m0): the "is empty, not stored" bits of filters m0 .. m0 + 31 (m0 a multiple of 32); d_of(m, f): calls f(kk, D[k0 + kk][m]), kk < 16.
// Returns (wave-uniform) whether anything was added.
template <typename FX, typename FE, typename FD>
__device__ __forceinline__ bool mm_clamp_corr(float (&acc)[MM_DCT_KB], int n_mels, float thr, FX x_of, FE ew_of, FD d_of) {
  bool touched = false;
#pragma unroll 1
  for (int m0 = 0; m0 < n_mels; m0 += 32) {
    // thirty-two loads in flight (the rows come from HBM: a clip's fix-up is a chain of such round trips); the rows of
    // empty filters were never stored and are not read
    const unsigned ew = ew_of(m0);
    float xv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      xv[j] = 0.0f;
      if (m0 + j < n_mels && !((ew >> j) & 1u)) xv[j] = x_of(m0 + j);             // (wave-uniform)
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (m0 + j < n_mels && !((ew >> j) & 1u)) {                  // (wave-uniform)
        const float d = fmaxf(thr - xv[j], 0.0f);
        if (__ballot(d > 0.0f) != 0ull) {
          touched = true;
          d_of(m0 + j, [&](int kk, float dk) { acc[kk] = __builtin_fmaf(dk, d, acc[kk]); });
        }
      }
    }
  }
  return touched;
}

struct RfftParams {
  const float* in;
  int64_t rows, in_len, in_stride;
  int n, log2nc, rows_per_wave;
  const float2* tw;
  float* out;  // complex64 [rows][n/2+1]
};

namespace {

int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <class T>
int upload(T** dst, const void* src, size_t bytes) {
  HIP_TRY(hipMalloc((void**)dst, bytes));
  HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return MM_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace
