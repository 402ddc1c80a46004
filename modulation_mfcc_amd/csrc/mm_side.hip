// libmodmfcc: the rows either side of the path that need no MFCC plan -- RMS and Hilbert
// amplitude envelopes (script/calc.py:284-343), PCM decode and sample-rate conversion of what librosa.load reads
// (script/mfcc.py:284,373) -- and the bench's device-copy kernel.  gfx950 only.
#include "mm_common.h"

#include "mm_hilbert.hip.inc"
#include "mm_resample.hip.inc"

extern "C" {

// one wave per output frame: coalesced strided sum of squares, zero padding outside the clip
__global__ __launch_bounds__(256) void rms_frames_kernel(const float* __restrict__ audio, int64_t n_samples,
                                                         int64_t stride, int frame_length, int hop, int pad,
                                                         int64_t n_out, int64_t total, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= total) return;
  const int64_t b = w / n_out, t = w - b * n_out;
  const float* a = audio + b * stride;
  const int64_t start = t * hop - pad;
  float acc = 0.0f;
  for (int j = lane; j < frame_length; j += 64) {
    const int64_t i = start + j;
    const float v = (i >= 0 && i < n_samples) ? a[i] : 0.0f;
    acc = fmaf(v, v, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[w] = sqrtf(acc / (float)frame_length);
}

// Tiled variant: a workgroup squares the (F-1)*hop + frame_length samples of F consecutive frames of
// one clip into LDS once (frames overlap frame_length/hop times), then every wave sums whole frames
// from LDS in the same order as rms_frames_kernel (lane-strided, then a butterfly): same bits, each
// sample read from global memory ~once instead of frame_length/hop times.
__global__ __launch_bounds__(256) void rms_tile_kernel(const float* __restrict__ audio, int64_t n_samples,
                                                       int64_t stride, int frame_length, int hop, int pad,
                                                       int64_t n_out, int frames_per_tile, int64_t tiles_per_clip,
                                                       float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float rms_sq[];
  float* sq = rms_sq;
  const int64_t b = blockIdx.x / tiles_per_clip, tile = blockIdx.x - b * tiles_per_clip;
  const int64_t t0 = tile * frames_per_tile;
  const int nf = (int)((n_out - t0) < frames_per_tile ? (n_out - t0) : frames_per_tile);
  const int span = (nf - 1) * hop + frame_length;
  const float* a = audio + b * stride;
  const int64_t start = t0 * hop - pad;
  for (int j0 = threadIdx.x; j0 < span; j0 += 1024) {   // four independent loads in flight per thread
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = start + j0 + 256 * u;
      v[u] = (i >= 0 && i < n_samples && j0 + 256 * u < span) ? a[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + 256 * u < span) sq[j0 + 256 * u] = v[u] * v[u];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = wave; f < nf; f += 4) {
    const float* s = sq + f * hop;
    float acc = 0.0f;
    for (int j = lane; j < frame_length; j += 64) acc += s[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[b * n_out + t0 + f] = sqrtf(acc / (float)frame_length);
  }
}

int64_t mm_rms_num_frames(int64_t n_samples, int32_t frame_length, int32_t hop_length, int32_t center) {
  if (n_samples < 1 || frame_length < 1 || hop_length < 1) return MM_ERR_INVALID_ARG;
  const int64_t padded = n_samples + (center ? 2 * (int64_t)(frame_length / 2) : 0);
  if (padded < frame_length) return MM_ERR_INVALID_ARG;
  return 1 + (padded - frame_length) / hop_length;
}

int mm_rms_f32(const float* d_audio, int64_t batch, int64_t n_samples, int64_t stride, int32_t frame_length,
               int32_t hop_length, int32_t center, float* d_rms, void* stream) {
  if (!d_audio || !d_rms || batch < 1 || stride < n_samples) return MM_ERR_INVALID_ARG;
  const int64_t n_out = mm_rms_num_frames(n_samples, frame_length, hop_length, center);
  if (n_out < 0) return (int)n_out;
  const int64_t total = batch * n_out;
  // tiled kernel: ~16 frames per workgroup (small tiles keep many workgroups per CU in flight), LDS
  // between 16 and 64 KB
  int64_t want = (int64_t)frame_length + 15 * (int64_t)hop_length;
  const int lds_floats = (int)(want < 4096 ? 4096 : (want > 16384 ? 16384 : want));
  if (frame_length <= lds_floats) {
    int64_t fpt = (lds_floats - frame_length) / hop_length + 1;
    if (fpt > 64) fpt = 64;
    if (fpt > n_out) fpt = n_out;
    const int64_t tpc = (n_out + fpt - 1) / fpt;
    if (batch * tpc > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    const size_t lds = (size_t)((fpt - 1) * hop_length + frame_length) * 4;
    hipLaunchKernelGGL(rms_tile_kernel, dim3((unsigned)(batch * tpc)), dim3(256), lds, (hipStream_t)stream, d_audio,
                       n_samples, stride, frame_length, hop_length, center ? frame_length / 2 : 0, n_out, (int)fpt, tpc,
                       d_rms);
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  const int64_t grid = (total + 3) / 4;
  if (grid > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rms_frames_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, d_audio,
                     n_samples, stride, frame_length, hop_length, center ? frame_length / 2 : 0, n_out, total, d_rms);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

// ------------------------------------------------------------------------------------------
// Input side: what librosa.load(path, sr=sigSr, mono=False) does before the hot
// path (script/mfcc.py:284,373) -- PCM decode to float32 in [-1, 1) and sample-rate conversion.
// ------------------------------------------------------------------------------------------
// interleaved PCM frames -> planar float32 [channels][n]; fmt: 1 = u8, 2 = s16, 3 = s24 (packed), 4 = s32,
// 5 = f32, 6 = f64 (little endian; scaling as libsndfile / soundfile: s16 / 32768, s24 / 2^23, s32 / 2^31,
// u8 (x - 128) / 128)
__global__ __launch_bounds__(256) void pcm_decode_kernel(const unsigned char* __restrict__ raw, int fmt, int channels,
                                                         int64_t n, float* __restrict__ out, int64_t out_stride) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * channels) return;
  const int64_t fr = idx / channels;
  const int ch = (int)(idx - fr * channels);
  float v;
  switch (fmt) {
    case 1: v = ((float)raw[idx] - 128.0f) * (1.0f / 128.0f); break;
    case 2: { const short q = (short)((unsigned)raw[2 * idx] | ((unsigned)raw[2 * idx + 1] << 8)); v = (float)q * (1.0f / 32768.0f); break; }
    case 3: { int q = (int)((unsigned)raw[3 * idx] | ((unsigned)raw[3 * idx + 1] << 8) | ((unsigned)raw[3 * idx + 2] << 16));
              q = (q << 8) >> 8; v = (float)q * (1.0f / 8388608.0f); break; }
    case 4: { const int q = (int)((unsigned)raw[4 * idx] | ((unsigned)raw[4 * idx + 1] << 8) | ((unsigned)raw[4 * idx + 2] << 16) |
                                  ((unsigned)raw[4 * idx + 3] << 24));
              v = (float)((double)q * (1.0 / 2147483648.0)); break; }
    case 5: { unsigned u = (unsigned)raw[4 * idx] | ((unsigned)raw[4 * idx + 1] << 8) | ((unsigned)raw[4 * idx + 2] << 16) |
                           ((unsigned)raw[4 * idx + 3] << 24);
              v = __uint_as_float(u); break; }
    default: { unsigned long long u = 0;
               for (int b = 0; b < 8; ++b) u |= (unsigned long long)raw[8 * idx + b] << (8 * b);
               v = (float)__longlong_as_double((long long)u); break; }
  }
  out[(int64_t)ch * out_stride + fr] = v;
}

// 16-bit PCM, mono or stereo (what almost every WAVE file is): a thread takes 16 bytes = 8 samples with ONE load and
// stores whole float4s per channel -- the byte-wise kernel above issues two 1-byte loads and one 4-byte store per sample
// (0.59 ms for 256 stereo clips x 10 s x 44.1 kHz = 2.3 TB/s of bytes read + written).  Same arithmetic (q / 32768).
extern "C++" {
template <int CH>
__global__ __launch_bounds__(256) void pcm_decode_s16_kernel(const uint4* __restrict__ raw, int64_t n_vec, float* __restrict__ out,
                                                             int64_t out_stride) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += stride) {
    const uint4 w = raw[v];
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = (float)(short)(u[i] & 0xFFFFu) * (1.0f / 32768.0f);
      f[2 * i + 1] = (float)(short)(u[i] >> 16) * (1.0f / 32768.0f);
    }
    if (CH == 1) {
      float4* o = reinterpret_cast<float4*>(out + 8 * v);
      o[0] = make_float4(f[0], f[1], f[2], f[3]);
      o[1] = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      *reinterpret_cast<float4*>(out + 4 * v) = make_float4(f[0], f[2], f[4], f[6]);
      *reinterpret_cast<float4*>(out + out_stride + 4 * v) = make_float4(f[1], f[3], f[5], f[7]);
    }
  }
}
// 32-bit float data, mono or stereo: 32 bytes (two 16-byte loads) per thread, 16-byte stores per channel
template <int CH>
__global__ __launch_bounds__(256) void pcm_decode_f32_kernel(const float4* __restrict__ raw, int64_t n_vec, float* __restrict__ out,
                                                             int64_t out_stride) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += stride) {
    const float4 a = raw[2 * v], b = raw[2 * v + 1];
    if (CH == 1) {
      float4* o = reinterpret_cast<float4*>(out + 8 * v);
      o[0] = a; o[1] = b;
    } else {
      *reinterpret_cast<float4*>(out + 4 * v) = make_float4(a.x, a.z, b.x, b.z);
      *reinterpret_cast<float4*>(out + out_stride + 4 * v) = make_float4(a.y, a.w, b.y, b.w);
    }
  }
}
}  // extern "C++"

int mm_pcm_decode_f32(const void* d_raw, int32_t fmt, int32_t channels, int64_t n_frames, float* d_out, int64_t out_stride,
                      void* stream) {
  if (!d_raw || !d_out || fmt < 1 || fmt > 6 || channels < 1 || n_frames < 1 || out_stride < n_frames) return MM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  int64_t done = 0;                                      // frames taken by the vector kernel
  if ((fmt == 2 || fmt == 5) && (channels == 1 || channels == 2) && (((uintptr_t)d_raw | (uintptr_t)d_out) & 15) == 0 &&
      (channels == 1 || (out_stride & 3) == 0)) {
    const int64_t fpv = 8 / channels, n_vec = n_frames / fpv;      // a thread takes 8 samples (16 / 32 bytes)
    if (n_vec > 0) {
      const dim3 gd((unsigned)std::min<int64_t>((n_vec + 255) / 256, 256 * 16)), bd(256);
      if (fmt == 2) {
        if (channels == 1) hipLaunchKernelGGL(pcm_decode_s16_kernel<1>, gd, bd, 0, st, (const uint4*)d_raw, n_vec, d_out, out_stride);
        else hipLaunchKernelGGL(pcm_decode_s16_kernel<2>, gd, bd, 0, st, (const uint4*)d_raw, n_vec, d_out, out_stride);
      } else {
        if (channels == 1) hipLaunchKernelGGL(pcm_decode_f32_kernel<1>, gd, bd, 0, st, (const float4*)d_raw, n_vec, d_out, out_stride);
        else hipLaunchKernelGGL(pcm_decode_f32_kernel<2>, gd, bd, 0, st, (const float4*)d_raw, n_vec, d_out, out_stride);
      }
      done = n_vec * fpv;
    }
  }
  const int64_t rest = n_frames - done;
  if (rest > 0) {
    const int64_t total = rest * channels;
    if ((total + 255) / 256 > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    const int bps = fmt == 1 ? 1 : fmt == 2 ? 2 : fmt == 3 ? 3 : fmt == 6 ? 8 : 4;
    hipLaunchKernelGGL(pcm_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const unsigned char*)d_raw + done * channels * bps, fmt, channels, rest, d_out + done, out_stride);
  }
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

// Rational-ratio polyphase FIR sample-rate conversion: output m = sum_j h[ph + j L] x[ih - j] with t = m M + c,
// ih = t div L, ph = t mod L, zero signal outside the clip -- upfirdn with the filter delay c removed, n_out =
// ceil(n L / M) (what scipy.signal.resample_poly and librosa.resample return); float64 accumulation.
// A thread computes MM_RS_P outputs of ONE phase (m, m + F, m + 2F, ..., F a multiple of L), so a tap is fetched
// once for all of them, and the taps come in OUTPUT-phase order in records of four, hq4[j / 4][t][j % 4] = h[ph_t +
// j L] with ph_t = (t M + c) mod L for the output index t within a period: adjacent threads = adjacent outputs read
// adjacent 16-byte records, their input samples lie within a few cache lines of each other and are fetched four at
// a time (16-byte loads at 4-byte aligned addresses).  (The first version -- one
// thread per output, taps in polyphase order [L][tpp], i.e. a 2 KB stride between lanes -- ran at 0.3 T
// multiply-adds per second: 66 ms for 256 ten-second clips 44.1 -> 16 kHz.)
#define MM_RS_P 4
struct __attribute__((packed, aligned(4))) MmRsFloat4U { float x, y, z, w; };
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t n_in, int64_t in_stride,
                                                       const float* __restrict__ hq, int L, int M, int tpp4, int64_t c,
                                                       int64_t n_out, int64_t F, float* __restrict__ y) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const int64_t r = blockIdx.y;
  const float* xr = x + r * in_stride;
  const int t = (int)(f % L);
  const int64_t t0 = f * M + c;                   // F is a multiple of L: output f + i F has the phase of output f
  const int64_t ih0 = t0 / L, step = (F / L) * M;
  double acc[MM_RS_P];
  int64_t ih[MM_RS_P];
#pragma unroll
  for (int i = 0; i < MM_RS_P; ++i) { acc[i] = 0.0; ih[i] = ih0 + i * step; }
  // taps of this output phase, four consecutive j per 16-byte record: record j4 at hq4[j4 * L + t]
  const float4* h4 = reinterpret_cast<const float4*>(hq) + t;
  const float* h1 = hq + 4 * (int64_t)t;
  const int tpp = 4 * tpp4;
  // taps j for which EVERY one of the thread's outputs reads inside the clip: no checks there, whole records only
  int64_t jlo = 0, jhi = tpp;
#pragma unroll
  for (int i = 0; i < MM_RS_P; ++i) {
    const int64_t lo = ih[i] - (n_in - 1), hi = ih[i] + 1;      // valid j: lo <= j < hi
    jlo = lo > jlo ? lo : jlo;
    jhi = hi < jhi ? hi : jhi;
  }
  jlo = (jlo + 3) / 4 * 4;
  jhi = jhi / 4 * 4;
  if (jlo > tpp) jlo = tpp;
  if (jhi < jlo) jhi = jlo;
  auto checked = [&](int64_t ja, int64_t jb) {
    for (int64_t j = ja; j < jb; ++j) {
      const double tap = (double)h1[(j >> 2) * 4 * L + (j & 3)];
#pragma unroll
      for (int i = 0; i < MM_RS_P; ++i) {
        const int64_t idx = ih[i] - j;
        if (idx >= 0 && idx < n_in) acc[i] = fma(tap, (double)xr[idx], acc[i]);
      }
    }
  };
  checked(0, jlo);
#pragma unroll 2
  for (int64_t j = jlo; j < jhi; j += 4) {
    const float4 tp = h4[(j >> 2) * L];
#pragma unroll
    for (int i = 0; i < MM_RS_P; ++i) {
      // x[ih - j - 3 .. ih - j]: one 16-byte load at a 4-byte aligned address
      const MmRsFloat4U xv = *reinterpret_cast<const MmRsFloat4U*>(xr + (ih[i] - j - 3));
      acc[i] = fma((double)tp.x, (double)xv.w, acc[i]);
      acc[i] = fma((double)tp.y, (double)xv.z, acc[i]);
      acc[i] = fma((double)tp.z, (double)xv.y, acc[i]);
      acc[i] = fma((double)tp.w, (double)xv.x, acc[i]);
    }
  }
  checked(jhi, tpp);
#pragma unroll
  for (int i = 0; i < MM_RS_P; ++i) {
    const int64_t m = f + i * F;
    if (m < n_out) y[r * n_out + m] = (float)acc[i];
  }
}

int mm_resample_f32(const float* d_x, int64_t rows, int64_t n_in, int64_t x_stride, const float* d_taps, int32_t L, int32_t M,
                    int32_t taps_per_phase, int64_t half_len, float* d_y, int64_t n_out, void* stream) {
  if (!d_x || !d_taps || !d_y || rows < 1 || n_in < 1 || x_stride < n_in || L < 1 || M < 1 || taps_per_phase < 4 ||
      (taps_per_phase & 3) || half_len < 0 || n_out < 1 || rows > 65535 || (((uintptr_t)d_taps) & 15))
    return MM_ERR_INVALID_ARG;
  if (n_out != (n_in * L + M - 1) / M) return MM_ERR_INVALID_ARG;
  // threads per row: ceil(n_out / P) rounded up to a multiple of L
  const int64_t per = (n_out + MM_RS_P - 1) / MM_RS_P;
  const int64_t F = (per + L - 1) / L * L;
  if ((F + 255) / 256 > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((F + 255) / 256), (unsigned)rows), dim3(256), 0, (hipStream_t)stream, d_x,
                     n_in, x_stride, d_taps, L, M, taps_per_phase / 4, half_len, n_out, F, d_y);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

// The same conversion as a banded GEMM on the matrix pipe (mm_resample.hip.inc).  The host lays the taps out as MFMA A
// operands (modulation_mfcc_amd/audio_io.py: banded_tables): d_atab [NB][ksteps][64] floats, d_lo_off [NB] int32;
// F = c L >= 16 outputs per period (the host picks the smallest multiple of L that wastes <= 13 % of its last block of 16),
// S = F M / L input samples per period, lo_min = first input sample (relative to a
// period's origin, may be negative) of the first block's window, win = floats of one period's window union.
int mm_resample_banded_f32(const float* d_x, int64_t rows, int64_t n_in, int64_t x_stride, const float* d_atab,
                           const int32_t* d_lo_off, int32_t F, int32_t S, int32_t NB, int32_t ksteps, int32_t lo_min,
                           int32_t win, float* d_y, int64_t n_out, void* stream) {
  if (!d_x || !d_atab || !d_lo_off || !d_y || rows < 1 || n_in < 1 || x_stride < n_in || F < 16 || S < 1 ||
      NB != (F + 15) / 16 || ksteps < 8 || (ksteps & 7) || win < 4 * ksteps || n_out < 1 || (((uintptr_t)d_atab) & 15))
    return MM_ERR_INVALID_ARG;
  // bank pattern of one ds_read_b32: lanes (k = 0, 1) x (q = 0 .. 15) read words q S + k -- pad the tile when more than
  // two of them share a bank
  int cnt[32] = {0}, worst = 0;
  for (int k = 0; k < 2; ++k)
    for (int q = 0; q < 16; ++q) worst = std::max(worst, ++cnt[(int)(((int64_t)q * S + k) & 31)]);
  const bool pad = worst > 2;
  auto tile_bytes = [&](int qt) {
    const int64_t fl = (int64_t)(16 * qt - 1) * S + win + 8;        // + the alignment shift and the last 16-byte vector
    return (size_t)(pad ? fl + (fl >> 5) + 1 : fl) * 4;
  };
  // periods per tile = 16 QT: the largest tile of which two fit a CU (two workgroups: one stages while the other
  // multiplies) -- ratios with few blocks per period (1 / 3: NB = 1) get their units from more period tiles
#ifndef MM_RSM_TILE_KB
#define MM_RSM_TILE_KB 80
#endif
#ifndef MM_RSM_WG_PER_CU
#define MM_RSM_WG_PER_CU 4          // small tiles (1 / 3, 2 / 1 ...): four workgroups per CU, 0.76 -> 0.67 ms at 48 -> 16 kHz
#endif
  int QT = 8;
  while (QT > 1 && tile_bytes(QT) > MM_RSM_TILE_KB * 1024) QT >>= 1;
  if (tile_bytes(QT) > MM_LM_LDS_MAX) return MM_ERR_UNSUPPORTED;     // (the caller falls back to mm_resample_f32)
  RsmParams q;
  q.x = d_x; q.rows = rows; q.n_in = n_in; q.x_stride = x_stride; q.atab = d_atab; q.lo_off = d_lo_off;
  q.F = F; q.S = S; q.NB = NB; q.ksteps = ksteps; q.lo_min = lo_min; q.QT = QT;
  q.tile_floats = (int)((int64_t)(16 * QT - 1) * S + win);
  q.y = d_y; q.n_out = n_out;
  const int64_t periods = (n_out + F - 1) / F;
  q.tiles_per_row = (periods + 16 * QT - 1) / (16 * QT);
  q.n_items = rows * q.tiles_per_row;
  static PerDeviceOnce attr_once;
  {
    const int rc = per_device_once(attr_once, "resample_mfma_kernel", [] {
      const void* kf[4] = {(const void*)resample_mfma_kernel<false, false>, (const void*)resample_mfma_kernel<false, true>,
                           (const void*)resample_mfma_kernel<true, false>, (const void*)resample_mfma_kernel<true, true>};
      for (int i = 0; i < 4; ++i)
        if (hipFuncSetAttribute(kf[i], hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) != hipSuccess) return false;
      return true;
    });
    if (rc) return rc;
  }
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  const size_t lds = tile_bytes(QT);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(MM_RSM_WG_PER_CU, (size_t)MM_LM_LDS_MAX / lds));
  const int64_t grid = std::min<int64_t>(q.n_items, (int64_t)per_cu * cus);
  const bool vec = (((uintptr_t)d_x) & 15) == 0 && (x_stride & 3) == 0;
  const dim3 gd((unsigned)grid), bd(256);
  hipStream_t st = (hipStream_t)stream;
  if (pad) { if (vec) hipLaunchKernelGGL((resample_mfma_kernel<true, true>), gd, bd, lds, st, q); else hipLaunchKernelGGL((resample_mfma_kernel<true, false>), gd, bd, lds, st, q); }
  else { if (vec) hipLaunchKernelGGL((resample_mfma_kernel<false, true>), gd, bd, lds, st, q); else hipLaunchKernelGGL((resample_mfma_kernel<false, false>), gd, bd, lds, st, q); }
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

// Measurement aid (bench.py): float4 grid-stride device-to-device copy on the caller's stream -- the
// practical HBM ceiling the stage-isolated rFFT figure is compared with (MI355X_MICROARCH.md quotes
// 6.29 TB/s for this shape of kernel).  n_floats must be a multiple of 4, pointers 16-byte aligned.
__global__ __launch_bounds__(256) void devcopy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + 256 * u < n4) dst[i + 256 * u] = v[u];
  }
}

int mm_devcopy_f32(const float* d_src, float* d_dst, int64_t n_floats, void* stream) {
  if (!d_src || !d_dst || n_floats < 4 || (n_floats & 3) || (((uintptr_t)d_src | (uintptr_t)d_dst) & 15)) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(devcopy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const float4*)d_src, (float4*)d_dst,
                     n_floats / 4);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

}  // extern "C"
