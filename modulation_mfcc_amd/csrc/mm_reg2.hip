// Two register stages for the common non-power-of-two frames: nn = n_fft / 2 = R1 x R2 complex points.
// Its own translation unit (28 instantiations); mm_any.h carries the launch wrappers for mm_api.hip.
#include "mm_any.h"
#include "mm_hb_math.h"

// The textbook 25 ms frame at 16 kHz (400 samples) is 200 = 8 x 25 complex points.  The LDS-resident Stockham passes of
// stft_any_kernel / stft_anyb_kernel pay three passes of [read, twiddle, butterfly, write] per frame; here the transform
// is TWO register stages with ONE LDS exchange between them, as the power-of-two kernels have it:
//   stage 1: Stockham pass of radix R1 straight from global memory -- task (frame, j < R2) loads the sample pairs
//            j + R2 t (t < R1), applies the window, runs the R1-point DFT in registers and stores its R1 outputs;
//   stage 2: pass of radix R2 -- lane = (frame, k < R1), FB = 64 / R1 frames per wave at once: R2 strided reads, the
//            twiddles W_nn^(k t) (loop-invariant per lane: kept in registers), the R2-point DFT, and the outputs go
//            back to the SAME R2 slots (pass 2 of a two-pass Stockham transform is in place per butterfly);
//   split:   the pairs (k, nn - k) of all FB frames into registers, then the power rows over the same LDS;
//   mel:     (filter, frame) pairs, frame fastest, as in stft_anyb_kernel.
// LDS per wave: FB x FP complex points.  BP = R1 | 1 points between a frame's R2 butterflies (odd: stage 1's b64 stores
// are conflict-free); FP >= R2 BP with FP = R1 (mod 32): lane (frame, k) of stage 2 then sits on banks 2 lane, 2 lane + 1.
// Per 1 025 024 frames (BASELINE configs[1] shape, tools/any_time.py), one-frame / batched LDS kernel -> this one:
// n_fft 400: 1.45 / 1.23 -> 0.62 ms, 800: 2.96 -> 1.12 ms (the n_fft 512 kernel: 0.45).  __launch_bounds__(256, 2): two
// workgroups per CU are what the LDS allows, so the register allocation must not take more than half the file (first-stage
// radix 32 / 40 had drifted to 257 - 265 registers, i.e. ONE workgroup per CU: n_fft 1600 4.42 -> 2.41 ms, 2000 7.4 -> 4.4).
template <int R1, int R2>
struct Reg2Geo {
  static constexpr int NN = R1 * R2, FB = 64 / R1, NP = NN / 2 + 1;
  static constexpr int BP = R1 | 1;                                    // butterfly pitch (complex points)
  static constexpr int FP = R2 * BP + ((R1 % 32) - (R2 * BP) % 32 + 32) % 32;      // frame pitch
  static constexpr int PPITCH = (NN + 2) | 1;                          // power-row pitch (floats, odd)
  static constexpr int NPL = (FB * NP + 63) / 64;                      // pairs per lane in the split
  static constexpr unsigned WAVE_BYTES = FB * FP * 8;
  // prefetch of the next batch's sample pairs into registers: first-stage radix 4 / 8 / 12 (at most 32 pairs per lane).
  // Same-box A/B against the kernel without it (ms per 1 025 024 frames): n_fft 200 0.457 -> 0.417, 240 0.347 -> 0.331,
  // 320 0.552 -> 0.507, 400 0.658 -> 0.617, 600 0.922 -> 0.911; radix 16 gains 1.5 % (640, 800) or loses (480: 0.58 ->
  // 0.72, its twiddles then come from LDS in every batch), radix 24 loses 3 %: off from 16 up.
#ifdef MM_REG2_NO_PF
  static constexpr bool PF = false;
#else
  static constexpr bool PF = ((FB * R2 + 63) / 64) * R1 <= 32 && R1 <= 12;
#endif
  static_assert(FP >= R2 * BP && FP % 32 == R1 % 32 && FB * PPITCH * 4 <= (int)WAVE_BYTES && FB >= 1, "layout");
};

template <int R1, int R2, int MODE>
__global__ __launch_bounds__(256, 2) void stft_reg2_kernel(AnyParams p) {
  using G = Reg2Geo<R1, R2>;
  constexpr int NN = G::NN, FB = G::FB, NP = G::NP, BP = G::BP, FP = G::FP, PPITCH = G::PPITCH, NPL = G::NPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x & 63, grp = threadIdx.x >> 6;
  char* base = smem + (size_t)grp * G::WAVE_BYTES;
  float* ctab = reinterpret_cast<float*>(smem + (size_t)4 * G::WAVE_BYTES);
  for (int i = threadIdx.x; i < p.tab_floats; i += 256) ctab[i] = p.tabpack[i];
  __syncthreads();
  const float2* c_win2 = reinterpret_cast<const float2*>(ctab);
  const float2* c_tw = reinterpret_cast<const float2*>(ctab + p.o_tw);
  const float2* c_split = reinterpret_cast<const float2*>(ctab + p.o_split);
  const float* c_melw = ctab + p.o_melw;
  const int* c_mstart = reinterpret_cast<const int*>(ctab + p.o_mstart);
  const int* c_mlen = reinterpret_cast<const int*>(ctab + p.o_mlen);
  const int* c_moff = reinterpret_cast<const int*>(ctab + p.o_moff);
  hb_c<float>* zA = reinterpret_cast<hb_c<float>*>(base);
  float* P = reinterpret_cast<float*>(base);

  // stage 2's lane: frame fr2, butterfly k2 (lanes >= FB R1 idle); its R2 - 1 twiddles W_nn^(k2 t) stay in registers
  // for the whole launch
  const int fr2 = tid / R1, k2 = tid - fr2 * R1;
  // PF: the next batch's sample pairs are prefetched into registers (below); the twiddles then come from LDS in every
  // batch instead of living in 2 (R2 - 1) registers -- both together do not fit two waves per SIMD
  constexpr int NR1 = (FB * R2 + 63) / 64;
  constexpr bool PF = G::PF;
  hb_c<float> tw2[PF ? 1 : R2 - 1];
  if constexpr (!PF) {
#pragma unroll
    for (int t = 1; t < R2; ++t) { const float2 w = c_tw[k2 * t]; tw2[t - 1].x = w.x; tw2[t - 1].y = w.y; }
  }

  const int fpb = 4 * p.frames_per_group;
  const int64_t tiles = (p.n_frames + fpb - 1) / fpb;
  const int64_t b = blockIdx.x / tiles;
  const int64_t t0 = (blockIdx.x % tiles) * fpb + (int64_t)grp * p.frames_per_group;
  const float* a = p.audio + b * p.stride;
  float vmax = -INFINITY;
  const float pre = p.preemph;
  // The NEXT batch's sample pairs are requested while this batch is in its second stage, split and mel sweep: a batch was
  // NR1 rounds of [R1 loads, wait, butterfly] -- ~2 us of memory latency each, most of the ~11 us a batch of eight
  // 400-sample frames took.  NR1 x R1 pairs in registers (at most 40); batches inside the clip without pre-emphasis only.
  float2 pf[PF ? NR1 : 1][PF ? R1 : 1];
  bool have_pf = false;
  auto batch_geo = [&](int f0, int64_t& tb, int& nf, int64_t& s_first, bool& inside) {
    tb = t0 + f0;
    nf = (int)(p.n_frames - tb < FB ? p.n_frames - tb : FB);
    s_first = tb * p.hop - (p.n_fft >> 1);
    const int64_t s_last = (tb + nf - 1) * p.hop - (p.n_fft >> 1);
    // (s_first >= 1: the pre-emphasis tap of the batch's first sample)
    inside = s_first >= 1 && s_last + p.n_fft <= p.n_samples;      // wave-uniform
  };
  auto prefetch = [&](int f0) {            // -> have_pf
    have_pf = false;
    if (!PF || f0 >= p.frames_per_group || t0 + f0 >= p.n_frames || pre != 0.0f) return;
    int64_t tb, s_first; int nf; bool inside;
    batch_geo(f0, tb, nf, s_first, inside);
    if (!inside) return;
#pragma unroll
    for (int r = 0; r < (PF ? NR1 : 0); ++r) {
      int tt = tid + 64 * r;
      tt = tt < nf * R2 ? tt : nf * R2 - 1;               // clamped: no branch around a load
      const int fr = tt / R2, j = tt - fr * R2;
      const float* af = a + s_first + (int64_t)fr * p.hop + 2 * j;
#pragma unroll
      for (int t = 0; t < R1; ++t) {
        const MmAnyFloat2U xv = *reinterpret_cast<const MmAnyFloat2U*>(af + 2 * R2 * t);
        pf[r][t] = make_float2(xv.x, xv.y);
      }
    }
    have_pf = true;
  };
  prefetch(0);
  for (int f0 = 0; f0 < p.frames_per_group; f0 += FB) {
    int64_t tb, s_first; int nf; bool inside;
    if (t0 + f0 >= p.n_frames) break;                     // wave-uniform
    batch_geo(f0, tb, nf, s_first, inside);
    // ---- stage 1: radix R1 from global memory (one task loop per way of loading: the choice is wave-uniform) ----
    auto stage1 = [&](auto load_pair) {
      for (int tt = tid; tt < nf * R2; tt += 64) {
        const int fr = tt / R2, j = tt - fr * R2;
        hb_c<float> v[R1];
        const int64_t s0 = s_first + (int64_t)fr * p.hop + 2 * j;
#pragma unroll
        for (int t = 0; t < R1; ++t) {
          const float2 w = c_win2[j + R2 * t];
          const float2 x = load_pair(s0 + 2 * R2 * t);
          v[t].x = x.x * w.x; v[t].y = x.y * w.y;
        }
        hb_dft<float, R1>(v);
        hb_c<float>* o = zA + fr * FP + j * BP;
#pragma unroll
        for (int u = 0; u < R1; ++u) o[u] = v[hb_perm<R1>(u)];
      }
    };
    if (PF && have_pf) {
      // the pairs are in registers (requested during the previous batch)
#pragma unroll
      for (int r = 0; r < (PF ? NR1 : 0); ++r) {
        const int tt = tid + 64 * r;
        if (tt < nf * R2) {
          const int fr = tt / R2, j = tt - fr * R2;
          hb_c<float> v[R1];
#pragma unroll
          for (int t = 0; t < R1; ++t) {
            const float2 w = c_win2[j + R2 * t];
            v[t].x = pf[r][t].x * w.x; v[t].y = pf[r][t].y * w.y;
          }
          hb_dft<float, R1>(v);
          hb_c<float>* o = zA + fr * FP + j * BP;
#pragma unroll
          for (int u = 0; u < R1; ++u) o[u] = v[hb_perm<R1>(u)];
        }
      }
    } else if (inside && pre == 0.0f) {
      stage1([&](int64_t s) {
        const MmAnyFloat2U xv = *reinterpret_cast<const MmAnyFloat2U*>(a + s);
        return make_float2(xv.x, xv.y);
      });
    } else if (inside) {
      // pre-emphasis y[n] - a y[n-1]: one more 4-byte load per pair; rounded product, then the subtraction (load_sample's
      // arithmetic)
      stage1([&](int64_t s) {
        const float xm = a[s - 1];
        const MmAnyFloat2U xv = *reinterpret_cast<const MmAnyFloat2U*>(a + s);
        return make_float2(xv.x - __fmul_rn(pre, xm), xv.y - __fmul_rn(pre, xv.x));
      });
    } else {
      stage1([&](int64_t s) {
        return make_float2(load_sample(a, s, p.n_samples, pre), load_sample(a, s + 1, p.n_samples, pre));
      });
    }
    wave_lds_sync();
    prefetch(f0 + FB);
    // ---- stage 2: radix R2, in place ----
    if (fr2 < nf) {
      hb_c<float>* zf = zA + fr2 * FP + k2;
      hb_c<float> v[R2];
#pragma unroll
      for (int t = 0; t < R2; ++t) v[t] = zf[t * BP];
#pragma unroll
      for (int t = 1; t < R2; ++t) {
        if constexpr (PF) {
          const float2 w = c_tw[k2 * t];
          hb_c<float> ww; ww.x = w.x; ww.y = w.y;
          v[t] = hb_mul(v[t], ww);
        } else {
          v[t] = hb_mul(v[t], tw2[t - 1]);
        }
      }
      hb_dft<float, R2>(v);
#pragma unroll
      for (int u = 0; u < R2; ++u) zf[u * BP] = v[u];
    }
    wave_lds_sync();
    // ---- real split + power: pairs (k, nn - k) into registers, then the power rows over the same LDS ----
    // (spectrum bin q of a frame sits at q + (q / R1) (BP - R1))
    {
      float pa[NPL], pb[NPL];
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int idx = tid + 64 * i;
        pa[i] = 0.0f; pb[i] = 0.0f;
        if (idx < nf * NP) {
          const int fr = idx / NP, k = idx - fr * NP;
          const int kb = k ? NN - k : 0;
          const hb_c<float>* Zf = zA + fr * FP;
          const hb_c<float> za = Zf[k + (k / R1) * (BP - R1)], zb = Zf[kb + (kb / R1) * (BP - R1)];
          const float ex = 0.5f * (za.x + zb.x), ey = 0.5f * (za.y - zb.y);
          const float dx = 0.5f * (za.x - zb.x), dy = 0.5f * (za.y + zb.y);
          const float ox = dy, oy = -dx;                       // -i D
          const float2 w = c_split[k];
          const float tx = w.x * ox - w.y * oy, ty = w.x * oy + w.y * ox;
          const float ar = ex + tx, ai = ey + ty, br = ex - tx, bi = ey - ty;
          pa[i] = ar * ar + ai * ai;
          pb[i] = br * br + bi * bi;
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int idx = tid + 64 * i;
        if (idx < nf * NP) {
          const int fr = idx / NP, k = idx - fr * NP;
          P[fr * PPITCH + k] = pa[i];
          P[fr * PPITCH + NN - k] = pb[i];
        }
      }
    }
    wave_lds_sync();
    if (MODE == 0) {
      for (int fr = 0; fr < nf; ++fr) {
        float* o = p.out_power + (b * p.n_frames + tb + fr) * p.n_bins;
        for (int k = tid; k < p.n_bins; k += 64) o[k] = P[fr * PPITCH + k];
      }
    } else {
      const float rnf = 1.0f / (float)nf;
      for (int idx = tid; idx < nf * p.n_mels; idx += 64) {
        const int m = (int)(((float)idx + 0.5f) * rnf), fr = idx - m * nf;
        const float* w = c_melw + c_moff[m];
        const float* pp = P + fr * PPITCH + c_mstart[m];
        const int len = c_mlen[m];
        float acc = 0.0f;
        int j = 0;
        for (; j + 4 <= len; j += 4) {
          const float w0 = w[j], w1 = w[j + 1], w2 = w[j + 2], w3 = w[j + 3];
          const float q0 = pp[j], q1 = pp[j + 1], q2 = pp[j + 2], q3 = pp[j + 3];
          acc = fmaf(w0, q0, acc); acc = fmaf(w1, q1, acc); acc = fmaf(w2, q2, acc); acc = fmaf(w3, q3, acc);
        }
        for (; j < len; ++j) acc = fmaf(w[j], pp[j], acc);
        const float db = 10.0f * log10f(fmaxf(p.amin, acc)) - p.db_offset;
        p.out_logmel[(b * p.n_mels + m) * p.n_frames + tb + fr] = db;
        vmax = fmaxf(vmax, db);
      }
    }
    wave_lds_sync();
  }
  if (MODE == 1) {
    vmax = wave_max(vmax);
    if (tid == 0 && vmax > -INFINITY) atomicMax(p.clip_key + b, float_key(vmax));
  }
}

// ---- host side ----
#define MM_REG2_PAIRS(X) X(4, 25) X(8, 25) X(12, 25) X(16, 25) X(20, 25) X(24, 25) X(32, 25) X(40, 25) \
  X(8, 15) X(16, 15) X(8, 20) X(16, 20) X(24, 20) X(32, 24)

// n_fft = 2 nn: 200, 400, 600, 800, 1000, 1200, 1600, 2000 (R2 = 25); 240, 480 (15); 320, 640, 960 (20); 1536 (24)
bool reg2_pick(int nn, int* r1, int* r2) {
#define X(A, B) if (nn == (A) * (B)) { *r1 = A; *r2 = B; return true; }
  MM_REG2_PAIRS(X)
#undef X
  return false;
}

size_t reg2_lds_bytes(int r1, int r2, int tab_floats) {
#define X(A, B) if (r1 == A && r2 == B) return 4 * (size_t)Reg2Geo<A, B>::WAVE_BYTES + (size_t)tab_floats * 4;
  MM_REG2_PAIRS(X)
#undef X
  return 0;
}

int reg2_frames_per_wave(int r1) { const int fb = 64 / r1; return 32 / fb * fb; }

bool reg2_set_attr(int r1, int r2, int bytes) {
#define X(A, B) if (r1 == A && r2 == B) \
    return hipFuncSetAttribute((const void*)stft_reg2_kernel<A, B, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && \
           hipFuncSetAttribute((const void*)stft_reg2_kernel<A, B, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  MM_REG2_PAIRS(X)
#undef X
  return false;
}

void reg2_launch(int r1, int r2, int mode, dim3 grid, size_t lds, hipStream_t st, const AnyParams& q) {
#define X(A, B) if (r1 == A && r2 == B) { \
    if (mode == 0) hipLaunchKernelGGL((stft_reg2_kernel<A, B, 0>), grid, dim3(256), lds, st, q); \
    else hipLaunchKernelGGL((stft_reg2_kernel<A, B, 1>), grid, dim3(256), lds, st, q); \
    return; }
  MM_REG2_PAIRS(X)
#undef X
}
