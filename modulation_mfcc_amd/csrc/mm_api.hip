// libmodmfcc device code + C ABI (include/modmfcc.h).  gfx950 (MI355X / CDNA4) only.
//
// Hot path = what librosa.feature.mfcc does for the call at script/mfcc.py:387:
//   A1 centre-pad + frame + periodic Hann   A2 rFFT   A3 |.|^2   A4 Slaney mel   A5 dB + per-clip
//   80 dB clamp   A6 DCT-II ortho           A8 rFFT over coefficient trajectories (build-defined)
//
// Kernels in this file (mm_api.hip: plan, dispatch, the C ABI of the hot path) (generic path: any power-of-two n_fft in [32, 4096], any hop):
//   stft_generic_kernel<MODE>  one wave per frame; half-length complex FFT (radix-2 DIT) in LDS,
//                              real-FFT split, |.|^2, then either the power row (MODE 0) or the
//                              CSR mel filterbank + 10*log10 + per-clip max (MODE 1)
//   dct_clamp_kernel           top_db clamp against the per-clip max + DCT-II, lane <-> frame
//   rfft_generic_kernel        stage-isolated batched rFFT of zero-padded rows (also the
//                              trajectory rFFT of the modulation spectrum)
// The register radix-16 kernels for n_fft 512/1024/2048 live in mm_fft16.hip.inc.
#include "mm_common.h"
#include "mm_plan.h"

thread_local std::string g_hip_err;

struct StftParams {
  const float* audio;
  int64_t batch, n_samples, stride, n_frames;
  int n_fft, log2nc, hop, n_bins, n_mels;
  float preemph, amin, db_offset;
  const float* window;
  const float2* tw;
  const int* mel_start;
  const int* mel_len;
  const int* mel_off;
  const float* mel_w;
  float* out_power;   // MODE 0: [B][T][n_bins]
  float* out_logmel;  // MODE 1: [B][n_mels][T]
  int* clip_key;      // MODE 1: [B]
  int frames_per_wave;
};

template <int MODE>
__global__ __launch_bounds__(256) void stft_generic_kernel(StftParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = 1 << p.log2nc;
  const size_t wave_bytes = ((size_t)nc * 8 + (size_t)(nc + 1) * 4 + 15) & ~(size_t)15;
  float2* z = reinterpret_cast<float2*>(smem + wave * wave_bytes);
  float* P = reinterpret_cast<float*>(smem + wave * wave_bytes + (size_t)nc * 8);

  const int fpb = 4 * p.frames_per_wave;
  const int64_t tiles = (p.n_frames + fpb - 1) / fpb;
  const int64_t b = blockIdx.x / tiles;
  const int64_t t0 = (blockIdx.x % tiles) * fpb + (int64_t)wave * p.frames_per_wave;
  const float* a = p.audio + b * p.stride;
  const int tw_stride = MM_TW_N / p.n_fft;
  float vmax = -INFINITY;

  for (int f = 0; f < p.frames_per_wave; ++f) {
    const int64_t t = t0 + f;
    if (t >= p.n_frames) break;  // wave-uniform
    const int64_t base = t * p.hop - (p.n_fft >> 1);
    // a frame inside the clip: unconditional loads, several in flight (load_sample's bounds test is a branch around
    // every load, i.e. one round trip to memory after the other)
    if (base >= 1 && base + p.n_fft <= p.n_samples) {
      const float* af = a + base;
      const float pre = p.preemph;
#pragma unroll 4
      for (int n = lane; n < nc; n += 64) {
        const float xm = pre != 0.0f ? af[2 * n - 1] : 0.0f;
        float x0 = af[2 * n], x1 = af[2 * n + 1];
        if (pre != 0.0f) { x1 -= __fmul_rn(pre, x0); x0 -= __fmul_rn(pre, xm); }
        z[__brev((unsigned)n) >> (32 - p.log2nc)] = make_float2(x0 * p.window[2 * n], x1 * p.window[2 * n + 1]);
      }
    } else {
      for (int n = lane; n < nc; n += 64) {
        const float x0 = load_sample(a, base + 2 * n, p.n_samples, p.preemph) * p.window[2 * n];
        const float x1 = load_sample(a, base + 2 * n + 1, p.n_samples, p.preemph) * p.window[2 * n + 1];
        z[__brev((unsigned)n) >> (32 - p.log2nc)] = make_float2(x0, x1);
      }
    }
    wave_lds_sync();
    wave_cfft_lds(z, p.log2nc, p.tw, lane);
    for (int k = lane; k <= (nc >> 1); k += 64) {
      float2 xa, xb;
      real_split(z, k, nc, p.tw, tw_stride, xa, xb);
      P[k] = xa.x * xa.x + xa.y * xa.y;
      P[nc - k] = xb.x * xb.x + xb.y * xb.y;
    }
    wave_lds_sync();
    if (MODE == 0) {
      float* o = p.out_power + (b * p.n_frames + t) * p.n_bins;
      for (int k = lane; k < p.n_bins; k += 64) o[k] = P[k];
    } else {
      for (int m = lane; m < p.n_mels; m += 64) {
        const float* w = p.mel_w + p.mel_off[m];
        const float* pp = P + p.mel_start[m];
        const int len = p.mel_len[m];
        float acc = 0.0f;
        for (int j = 0; j < len; ++j) acc = fmaf(w[j], pp[j], acc);
        const float db = 10.0f * log10f(fmaxf(p.amin, acc)) - p.db_offset;
        p.out_logmel[(b * p.n_mels + m) * p.n_frames + t] = db;
        vmax = fmaxf(vmax, db);
      }
    }
    wave_lds_sync();
  }
  if (MODE == 1) {
    vmax = wave_max(vmax);
    if (lane == 0 && vmax > -INFINITY) atomicMax(p.clip_key + b, float_key(vmax));
  }
}

// Clamp against the per-clip max and apply the DCT-II matrix.  lane <-> frame so that both the
// logmel reads [B][n_mels][T] and the MFCC writes [B][n_mfcc][T] are coalesced; the DCT row
// index is wave-uniform, so the coefficients come through the scalar cache.
__global__ __launch_bounds__(256) void dct_clamp_kernel(const float* __restrict__ logmel,
                                                         const int* __restrict__ clip_key,
                                                         const float* __restrict__ dct_t /*[n_mels][KP]*/,
                                                         float* __restrict__ out, int64_t n_frames,
                                                         int n_mels, int n_mfcc, int kp, float top_db) {
  const int64_t bpc = (n_frames + 255) / 256;
  const int64_t b = blockIdx.x / bpc;
  const int64_t t = (blockIdx.x % bpc) * 256 + threadIdx.x;
  if (t >= n_frames) return;
  const float thr = top_db >= 0.0f ? key_float(clip_key[b]) - top_db : -INFINITY;
  const float* lm = logmel + b * n_mels * n_frames + t;
  float* o = out + b * n_mfcc * n_frames + t;
  for (int k0 = 0; k0 < n_mfcc; k0 += MM_DCT_KB) {
    float acc[MM_DCT_KB];
#pragma unroll
    for (int kk = 0; kk < MM_DCT_KB; ++kk) acc[kk] = 0.0f;
#pragma unroll 4
    for (int m = 0; m < n_mels; ++m) {
      const float x = fmaxf(lm[(int64_t)m * n_frames], thr);
      const float* d = dct_t + (size_t)m * kp + k0;
#pragma unroll
      for (int kk = 0; kk < MM_DCT_KB; ++kk) acc[kk] = fmaf(d[kk], x, acc[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < MM_DCT_KB; ++kk)
      if (k0 + kk < n_mfcc) o[(int64_t)(k0 + kk) * n_frames] = acc[kk];
  }
}

__global__ void decode_keys_kernel(int* inout, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) reinterpret_cast<float*>(inout)[i] = key_float(inout[i]);
}


__global__ __launch_bounds__(256) void rfft_generic_kernel(RfftParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nc = 1 << p.log2nc;
  float2* z = reinterpret_cast<float2*>(smem) + (size_t)wave * nc;
  const int tw_stride = MM_TW_N / p.n;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * p.rows_per_wave;
  for (int f = 0; f < p.rows_per_wave; ++f) {
    const int64_t r = r0 + f;
    if (r >= p.rows) break;
    const float* a = p.in + r * p.in_stride;
    for (int n = lane; n < nc; n += 64) {
      const float x0 = (2 * n < p.in_len) ? a[2 * n] : 0.0f;
      const float x1 = (2 * n + 1 < p.in_len) ? a[2 * n + 1] : 0.0f;
      z[__brev((unsigned)n) >> (32 - p.log2nc)] = make_float2(x0, x1);
    }
    wave_lds_sync();
    wave_cfft_lds(z, p.log2nc, p.tw, lane);
    float2* o = reinterpret_cast<float2*>(p.out) + r * (nc + 1);
    for (int k = lane; k <= (nc >> 1); k += 64) {
      float2 xa, xb;
      real_split(z, k, nc, p.tw, tw_stride, xa, xb);
      o[k] = xa;
      o[nc - k] = xb;
    }
    wave_lds_sync();
  }
}

#include "mm_fft16.hip.inc"
#include "mm_logmel16w.hip.inc"
#include "mm_s16.h"
#include "mm_logmel12m.hip.inc"
#include "mm_logmel16h.hip.inc"
#include "mm_wpf.hip.inc"
#include "mm_hb_math.h"
#include "mm_anyfft.hip.inc"

extern "C" {

int mm_version(void) { return MM_VERSION; }

const char* mm_strerror(int s) {
  switch (s) {
    case MM_OK: return "ok";
    case MM_ERR_INVALID_ARG: return "invalid argument";
    case MM_ERR_UNSUPPORTED: return "unsupported configuration";
    case MM_ERR_HIP: return "HIP runtime error";
    case MM_ERR_WORKSPACE: return "workspace too small";
    case MM_ERR_ALLOC: return "allocation failed";
    default: return "unknown status";
  }
}

const char* mm_last_hip_error(void) { return g_hip_err.c_str(); }

int mm_config_default(mm_config* c) {
  if (!c) return MM_ERR_INVALID_ARG;
  // defaults of get_MFCCS_change as the UI calls it (script/main.py:732-748) at 10 kHz
  c->sr = 10000.0;
  c->n_fft = 512;
  c->win_length = 250;
  c->hop_length = 50;
  c->n_mels = 128;
  c->n_mfcc = 13;
  c->fmin = 100.0;
  c->fmax = 10000.0;
  c->preemph = 0.0f;
  c->top_db = 80.0f;
  c->amin = 1e-10f;
  c->center = 1;
  c->n_mod_fft = 0;
  return MM_OK;
}

int mm_config_validate(const mm_config* c) { return mm::validate(c); }

int64_t mm_num_frames(const mm_config* c, int64_t n_samples) {
  if (!c || c->hop_length < 1 || n_samples < 0) return MM_ERR_INVALID_ARG;
  // librosa pads n_fft // 2 samples on both sides and keeps 1 + (padded - n_fft) // hop frames: an ODD n_fft pads
  // one sample less than it consumes
  const int64_t padded = n_samples + 2 * (int64_t)(c->n_fft / 2);
  return padded < c->n_fft ? 0 : 1 + (padded - c->n_fft) / c->hop_length;
}

int32_t mm_num_bins(const mm_config* c) { return c ? c->n_fft / 2 + 1 : MM_ERR_INVALID_ARG; }

int32_t mm_mod_fft_len(const mm_config* c, int64_t n_frames) {
  if (!c || n_frames < 1) return MM_ERR_INVALID_ARG;
  if (c->n_mod_fft) return c->n_mod_fft >= n_frames ? c->n_mod_fft : MM_ERR_INVALID_ARG;
  int64_t n = 32;
  while (n < n_frames) n *= 2;
  // (up to 8192 points: mm_modspec_f32 / the fused tail; beyond: mm_hilbert_rfft_f32 -- the caller's choice, the length
  // is the same rule)
  return n <= ((int64_t)1 << 24) ? (int32_t)n : MM_ERR_UNSUPPORTED;
}

int mm_build_window(const mm_config* c, float* out) {
  int s = mm::validate(c);
  if (s || !out) return s ? s : MM_ERR_INVALID_ARG;
  mm::build_window(*c, out);
  return MM_OK;
}
int mm_build_mel(const mm_config* c, float* out) {
  int s = mm::validate(c);
  if (s || !out) return s ? s : MM_ERR_INVALID_ARG;
  mm::build_mel(*c, out);
  return MM_OK;
}
int mm_build_dct(const mm_config* c, float* out) {
  int s = mm::validate(c);
  if (s || !out) return s ? s : MM_ERR_INVALID_ARG;
  mm::build_dct(*c, out);
  return MM_OK;
}
int mm_build_mel_sweep(const mm_config* c, int n_waves, float* wlo, float* whi, int32_t* d, int32_t* part) {
  int s = mm::validate(c);
  if (s) return s;
  if (n_waves < 1 || !wlo || !whi || !d || !part) return MM_ERR_INVALID_ARG;
  std::vector<float> mel((size_t)c->n_mels * (c->n_fft / 2 + 1));
  mm::build_mel(*c, mel.data());
  mm::MelSweep sw;
  if (!mm::build_mel_sweep(*c, mel.data(), n_waves, &sw)) return MM_ERR_UNSUPPORTED;
  std::memcpy(wlo, sw.wlo.data(), sw.wlo.size() * 4);
  std::memcpy(whi, sw.whi.data(), sw.whi.size() * 4);
  std::memcpy(d, sw.d.data(), sw.d.size() * 4);
  std::memcpy(part, sw.part.data(), sw.part.size() * 4);
  return MM_OK;
}
int mm_build_mel_runs(const mm_config* c, int n_waves, int32_t* hdr, int32_t hdr_cap, float* grp,
                      int32_t grp_cap, int32_t* part, int32_t* counts) {
  int s = mm::validate(c);
  if (s) return s;
  if (n_waves < 1 || !hdr || !grp || !part || !counts) return MM_ERR_INVALID_ARG;
  std::vector<float> mel((size_t)c->n_mels * (c->n_fft / 2 + 1));
  mm::build_mel(*c, mel.data());
  mm::MelSweep sw;
  if (!mm::build_mel_sweep(*c, mel.data(), n_waves, &sw)) return MM_ERR_UNSUPPORTED;
  mm::MelRuns r;
  mm::build_mel_runs(*c, sw, n_waves, &r);
  counts[0] = (int32_t)(r.hdr.size() / 4);
  counts[1] = (int32_t)(r.grp.size() / 8);
  if (counts[0] > hdr_cap || counts[1] > grp_cap) return MM_ERR_WORKSPACE;
  std::memcpy(hdr, r.hdr.data(), r.hdr.size() * 4);
  std::memcpy(grp, r.grp.data(), r.grp.size() * 4);
  std::memcpy(part, r.part.data(), r.part.size() * 4);
  return MM_OK;
}
int mm_build_butter_sos(int order, double wn, double* sos) { return mm::build_butter_sos(order, wn, sos); }

// mm_build_mel_runs emits a 4-bin group as {wlo x4, whi x4}; the kernels want {wlo0, whi0, wlo1, whi1}
// {wlo2, whi2, wlo3, whi3}: (wlo_i, whi_i) is then an aligned register pair and the two accumulators
// advance with one v_pk_fma_f32 per bin, no register shuffling.
static void interleave_run_groups(float* grp, size_t n_groups) {
  for (size_t g = 0; g < n_groups; ++g) {
    float* r = grp + 8 * g;
    const float t[8] = {r[0], r[4], r[1], r[5], r[2], r[6], r[3], r[7]};
    std::memcpy(r, t, sizeof t);
  }
}

// The 16 waves of the fused kernels sit on 4 SIMDs (wave w on SIMD w % 4) and the mel phase is bound by
// instruction issue per SIMD (tools/stamps.py), so which wave walks which part of the run table matters:
// hand the parts out heaviest first to the least loaded SIMD (cost model: ~58 instructions per run +
// ~8.5 per 4-bin group), heavier parts on the older (= favoured) wave of a SIMD.
static std::vector<int> balance_parts_over_simds(const mm::MelRuns& r, const double* extra = nullptr,
                                                 std::vector<int>* wave_of_part = nullptr) {
  const int n = (int)(r.part.size() / 4);
  std::vector<int> out(r.part.size());
  if (wave_of_part) { wave_of_part->assign(n, 0); for (int w = 0; w < n; ++w) (*wave_of_part)[w] = w; }
  if (n != 16) return r.part;
  std::vector<std::pair<double, int>> cost(n);
  for (int w = 0; w < n; ++w) {
    double c = extra ? extra[w] : 0.0;
    for (int i = r.part[w * 4 + 0]; i < r.part[w * 4 + 1]; ++i) c += 58.0 + 8.5 * r.hdr[4 * i + 1];
    cost[w] = {c, w};
  }
  std::sort(cost.begin(), cost.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
    return a.first > b.first || (a.first == b.first && a.second < b.second); });
  double load[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    int best = -1;
    for (int sd = 0; sd < 4; ++sd)
      if (cnt[sd] < 4 && (best < 0 || load[sd] < load[best])) best = sd;
    const int wave = best + 4 * cnt[best];
    std::memcpy(&out[wave * 4], &r.part[cost[i].second * 4], 16);
    if (wave_of_part) (*wave_of_part)[cost[i].second] = wave;
    load[best] += cost[i].first; ++cnt[best];
  }
  return out;
}

// Per-lane constants of the wpf transform (mm_wpf.hip.inc) for n = 512*R: window (or zeros when
// win == nullptr: the plain rFFT kernel does not read it) | W_NC^(l*k1) | W_L^(p*j) | split twiddles.
static std::vector<float> wpf_lane_table(int R, const float* win, const float* tw) {
  const int L = 16 * R, NC = 256 * R;
  std::vector<float> lt((size_t)L * MM_WPF_LT_PITCH, 0.0f);
  for (int l = 0; l < L; ++l) {
    float* r = lt.data() + l * MM_WPF_LT_PITCH;
    const int pq = l % R;
    if (win)
      for (int n1 = 0; n1 < 16; ++n1) { r[2 * n1] = win[2 * L * n1 + 2 * l]; r[2 * n1 + 1] = win[2 * L * n1 + 2 * l + 1]; }
    for (int k1 = 1; k1 < 16; ++k1) {
      const int i1 = ((l * k1) % NC) * (MM_TW_N / NC);          // W_NC^(n2*k1), n2 = lane in frame
      r[32 + 2 * (k1 - 1)] = tw[2 * i1]; r[32 + 2 * (k1 - 1) + 1] = tw[2 * i1 + 1];
      const int i2 = ((pq * k1) % L) * (MM_TW_N / L);            // W_L^(p*j)
      r[64 + 2 * (k1 - 1)] = tw[2 * i2]; r[64 + 2 * (k1 - 1) + 1] = tw[2 * i2 + 1];
    }
    for (int i = 0; i < 8; ++i) {
      const int idx = (l + L * i) * (MM_TW_N / (2 * NC));        // 0.5 * (-i) * W_n^k
      r[96 + 2 * i] = 0.5f * tw[2 * idx + 1]; r[96 + 2 * i + 1] = -0.5f * tw[2 * idx];
    }
  }
  return lt;
}

// n_fft the power-of-two kernels (radix-16 register kernels, stft_generic_kernel) take; everything else in [2, 8192]
// runs on stft_any_kernel
static bool nfft_is_pow2_class(int n) { return n >= 32 && n <= 4096 && (n & (n - 1)) == 0; }

int mm_plan_create(const mm_config* cfg, mm_plan** out) {
  if (!out) return MM_ERR_INVALID_ARG;
  *out = nullptr;
  int s = mm::validate(cfg);
  if (s) return s;
  mm_plan* p = new (std::nothrow) mm_plan();
  if (!p) return MM_ERR_ALLOC;
  p->cfg = *cfg;
  p->n_bins = cfg->n_fft / 2 + 1;
  p->log2nc = nfft_is_pow2_class(cfg->n_fft) ? ilog2(cfg->n_fft) - 1 : 0;
  p->kp = (cfg->n_mfcc + MM_DCT_KB - 1) / MM_DCT_KB * MM_DCT_KB;
  p->db_offset = 10.0f * log10f(fmaxf(cfg->amin, 1.0f));
  p->path = 0;
  p->force_generic = 0;
  p->timing_on = 0;
  p->ev_used = 0;
  std::memset(p->t_sum, 0, sizeof(p->t_sum));
  std::memset(p->t_cnt, 0, sizeof(p->t_cnt));
  p->d_window = nullptr; p->d_tw = nullptr; p->d_mel_start = p->d_mel_len = p->d_mel_off = nullptr;
  p->d_mel_w = nullptr; p->d_dct_t = nullptr;
  p->d_sw_tab = nullptr; p->d_sw_part = nullptr;
  p->d_w16_tab = p->d_lane_tab = nullptr; p->d_w16_part = nullptr; p->w16_ok = 0; p->s16_nr = 0; p->s16_lds_bytes = 0;

  p->d_k2_lane_tab = p->d_k2_mel_lane = nullptr; p->k2_ok = 0; p->d_window_e = nullptr; p->embed = 1;
  p->d_rf2k_lane_tab = nullptr; p->rf2k_ok = 0;
  p->d_h16_tab = nullptr; p->d_h16_part = nullptr; p->h16_ok = 0;
  p->d_s16f_tab = p->d_s16f_dcta = nullptr; p->d_s16f_part = nullptr; p->s16f_ok = 0; p->s16f_flags = 0;
  p->d_m12_a = p->d_m12_dct = p->d_zeros = nullptr; p->m12_ok = 0; p->variant = 0; p->no_fuse = 0; p->no_fuse_tail = 0; p->fuse_tail_wide = 0; p->s16f_red_off = 0; p->d_dctfm_a = nullptr; p->d_dctw_a = nullptr; p->dctw_nk = p->dctw_ch = 0;
  p->sw_n_runs = p->sw_n_tab16 = 0; p->lm_lds_bytes = 0;
  p->num_cus = 256;
  if (hipGetDevice(&p->device) != hipSuccess) {
    g_hip_err = "hipGetDevice failed (no GPU?)";
    delete p;
    return MM_ERR_HIP;
  }
  std::vector<float> win(cfg->n_fft), mel((size_t)cfg->n_mels * p->n_bins),
      dct((size_t)cfg->n_mfcc * cfg->n_mels), dct_t((size_t)cfg->n_mels * p->kp, 0.0f),
      tw(2 * MM_TW_N);
  mm::build_window(*cfg, win.data());
  mm::build_mel(*cfg, mel.data());
  mm::build_dct(*cfg, dct.data());
  mm::build_twiddles(MM_TW_N, tw.data());
  for (int k = 0; k < cfg->n_mfcc; ++k)
    for (int m = 0; m < cfg->n_mels; ++m) dct_t[(size_t)m * p->kp + k] = dct[(size_t)k * cfg->n_mels + m];
  mm::MelCsr csr;
  mm::build_mel_csr(*cfg, mel.data(), &csr);
  int rc = MM_OK;
  if ((rc = upload(&p->d_window, win.data(), win.size() * 4)) ||
      (rc = upload(&p->d_tw, tw.data(), tw.size() * 4)) ||
      (rc = upload(&p->d_mel_start, csr.start.data(), csr.start.size() * 4)) ||
      (rc = upload(&p->d_mel_len, csr.len.data(), csr.len.size() * 4)) ||
      (rc = upload(&p->d_mel_off, csr.off.data(), csr.off.size() * 4)) ||
      (rc = upload(&p->d_mel_w, csr.w.data(), csr.w.size() * 4)) ||
      (rc = upload(&p->d_dct_t, dct_t.data(), dct_t.size() * 4))) {
    mm_plan_destroy(p);
    return rc;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0)
      p->num_cus = prop.multiProcessorCount;
  }
  if (!nfft_is_pow2_class(cfg->n_fft)) {
    // any other length: mixed-radix / Bluestein STFT in LDS (mm_anyfft.hip.inc); the only kernel of such a plan
    std::vector<float> atw, asplit, achirp, abhat;
    if (!any_plan_host(cfg->n_fft, &p->any, &atw, &asplit, &achirp, &abhat, MM_LM_LDS_MAX)) {
      mm_plan_destroy(p);
      return MM_ERR_UNSUPPORTED;
    }
    if ((rc = upload(&p->any.d_tw, atw.data(), std::max<size_t>(atw.size(), 2) * 4)) ||
        (!asplit.empty() && (rc = upload(&p->any.d_split, asplit.data(), asplit.size() * 4))) ||
        (!achirp.empty() && (rc = upload(&p->any.d_chirp, achirp.data(), achirp.size() * 4))) ||
        (!abhat.empty() && (rc = upload(&p->any.d_bhat, abhat.data(), abhat.size() * 4)))) {
      mm_plan_destroy(p);
      return rc;
    }
    {
      // LDSTAB: the constant tables packed into one array (window | tw | split | chirp | mel_w | mel_start | mel_len |
      // mel_off), copied to LDS by every workgroup when they fit beside the frame buffers (64 KB budget with a wave per
      // frame, so that several workgroups stay resident; the whole LDS with a workgroup per frame)
      AnyPlan& ap = p->any;
      std::vector<float> pack(win.begin(), win.end());
      auto put = [&](const void* src, size_t n_floats) {
        const int off = (int)pack.size();
        pack.resize(pack.size() + ((n_floats + 3) & ~(size_t)3), 0.0f);
        if (n_floats) std::memcpy(pack.data() + off, src, n_floats * 4);
        return off;
      };
      ap.o_tw = put(atw.data(), atw.size());
      ap.o_split = put(asplit.data(), asplit.size());
      ap.o_chirp = put(achirp.data(), achirp.size());
      ap.o_melw = put(csr.w.data(), csr.w.size());
      ap.o_mstart = put(csr.start.data(), csr.start.size());
      ap.o_mlen = put(csr.len.data(), csr.len.size());
      ap.o_moff = put(csr.off.data(), csr.off.size());
      ap.tab_floats = (int)pack.size();
      const size_t G = 256 / ap.tpf, budget = ap.tpf == 64 ? 65536 : MM_LM_LDS_MAX;
      // (direct lengths only: the Bluestein path's radix-2 stages are LDS-bound already -- with its twiddles in LDS too
      // n_fft 499 took 33 ms per 1 025 024 frames instead of 14.5)
      ap.lds_tab = ap.M == 0 && G * ap.grp_bytes + pack.size() * 4 <= budget;
      // two register stages (mm_reg2.hip): nn = R1 x R2 with an instantiation -- its own (smaller) buffers beside the tables
      ap.reg2 = ap.reg2_r2 = 0;
      {
        int r1 = 0, r2 = 0;
        if (ap.packed && ap.M == 0 && reg2_pick(ap.nn, &r1, &r2) && reg2_lds_bytes(r1, r2, ap.tab_floats) <= 80 * 1024 &&
            reg2_set_attr(r1, r2, MM_LM_LDS_MAX)) {
          ap.reg2 = r1; ap.reg2_r2 = r2;
        }
      }
      if ((ap.lds_tab || ap.reg2) && (rc = upload(&ap.d_tabpack, pack.data(), pack.size() * 4))) {
        mm_plan_destroy(p);
        return rc;
      }
      // batched form (stft_anyb_kernel): TWO frames per wave at once, for the small packed direct lengths whose tables
      // sit in LDS on a wave per frame.  Measured per 1 025 024 frames (tools/any_time.py, forms 1 / 2 / 3 / 4 frames):
      // n_fft 400: 1.49 / 1.24 / 1.25 / 1.56 ms; 600: 1.74 / 1.88 / 2.14 / 2.14; 800: 2.93 / 3.39 / 3.39 / 3.39 -- the
      // batch fills the lanes of a 200-point frame's passes (25 / 40 butterflies), beyond that the LDS it takes costs
      // occupancy and the next frame's register prefetch of the one-frame kernel is worth more.
      ap.fb = 0;
      if (ap.lds_tab && ap.tpf == 64 && ap.packed && ap.M == 0 && ap.nn <= 256) {
        const int fb = 2;
        const size_t per_wave = (size_t)2 * fb * ap.nn * 8;
        if ((size_t)fb * (ap.nn + 2) * 4 <= (size_t)fb * ap.nn * 8 && 4 * per_wave + pack.size() * 4 <= 80 * 1024) {
          ap.fb = fb; ap.fb_grp_bytes = (unsigned)per_wave;
        }
      }
    }
    const void* kfn[8] = {(const void*)stft_any_kernel<0, 64, false>, (const void*)stft_any_kernel<1, 64, false>,
                          (const void*)stft_any_kernel<0, 256, false>, (const void*)stft_any_kernel<1, 256, false>,
                          (const void*)stft_any_kernel<0, 64, true>, (const void*)stft_any_kernel<1, 64, true>,
                          (const void*)stft_any_kernel<0, 256, true>, (const void*)stft_any_kernel<1, 256, true>};
    for (int i = 0; i < 8; ++i)
      if (hipFuncSetAttribute(kfn[i], hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) != hipSuccess) {
        g_hip_err = "hipFuncSetAttribute(stft_any_kernel) failed";
        mm_plan_destroy(p);
        return MM_ERR_HIP;
      }
    if (hipFuncSetAttribute((const void*)stft_anyb_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) != hipSuccess ||
        hipFuncSetAttribute((const void*)stft_anyb_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) != hipSuccess)
      p->any.fb = 0;
    p->any.ok = true;       // (the n_fft-specific set-up below does not apply; the trajectory rFFT set-up at the end does)
  }
  // n_fft 64 / 128 / 256 ride on the n_fft = 512 tile kernels: a frame zero-padded to 512 points around
  // its centre has X512[E*k] = (-1)^k X_nfft[k] (E = 512 / n_fft), i.e. the same power at every E-th
  // bin.  So the window is the Hann(win_length) centred in 512 and the mel weights sit at the bins E*k
  // (zero elsewhere); frame count, centre padding and everything downstream are unchanged.  (The
  // stage output mm_stft_power_f32 keeps the generic kernel: its rows have n_fft/2 + 1 bins.)
  mm_config cfg_e = *cfg;
  std::vector<float> win_e, mel_e;
  const mm_config* ce = cfg;
  const float* melp = mel.data();
  const float* winp = win.data();
  p->embed = 1;
  if (cfg->n_fft == 256 || cfg->n_fft == 128 || cfg->n_fft == 64) {
    const int E = 512 / cfg->n_fft;
    cfg_e.n_fft = 512;
    win_e.resize(512);
    mm::build_window(cfg_e, win_e.data());
    mel_e.assign((size_t)cfg->n_mels * 257, 0.0f);
    for (int m = 0; m < cfg->n_mels; ++m)
      for (int k = 0; k < p->n_bins; ++k) mel_e[(size_t)m * 257 + (size_t)E * k] = mel[(size_t)m * p->n_bins + k];
    if (upload(&p->d_window_e, win_e.data(), win_e.size() * 4) == MM_OK) {
      ce = &cfg_e; melp = mel_e.data(); winp = win_e.data(); p->embed = E;
    }
  }
  // a window that is zero outside samples [128, 384) of its 512-point frame (win_length <= 256, centred: the reference's
  // default 250; every n_fft <= 256 plan): the staged kernel reads and transforms a lane's middle eight pairs only
  p->s16_halfwin = 0;
  if (ce->n_fft == 512) {
    bool z = true;
    for (int i = 0; i < 128 && z; ++i) z = winp[i] == 0.0f && winp[511 - i] == 0.0f;
    p->s16_halfwin = z ? 1 : 0;
  }
  // register radix-16 path: n_fft 512; the 8-wave and the direct-load kernel need an even hop (8-byte
  // frame loads) and have no pre-emphasis: with an odd hop or pre-emphasis only the staged-sample
  // kernel applies (launch_stft sends the calls it cannot take to the generic kernel)
  mm::MelSweep sw;
  if (ce->n_fft == 512 && mm::build_mel_sweep(*ce, melp, 8, &sw)) {
    mm::MelRuns runs;
    mm::build_mel_runs(*ce, sw, 8, &runs);
    std::vector<float> tab(runs.hdr.size() + runs.grp.size());
    std::memcpy(tab.data(), runs.hdr.data(), runs.hdr.size() * 4);
    std::memcpy(tab.data() + runs.hdr.size(), runs.grp.data(), runs.grp.size() * 4);
    interleave_run_groups(tab.data() + runs.hdr.size(), runs.grp.size() / 8);
    p->sw_n_runs = (int)(runs.hdr.size() / 4);
    p->sw_n_tab16 = (int)(tab.size() / 4);
    p->lm_lds_bytes = (size_t)MM_LM_TAB_OFF + tab.size() * 4;
    // (a run table that does not fit beside the 8-wave kernel's tiles leaves n_fft 512 to the
    // wave-per-frame or the generic kernel; the set-up below this block still runs)
    if (p->lm_lds_bytes <= MM_LM_LDS_MAX) {
      if ((rc = upload(&p->d_sw_tab, tab.data(), tab.size() * 4)) ||
          (rc = upload(&p->d_sw_part, runs.part.data(), runs.part.size() * 4))) {
        mm_plan_destroy(p);
        return rc;
      }
      if (hipFuncSetAttribute((const void*)logmel512_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              MM_LM_LDS_MAX) == hipSuccess &&
          hipFuncSetAttribute((const void*)logmel512_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              MM_LM_LDS_MAX) == hipSuccess)   // the attribute is per function, not per plan:
                                                               // always the 160 KB maximum
        p->path = 1;
    }
    // 16-wave variant (4 waves per SIMD): needs its own 16-way mel partition and lane records
    mm::MelSweep sw16;
    if (p->path == 1 && mm::build_mel_sweep(*ce, melp, 16, &sw16)) {
      mm::MelRuns r16;
      mm::build_mel_runs(*ce, sw16, 16, &r16);
      std::vector<float> tab16(r16.hdr.size() + r16.grp.size());
      std::memcpy(tab16.data(), r16.hdr.data(), r16.hdr.size() * 4);
      std::memcpy(tab16.data() + r16.hdr.size(), r16.grp.data(), r16.grp.size() * 4);
      interleave_run_groups(tab16.data() + r16.hdr.size(), r16.grp.size() / 8);
      p->w16_n_runs = (int)(r16.hdr.size() / 4);
      p->w16_n_tab16 = (int)(tab16.size() / 4);
      p->w16_lds_bytes = (size_t)MM_W16_TAB_OFF + tab16.size() * 4;
      std::vector<float> lt(16 * MM_W16_LT_PITCH, 0.0f);
      for (int q = 0; q < 16; ++q) {
        float* r = lt.data() + q * MM_W16_LT_PITCH;
        for (int n1 = 0; n1 < 16; ++n1) { r[2 * n1] = winp[32 * n1 + 2 * q]; r[2 * n1 + 1] = winp[32 * n1 + 2 * q + 1]; }
        for (int k1 = 1; k1 < 16; ++k1) {
          const int idx = (q * k1) * (MM_TW_N / 256);
          r[32 + 2 * (k1 - 1)] = tw[2 * idx]; r[32 + 2 * (k1 - 1) + 1] = tw[2 * idx + 1];
        }
        for (int j = 0; j < 8; ++j) {
          const int idx = (q + 16 * j) * (MM_TW_N / 512);
          r[64 + 2 * j] = 0.5f * tw[2 * idx + 1]; r[64 + 2 * j + 1] = -0.5f * tw[2 * idx];
        }
      }
      if (p->w16_lds_bytes <= MM_LM_LDS_MAX &&
          upload(&p->d_w16_tab, tab16.data(), tab16.size() * 4) == MM_OK &&
          upload(&p->d_lane_tab, lt.data(), lt.size() * 4) == MM_OK &&
          upload(&p->d_w16_part, balance_parts_over_simds(r16).data(), r16.part.size() * 4) == MM_OK &&
          hipFuncSetAttribute((const void*)logmel512w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              MM_LM_LDS_MAX) == hipSuccess &&
          hipFuncSetAttribute((const void*)logmel512w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              MM_LM_LDS_MAX) == hipSuccess)
        p->w16_ok = 1;
      // 32-frame tiles, two 8-wave workgroups per CU (mm_logmel16h.hip.inc; opt-in variant 7): pair table from r16
      if (31 * cfg->hop_length + 512 + 256 <= MM_H16_NR * 2048 && (cfg->hop_length % 2) == 0 && cfg->preemph == 0.0f) {
        H16Tables ht;
        build_h16_tables(r16, &ht);
        p->h16_n_pairs = ht.n_pairs;
        p->h16_n_tab16 = (int)(ht.tab.size() / 4);
        p->h16_lds_bytes = (size_t)MM_H16_TAB_OFF + ht.tab.size() * 4;
        if (p->h16_lds_bytes <= 80 * 1024 &&
            upload(&p->d_h16_tab, ht.tab.data(), ht.tab.size() * 4) == MM_OK &&
            upload(&p->d_h16_part, ht.part.data(), ht.part.size() * 4) == MM_OK &&
            hipFuncSetAttribute((const void*)logmel512h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess)
          p->h16_ok = 1;
      }
      // staged-sample variant (mm_logmel16s.hip.inc): the tile's 63*hop + 512 samples must fit NR*4096 floats
      p->s16_nr = 0;
      if (p->w16_ok) {
        const int span = 63 * cfg->hop_length + 512;
        // 16-byte staging groups per thread: the fewest that hold the tile's samples (1 / 2 for short hops -- without
        // pre-emphasis only: those instantiations do not exist -- leave LDS for the log-mel tile of a 128-filter bank)
        int nr = span <= 4096 ? 1 : span <= 2 * 4096 ? 2 : span <= 3 * 4096 ? 3 : (span <= 4 * 4096 ? 4 : 0);
        if (nr && nr < 3 && cfg->preemph != 0.0f) nr = 3;
        const size_t lds = nr ? (size_t)MM_S16_TAB_OFF(nr) + tab16.size() * 4 : 0;
        const bool ok = nr && lds <= MM_LM_LDS_MAX && set_s16_attr(MM_LM_LDS_MAX);
        if (ok) { p->s16_nr = nr; p->s16_lds_bytes = lds; }
      }
      // staged-sample variant with the DCT fused in.  Two layouts of the log-mel tile Lt[filter][frame]:
      //   double buffered (up to ~64 filters): parts 3 / 7 / 11 / 15 of a weighted partition are half size and their waves
      //     compute one frame block's DCT each under phase B (10 MFMAs + 20 LDS reads ~ half a mel share at 40 mel);
      //   single (a 128-filter bank: 40 KB once, 80 KB twice does not fit beside the power tile): equal mel parts, waves
      //     0 .. 3 -- one per SIMD -- take the previous tile's DCT at the top of phase A.
      // Filters without a single weight (fmax above Nyquist: 26 of the reference default's 128) are handled analytically
      // (Logmel512Params::skip_empty): their runs are flagged, their A-operand columns zero, E[k] follows the A operands.
      if (p->s16_nr && cfg->n_mels <= 256) {
        const int lt_rows = (cfg->n_mels + 3) & ~3, kbn = (cfg->n_mfcc + 15) / 16;
        std::vector<char> empty(cfg->n_mels, 1);
        int n_empty = 0;
        for (int m = 0; m < cfg->n_mels; ++m) {
          for (int k = 0; k < 257 && empty[m]; ++k) if (melp[(size_t)m * 257 + k] != 0.0f) empty[m] = 0;
          n_empty += empty[m];
        }
        const bool skip = n_empty > 0 && n_empty < cfg->n_mels;
        // DCT steps (four filters each): the steps behind the last filter with a weight hold zero columns only -- the
        // reference default's 26 empty filters are its top six steps of 32 -- and are not run (the f32 matrix instruction
        // holds its SIMD's vector issue: every step is 32 cycles of the tile's time; measured 0.92 -> 0.90 ms)
        int nk = lt_rows / 4;
        if (skip) {
          int last = cfg->n_mels - 1;
          while (last > 0 && empty[last]) --last;
          nk = last / 4 + 1;
        }
        for (int layout = 0; layout < 2 && !p->s16f_ok; ++layout) {
          const bool single = layout == 1;
          const double wts_d[16] = {1, 1, 1, MM_S16F_W, 1, 1, 1, MM_S16F_W, 1, 1, 1, MM_S16F_W, 1, 1, 1, MM_S16F_W};
          double extra[16];
          for (int w = 0; w < 16; ++w) extra[w] = (!single && wts_d[w] < 1.0) ? 700.0 : 0.0;     // cost model units: instructions
          mm::MelSweep swf;
          if (!mm::build_mel_sweep(*ce, melp, 16, &swf, single ? nullptr : wts_d)) break;
          mm::MelRuns rf;
          mm::build_mel_runs(*ce, swf, 16, &rf);
          if (skip)
            for (size_t r = 0; r < rf.hdr.size() / 4; ++r) {
              const int d = rf.hdr[4 * r + 3];
              if (d >= 0 && d < cfg->n_mels && empty[d]) rf.hdr[4 * r + 0] |= 1 << 16;
            }
          std::vector<float> tabf(rf.hdr.size() + rf.grp.size());
          std::memcpy(tabf.data(), rf.hdr.data(), rf.hdr.size() * 4);
          std::memcpy(tabf.data() + rf.hdr.size(), rf.grp.data(), rf.grp.size() * 4);
          interleave_run_groups(tabf.data() + rf.hdr.size(), rf.grp.size() / 8);
          std::vector<int> wave_of_part;
          const std::vector<int> partf = balance_parts_over_simds(rf, extra, &wave_of_part);
          unsigned long long roles = ~0ull;
          for (int f = 0; f < 4; ++f) {
            const int w = single ? f : wave_of_part[4 * f + 3];       // single: waves 0 .. 3 sit on the four SIMDs
            roles = (roles & ~(0xFull << (4 * w))) | ((unsigned long long)f << (4 * w));
          }
          // A operands dct[16 kb + (l & 15)][4 s + (l >> 4)] (zero columns for skipped filters), then E[k] = the skipped
          // filters' column sum (float64 sum, rounded once)
          std::vector<float> dcta((size_t)kbn * nk * 64 + (size_t)kbn * 16 + 8, 0.0f);      // A | E | 256 "is empty" bits
          for (int kb = 0; kb < kbn; ++kb)
            for (int s = 0; s < nk; ++s)
              for (int l = 0; l < 64; ++l) {
                const int k = 16 * kb + (l & 15), m = 4 * s + (l >> 4);
                if (k < cfg->n_mfcc && m < cfg->n_mels && !(skip && empty[m]))
                  dcta[((size_t)kb * nk + s) * 64 + l] = dct[(size_t)k * cfg->n_mels + m];
              }
          if (skip)
            for (int k = 0; k < cfg->n_mfcc; ++k) {
              double e = 0.0;
              for (int m = 0; m < cfg->n_mels; ++m) if (empty[m]) e += (double)dct[(size_t)k * cfg->n_mels + m];
              dcta[(size_t)kbn * nk * 64 + k] = (float)e;
            }
          if (skip) {
            unsigned bits[8] = {0};
            for (int m = 0; m < cfg->n_mels; ++m) if (empty[m]) bits[m >> 5] |= 1u << (m & 31);
            std::memcpy(&dcta[(size_t)kbn * nk * 64 + (size_t)kbn * 16], bits, sizeof(bits));
          }
          const size_t tab_end = (size_t)MM_S16_TAB_OFF(p->s16_nr) + tabf.size() * 4;
          p->s16f_lt_off = (unsigned)align_up(tab_end, 16);
          p->s16f_dcta_off = p->s16f_lt_off + (single ? 1u : 2u) * (unsigned)lt_rows * 320u;
          p->s16f_red_off = (unsigned)align_up((size_t)p->s16f_dcta_off + dcta.size() * 4, 16);
          p->s16f_lds_bytes = (size_t)p->s16f_red_off + 2 * 16 * 8;
          if (p->s16f_lds_bytes > MM_LM_LDS_MAX) continue;          // try the single-tile layout
          if (upload(&p->d_s16f_tab, tabf.data(), tabf.size() * 4) == MM_OK &&
              upload(&p->d_s16f_part, partf.data(), partf.size() * 4) == MM_OK &&
              upload(&p->d_s16f_dcta, dcta.data(), dcta.size() * 4) == MM_OK) {
            p->s16f_n_runs = (int)(rf.hdr.size() / 4); p->s16f_n_tab16 = (int)(tabf.size() / 4);
            p->s16f_lt_rows = lt_rows; p->s16f_nk = nk; p->s16f_kb = kbn; p->s16f_roles = roles;
            p->s16f_flags = (single ? MM_S16F_SINGLE : 0) | (skip ? MM_S16F_SKIP : 0);
            p->s16f_ok = 1;
          }
          break;
        }
      }
      // 12-wave MFMA-mel variant (mm_logmel12m.hip.inc): banded A-operand table, unit lists, LDS budget
      if (p->w16_ok) {
        M12Tables mt;
        const int s_floats = (47 * cfg->hop_length + 512 + 255) & ~255;
        const int nr = (s_floats / 256 + 15) / 16;     // 1 KiB pieces (or 4 KiB of register-staged groups) per wave
        if (nr <= 3 && build_m12_tables(*ce, melp, dct.data(), &mt)) {
          p->m12_nb = mt.nb; p->m12_nr = nr; p->m12_s_floats = s_floats;
          p->m12_fused_dct = cfg->n_mfcc <= 16;
          if (mt.a_tab.empty()) mt.a_tab.assign(128, 0.0f);
          p->m12_win_off = (unsigned)(MM_M12_S_OFF + (size_t)s_floats * 4);
          p->m12_tw_off = p->m12_win_off + 2048u;
          p->m12_a_off = p->m12_tw_off + 16u * MM_M12_TW_PITCH * 4u;
          p->m12_n_a2 = (int)(mt.a_tab.size() / 2);
          p->m12_dct_off = p->m12_a_off + (unsigned)align_up(mt.a_tab.size() * 4, 16);
          p->m12_part_off = p->m12_dct_off + (unsigned)mt.nb * 1024u;
          p->m12_cnt_off = p->m12_part_off + (p->m12_fused_dct ? (unsigned)mt.n_slots * 1024u : 0u);
          p->m12_lds_bytes = (size_t)p->m12_cnt_off + 16;
          const std::vector<float> zeros(64, 0.0f);
          if (p->m12_lds_bytes <= MM_LM_LDS_MAX &&
              upload(&p->d_m12_a, mt.a_tab.data(), mt.a_tab.size() * 4) == MM_OK &&
              upload(&p->d_m12_dct, mt.dct_tab.data(), mt.dct_tab.size() * 4) == MM_OK &&
              upload(&p->d_zeros, zeros.data(), zeros.size() * 4) == MM_OK &&
              set_m12_attr(MM_LM_LDS_MAX)) {
            std::memcpy(p->m12_units, mt.units.data(), sizeof(p->m12_units));
            std::memcpy(p->m12_nunits, mt.n_units.data(), sizeof(p->m12_nunits));
            p->m12_ok = 1;
          }
        }
      }
    }
  }
  // wave-per-frame-group kernel (n_fft = 512*R, R = 1, 2, 4): the mel
  // sweep must advance by at most one filter between consecutive bins of a lane's 16-bin slice
  if ((cfg->n_fft == 512 || cfg->n_fft == 1024 || cfg->n_fft == 2048) && cfg->n_mels <= MM_WPF_MAXMEL) {
    const int R = cfg->n_fft / 512, L = 16 * R, NC = 256 * R;
    mm::MelSweep sw2;
    if (mm::build_mel_sweep(*cfg, mel.data(), 1, &sw2)) {
      // the highest bin any filter weighs: below NC / 2 the n_fft 2048 kernel forms no mirror bins, only the split pairs up to
      // it (NI = 4 .. 7), and its lanes sweep slices of EIGHT bins (64 lanes x 8 = bins 0 .. 511) instead of sixteen
      {
        int k_hi = 0;
        for (int m = 0; m < cfg->n_mels; ++m)
          for (int k = p->n_bins - 1; k > k_hi; --k)
            if (mel[(size_t)m * p->n_bins + k] != 0.0f) { k_hi = k; break; }
        p->wpf_half = (R >= 2 && k_hi < NC / 2 && cfg->preemph == 0.0f) ? 1 : 0;
        p->wpf_pairs = p->wpf_half ? std::max(4, k_hi / L + 1) : 8;
        if (p->wpf_pairs > 7) p->wpf_half = 0;
      }
      std::vector<float> ml;
      int group_max = 0;
      bool ok = true;
      for (int attempt = 0; attempt < 2; ++attempt) {
      const int SL = p->wpf_half ? 8 : 16;                 // bins per lane of the sweep
      ml.assign((size_t)L * 36, 0.0f);
      std::vector<int> d_end(L, -2);
      ok = true;
      for (int l = 0; l < L && ok; ++l) {
        float* r = ml.data() + l * 36;
        int dprev = sw2.d[SL * l];
        const int dstart = dprev;
        unsigned bits = 0;
        const int nslots = SL == 8 ? 8 : ((l == L - 1) ? 17 : 16);
        for (int i = 0; i < nslots; ++i) {
          const int k = (i < SL) ? SL * l + i : NC;
          const int adv = sw2.d[k] - dprev;
          if (adv < 0 || adv > 1) { ok = false; break; }
          if (adv == 1) bits |= (1u << i);
          dprev = sw2.d[k];
          r[i] = sw2.wlo[k];
          r[17 + i] = sw2.whi[k];
        }
        std::memcpy(&r[34], &dstart, 4);
        std::memcpy(&r[35], &bits, 4);
        d_end[l] = dprev;
      }
      // lanes whose sweep ends in the same run form a contiguous group: distance to its first lane in
      // bits 20..23 of the flag word, "last lane of the group" in bit 24 (the kernel pre-sums a group
      // in registers, at most 16 lanes)
      // (lanes whose remaining weights are all zero -- bins above fmax -- take no part)
      std::vector<char> act(L, 0);
      for (int l = 0; l < L; ++l) {
        const float* r = ml.data() + (size_t)l * 36;
        unsigned bits;
        std::memcpy(&bits, &r[35], 4);
        int from = 0;
        for (int i = 0; i < 17; ++i) if ((bits >> i) & 1u) from = i;
        for (int i = from; i < 17; ++i) if (r[i] != 0.0f || r[17 + i] != 0.0f) act[l] = 1;
      }
      group_max = 0;
      for (int l = 0; l < L && ok; ++l) {
        if (!act[l]) continue;
        int first = l;
        while (first > 0 && act[first - 1] && d_end[first - 1] == d_end[l]) --first;
        const int dist = l - first;
        const bool last = (l == L - 1) || !act[l + 1] || d_end[l + 1] != d_end[l];
        if (dist > 15) { ok = false; break; }     // (four bits: the last filter's run -- rising and falling part, no successor -- of a
                                                  //  40-filter bank at n_fft 2048 spans nine lanes)
        group_max = std::max(group_max, dist);
        unsigned bits;
        std::memcpy(&bits, &ml[(size_t)l * 36 + 35], 4);
        bits |= (unsigned)dist << 20;
        if (last) bits |= 1u << 24;
        std::memcpy(&ml[(size_t)l * 36 + 35], &bits, 4);
      }
      if (ok || !p->wpf_half) break;
      p->wpf_half = 0; p->wpf_pairs = 8;                   // eight-bin slices do not fit this bank: the sixteen-bin tables
      }
      p->wpf_group_max = group_max;
      std::vector<float> lt = wpf_lane_table(R, win.data(), tw.data());
      const int macc_stride = (cfg->n_mels + 2 + 63) / 64 * 64;      // slots -1 .. n_mels: no bounds tests in the sweep
      const int F = 4 / R;
      const int xbuf = (R == 1) ? 1280 : 1152;
      const int pbuf = F * (NC + NC / 16 + 4);
      const size_t wave_bytes = (size_t)(xbuf + pbuf + F * 3 * macc_stride) * 4;
      p->wpf_r = R;
      // the half-band instantiations (NI < 8) need ~120 registers and a short power row: sixteen waves where the LDS has the
      // room -- THEIR launch geometry only: the power stage and every other instantiation keep the twelve-wave layout
      p->wpf_waves_half = 0; p->wpf_lds_half = 0;
      if (p->wpf_half) {
        const size_t wb = (size_t)(xbuf + MM_WPF_PBUF_HALF + F * 3 * macc_stride) * 4;
        int wv = 16;
        while (wv > 4 && (size_t)L * MM_WPF_LT_PITCH * 4 + wv * wb > MM_LM_LDS_MAX) wv -= 4;
        p->wpf_waves_half = wv; p->wpf_lds_half = (size_t)L * MM_WPF_LT_PITCH * 4 + wv * wb;
      }
      p->wpf_waves = 12;
      while (p->wpf_waves > 4 && (size_t)L * MM_WPF_LT_PITCH * 4 + p->wpf_waves * wave_bytes > MM_LM_LDS_MAX) p->wpf_waves -= 4;
      p->wpf_lds_bytes = (size_t)L * MM_WPF_LT_PITCH * 4 + p->wpf_waves * wave_bytes;
      // sixteen waves (W16 instantiation: mel weights in LDS, power row over the exchange buffer), n_fft 1024 / 2048 log-mel mode
      p->wpf_lds16 = (size_t)L * (MM_WPF_LT_PITCH + MM_WPF_ML_PITCH) * 4 + 16 * (size_t)(xbuf + F * 3 * macc_stride) * 4;
      p->wpf_w16 = R >= 2 && p->wpf_lds16 <= MM_LM_LDS_MAX &&
                   hipFuncSetAttribute((const void*)logmel_wpf_kernel<2, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) == hipSuccess &&
                   hipFuncSetAttribute((const void*)logmel_wpf_kernel<4, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) == hipSuccess;
      const void* kfn[12] = {(const void*)logmel_wpf_kernel<1, 0, false>, (const void*)logmel_wpf_kernel<1, 1, false>,
                             (const void*)logmel_wpf_kernel<2, 0, false>, (const void*)logmel_wpf_kernel<2, 1, false>,
                             (const void*)logmel_wpf_kernel<4, 0, false>, (const void*)logmel_wpf_kernel<4, 1, false>,
                             (const void*)logmel_wpf_kernel<1, 0, true>, (const void*)logmel_wpf_kernel<1, 1, true>,
                             (const void*)logmel_wpf_kernel<2, 0, true>, (const void*)logmel_wpf_kernel<2, 1, true>,
                             (const void*)logmel_wpf_kernel<4, 0, true>, (const void*)logmel_wpf_kernel<4, 1, true>};
      bool attr_ok = p->wpf_lds_bytes <= MM_LM_LDS_MAX;
      for (int i = 0; i < 12 && attr_ok; ++i)
        attr_ok = hipFuncSetAttribute(kfn[i], hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) == hipSuccess;
      {
        if (p->wpf_half && p->wpf_pairs <= 7) {
#define MM_WPF_NFN(RR) (const void*)logmel_wpf_kernel<RR, 1, false, false, 0, 4>, (const void*)logmel_wpf_kernel<RR, 1, false, false, 0, 5>, \
                       (const void*)logmel_wpf_kernel<RR, 1, false, false, 0, 6>, (const void*)logmel_wpf_kernel<RR, 1, false, false, 0, 7>, \
                       (const void*)logmel_wpf_kernel<RR, 1, false, false, 3, 4>, (const void*)logmel_wpf_kernel<RR, 1, false, false, 3, 5>, \
                       (const void*)logmel_wpf_kernel<RR, 1, false, false, 3, 6>, (const void*)logmel_wpf_kernel<RR, 1, false, false, 3, 7>
          const void* nfn[16] = {MM_WPF_NFN(4), MM_WPF_NFN(2)};
#undef MM_WPF_NFN
          for (int i = 0; i < 16; ++i)
            if (hipFuncSetAttribute(nfn[i], hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) != hipSuccess) attr_ok = false;   // (the tables are the eight-bin ones)
        }
      }
      // a window that leaves the first and last Z / 16 of the frame zero (centred): the Z instantiations skip those pairs' loads,
      // products and additions -- 3 (win_length <= 0.625 n_fft: BASELINE configs[3], 1200 in 2048), and 5 / 6 / 7 for the
      // zero-padded frames the reference's dialog produces (n_fft typed, winLen 25 ms: 250 samples in 1024 -> 6, in 2048 -> 7)
      p->wpf_z = 0;
      if (R >= 2) {
        int zmax = 0;
        for (int zz = 1; zz <= 7; ++zz) {
          bool z = true;
          for (int i = 2 * L * (zz - 1); i < 2 * L * zz && z; ++i) z = win[i] == 0.0f && win[cfg->n_fft - 1 - i] == 0.0f;
          if (!z) break;
          zmax = zz;
        }
        const int zsel = zmax >= 7 ? 7 : zmax >= 6 ? 6 : zmax >= 5 ? 5 : zmax >= 3 ? 3 : 0;
        const void* zfn[8] = {(const void*)logmel_wpf_kernel<2, 1, false, false, 3>, (const void*)logmel_wpf_kernel<4, 1, false, false, 3>,
                              (const void*)logmel_wpf_kernel<2, 1, false, false, 5>, (const void*)logmel_wpf_kernel<4, 1, false, false, 5>,
                              (const void*)logmel_wpf_kernel<2, 1, false, false, 6>, (const void*)logmel_wpf_kernel<4, 1, false, false, 6>,
                              (const void*)logmel_wpf_kernel<2, 1, false, false, 7>, (const void*)logmel_wpf_kernel<4, 1, false, false, 7>};
        bool zok = zsel > 0;
        for (int i = 0; i < 8 && zok; ++i)
          zok = hipFuncSetAttribute(zfn[i], hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) == hipSuccess;
        if (zok) p->wpf_z = zsel;
      }
      if (ok && attr_ok && upload(&p->d_k2_lane_tab, lt.data(), lt.size() * 4) == MM_OK &&
          upload(&p->d_k2_mel_lane, ml.data(), ml.size() * 4) == MM_OK &&
          set_dct_fm_attr(64 * (MM_WPF_MAXMEL + 1) * 4))
        p->k2_ok = 1;
      if (p->k2_ok) {      // A operands of the matrix-pipe clamp + DCT kernel: dct[16 kb + (l & 15)][4 s + (l >> 4)]
        const int nk = (cfg->n_mels + 3) / 4, kbn = (cfg->n_mfcc + 15) / 16;
        const size_t lds = ((size_t)64 * ((4 * nk) | 1) + (size_t)kbn * nk * 64) * 4;
        if (lds <= 65536 && cfg->n_mels <= 128) {      // the kernel's loader holds 64 frames x 128 filters in registers
          std::vector<float> da((size_t)kbn * nk * 64, 0.0f);
          for (int kb = 0; kb < kbn; ++kb)
            for (int s2 = 0; s2 < nk; ++s2)
              for (int l = 0; l < 64; ++l) {
                const int k = 16 * kb + (l & 15), m = 4 * s2 + (l >> 4);
                if (k < cfg->n_mfcc && m < cfg->n_mels) da[((size_t)kb * nk + s2) * 64 + l] = dct[(size_t)k * cfg->n_mels + m];
              }
          if (upload(&p->d_dctfm_a, da.data(), da.size() * 4) == MM_OK) { p->dctfm_nk = nk; p->dctfm_kb = kbn; p->dctfm_lds = lds; }
          // wave-per-tile kernel: batch of 10 or 8 steps, whichever pads less
          const int ch = ((nk + 9) / 10 * 10 <= (nk + 7) / 8 * 8) ? 10 : 8, nkp = (nk + ch - 1) / ch * ch;
          std::vector<float> dw((size_t)kbn * nkp * 64, 0.0f);
          for (int kb = 0; kb < kbn; ++kb)
            std::memcpy(&dw[(size_t)kb * nkp * 64], &da[(size_t)kb * nk * 64], (size_t)nk * 64 * 4);
          const size_t ldsw = ((size_t)kbn * nkp * 64 + 4 * 16 * (size_t)((4 * nkp) | 1)) * 4;
          if (p->d_dctfm_a && ldsw <= 65536 && upload(&p->d_dctw_a, dw.data(), dw.size() * 4) == MM_OK) { p->dctw_nk = nkp; p->dctw_ch = ch; }
        }
      }
    }
  }
  if (hipFuncSetAttribute((const void*)rfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LM_LDS_MAX) !=
      hipSuccess) {
    g_hip_err = "hipFuncSetAttribute(rfft_generic_kernel) failed";
    mm_plan_destroy(p);
    return MM_ERR_HIP;
  }
  {
    const std::vector<float> lt = wpf_lane_table(4, nullptr, tw.data());
    if (upload(&p->d_rf2k_lane_tab, lt.data(), lt.size() * 4) == MM_OK &&
        hipFuncSetAttribute((const void*)rfft_wpf_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            MM_LM_LDS_MAX) == hipSuccess &&
        hipFuncSetAttribute((const void*)rfft_wpf_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            MM_LM_LDS_MAX) == hipSuccess)
      p->rf2k_ok = 1;
  }
  *out = p;
  return MM_OK;
}

int mm_plan_destroy(mm_plan* p) {
  if (!p) return MM_OK;
  (void)hipFree(p->d_window); (void)hipFree(p->d_tw); (void)hipFree(p->d_mel_start);
  (void)hipFree(p->d_mel_len); (void)hipFree(p->d_mel_off); (void)hipFree(p->d_mel_w);
  (void)hipFree(p->d_dct_t);
  (void)hipFree(p->d_sw_tab); (void)hipFree(p->d_sw_part);
  (void)hipFree(p->d_w16_tab); (void)hipFree(p->d_lane_tab); (void)hipFree(p->d_w16_part);

  (void)hipFree(p->d_k2_lane_tab); (void)hipFree(p->d_k2_mel_lane); (void)hipFree(p->d_window_e);
  (void)hipFree(p->d_rf2k_lane_tab);
  (void)hipFree(p->d_m12_a); (void)hipFree(p->d_m12_dct); (void)hipFree(p->d_zeros);
  (void)hipFree(p->d_dctfm_a); (void)hipFree(p->d_dctw_a);
  (void)hipFree(p->d_s16f_tab); (void)hipFree(p->d_s16f_dcta); (void)hipFree(p->d_s16f_part);
  (void)hipFree(p->d_h16_tab); (void)hipFree(p->d_h16_part);
  (void)hipFree(p->any.d_tw); (void)hipFree(p->any.d_split); (void)hipFree(p->any.d_chirp); (void)hipFree(p->any.d_bhat);
  (void)hipFree(p->any.d_tabpack);
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  delete p;
  return MM_OK;
}

int mm_plan_config(const mm_plan* p, mm_config* out) {
  if (!p || !out) return MM_ERR_INVALID_ARG;
  *out = p->cfg;
  return MM_OK;
}

// Which fused kernel a log-mel / MFCC call runs on (mode 1; mode 0 = the power stage output).  `call`
// = false answers for a regular call (aligned rows, n_samples >= 4).  p->variant (mm_plan_set_variant)
// pins a variant where it applies; what a variant cannot take falls through to the next one.
enum { MM_K_GENERIC = 0, MM_K_W8 = 1, MM_K_W16 = 2, MM_K_WPF = 3, MM_K_W16S = 4, MM_K_M12 = 5, MM_K_ANY = 6, MM_K_H16 = 7 };
static int choose_kernel(const mm_plan* p, int mode, bool call, const float* d_audio, int64_t n_samples, int64_t stride) {
  if (p->any.ok) return MM_K_ANY;
  if (p->force_generic) return MM_K_GENERIC;
  const int v = p->variant;
  const bool force_wpf = v == MM_K_WPF;
  const bool n4 = !call || n_samples >= 4, n2 = !call || n_samples >= 2;
  // (the matrix-pipe variant is opt-in: on gfx950 v_mfma_f32_16x16x4_f32 holds the SIMD's VALU issue for its
  // whole 32 cycles -- tools/probe/mfma_f32_coexec.hip -- so the mel MFMAs do not run under the transforms
  // and the kernel measures 0.44 ms where the run-table kernel takes 0.37 ms)
  // (the 32-frame-tile / two-workgroup experiment: opt-in, log-mel mode, plain aligned rows, length a multiple of 4)
  if (v == MM_K_H16 && p->h16_ok && mode == 1 &&
      (!call || (n_samples >= 4 && (n_samples % 4) == 0 && (stride % 4) == 0 && (((uintptr_t)d_audio) & 15) == 0)))
    return MM_K_H16;
  const bool m12_ok = p->m12_ok && mode == 1 && n4 && v == MM_K_M12;
  const bool staged_ok = p->s16_nr && p->w16_ok && n4 && v != MM_K_W16 && v != MM_K_W8;
  const bool direct_ok = (!call || ((stride % 2) == 0 && (((uintptr_t)d_audio) & 7) == 0)) && n2 &&
                         p->cfg.preemph == 0.0f && (p->cfg.hop_length % 2) == 0;
  const bool tile_ok = p->path == 1 && (direct_ok || staged_ok || m12_ok) && (p->embed == 1 || mode != 0);
  if (p->k2_ok && (p->cfg.n_fft != 512 || force_wpf || !tile_ok) && n2) return MM_K_WPF;
  if (!tile_ok) return MM_K_GENERIC;
  if (m12_ok) return MM_K_M12;
  if (p->w16_ok && v != MM_K_W8) return staged_ok ? MM_K_W16S : MM_K_W16;
  return direct_ok ? MM_K_W8 : MM_K_GENERIC;
}

int mm_plan_kernel_path(const mm_plan* p) {
  if (!p) return MM_ERR_INVALID_ARG;
  return choose_kernel(p, 1, false, nullptr, 0, 0);
}

int mm_plan_fused_dct(const mm_plan* p) {
  if (!p) return MM_ERR_INVALID_ARG;
  const int k = choose_kernel(p, 1, false, nullptr, 0, 0);
  return !p->no_fuse && ((k == MM_K_M12 && p->m12_fused_dct) || (k == MM_K_W16S && p->s16f_ok)) ? 1 : 0;
}

int mm_plan_set_fuse_dct(mm_plan* p, int on) {
  if (!p) return MM_ERR_INVALID_ARG;
  const int prev = !p->no_fuse;
  p->no_fuse = on ? 0 : 1;
  return prev;
}

int mm_plan_set_fuse_tail(mm_plan* p, int on) {
  if (!p) return MM_ERR_INVALID_ARG;
  const int prev = p->no_fuse_tail ? 0 : (p->fuse_tail_wide ? 2 : 1);
  p->no_fuse_tail = on ? 0 : 1;
  p->fuse_tail_wide = on == 2 ? 1 : 0;
  return prev;
}

int mm_plan_set_variant(mm_plan* p, int variant) {
  if (!p || variant < 0 || variant > MM_K_H16 || variant == MM_K_ANY) return MM_ERR_INVALID_ARG;   // (MM_K_ANY is not a choice: such plans have one kernel)
  // (on an any-length plan the value selects that kernel's form instead -- 0 automatic, 1 one frame per wave, 2 .. 4
  // frames per wave at once: development A/B, launch_stft)
  const int prev = p->variant;
  p->variant = variant;
  return prev;
}

int mm_plan_force_generic(mm_plan* p, int on) {
  if (!p) return MM_ERR_INVALID_ARG;
  int prev = p->force_generic;
  p->force_generic = on ? 1 : 0;
  return prev;
}

size_t mm_workspace_bytes(const mm_plan* p, int64_t batch, int64_t n_samples) {
  if (!p || batch < 1 || n_samples < 0) return 0;
  const int64_t T = mm_num_frames(&p->cfg, n_samples);
  return align_up((size_t)batch * p->cfg.n_mels * T * 4, 256) + align_up((size_t)batch * 8, 256);   // log-mel rows | max keys | -min keys
}

// frame_major: the caller accepts (and, where the n_fft = 2048 kernel runs, gets) log-mel rows laid out
// [B][T][n_mels]; *is_fm reports which layout was written.
struct StftOut {
  float* power = nullptr;      // mode 0: [B][T][n_bins]
  float* logmel = nullptr;     // mode 1: log-mel rows (may be null on the fused-DCT kernel when top_db < 0)
  int* key_max = nullptr;      // [B]
  int* key_nmin = nullptr;     // [B] or null
  float* mfcc = nullptr;       // fused (unclamped) DCT output, MM_K_M12 only
  bool frame_major = false;    // the caller accepts log-mel rows laid out [B][T][n_mels]
  bool is_fm = false;          // out: that layout was written
  bool fused_dct = false;      // out: mfcc holds the unclamped DCT
  float* mod = nullptr;        // in: modulation-spectrum output wanted from the same launch (clip mode)
  bool clip_only = false;      // in: clip mode wanted even without a modulation spectrum (fix-up / empty filters' add in the launch)
  int n_mod = 0;
  bool fused_tail = false;     // out: clamp fix-up and trajectory rFFT were part of the launch (no keys used)
  bool skip_empty = false;     // out: filters without weights were handled analytically (rows not stored, E[k] L0 in the DCT)
};

// Clip mode of the staged-sample kernel (whole clips per workgroup, tail fused in): the trajectory length must be
// one the in-kernel rFFT covers, and the clips must spread evenly -- a workgroup that gets one clip more than the
// others sets the launch time, so the uneven case stays on the tile-granular split + separate launches.
static bool s16_clip_mode_ok(const mm_plan* p, int64_t batch, int n_mod) {
  // n_mod 512 / 1024: the default.  Opt-in (mm_plan_set_fuse_tail(plan, 2)): n_mod 2048 (1025 .. 2048 frames per clip) and
  // n_mod 0 (no modulation spectrum: mm_mfcc_f32 on a plan with empty filters, whose add would otherwise be a launch).
  // Measured on the reference's default call as a batch (1024 x 2001 frames, 128 mel, 26 of them empty): one launch
  // 0.97 ms against 1.05 ms when no clip clamps -- but a clamping clip's fix-up (a pass over ITS log-mel rows from HBM,
  // ~40 us on one workgroup while the others wait) makes the launch as slow as its unluckiest workgroup: with one clip
  // in ten clamping (bench.py's noise + tone signal through 128 narrow filters) 1.10 against 1.05 ms.  The separate
  // fix-up launch spreads those clips over the whole chip, so it stays the default for these two cases.
  if (p->no_fuse_tail || !(n_mod == 512 || n_mod == 1024 || (p->fuse_tail_wide && (n_mod == 0 || (n_mod == 2048 && p->rf2k_ok))))) return false;
  const int64_t g = p->num_cus;
  if (batch < g) return false;
  const int64_t per = (batch + g - 1) / g;
  if (per > MM_S16_CPW_MAX) return false;                                          // extreme slots [per][16] in LDS
  if ((size_t)p->s16f_red_off + (size_t)per * 128 > MM_LM_LDS_MAX) return false;
  if ((size_t)MM_S16_DELTA_OFF((size_t)p->s16f_red_off, (size_t)per, n_mod) + (size_t)per * 4 > MM_LM_LDS_MAX) return false;
  return per * g * 100 <= batch * 104;          // at most 4 % of idle workgroup time
}

static int launch_stft(mm_plan* p, int mode, const float* d_audio, int64_t batch, int64_t n_samples,
                       int64_t stride, StftOut& o, hipStream_t st) {
  o.is_fm = false; o.fused_dct = false;
  const int kern = choose_kernel(p, mode, true, d_audio, n_samples, stride);
  if (kern == MM_K_ANY) {
    const AnyPlan& ap = p->any;
    AnyParams q;
    q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
    q.n_frames = mm_num_frames(&p->cfg, n_samples);
    q.n_fft = p->cfg.n_fft; q.hop = p->cfg.hop_length; q.n_bins = p->n_bins; q.n_mels = p->cfg.n_mels;
    q.preemph = p->cfg.preemph; q.amin = p->cfg.amin; q.db_offset = p->db_offset; q.window = p->d_window;
    q.nn = ap.nn; q.packed = ap.packed; q.n_pass = ap.n_pass; std::memcpy(q.radix, ap.radix, sizeof(q.radix));
    q.M = ap.M; q.log2M = ap.log2M; q.tw = ap.d_tw; q.split = ap.d_split; q.chirp = ap.d_chirp; q.bhat = ap.d_bhat;
    q.mel_start = p->d_mel_start; q.mel_len = p->d_mel_len; q.mel_off = p->d_mel_off; q.mel_w = p->d_mel_w;
    q.out_power = o.power; q.out_logmel = o.logmel; q.clip_key = o.key_max;
    // with the tables in LDS a thread group takes 16 consecutive frames (the copy is paid once per 64 / 16 frames)
    q.frames_per_group = ap.lds_tab ? 16 : 4; q.grp_bytes = ap.grp_bytes; q.b_off = ap.b_off; q.p_off = ap.p_off;
    q.tabpack = ap.d_tabpack; q.tab_floats = ap.tab_floats; q.o_tw = ap.o_tw; q.o_split = ap.o_split; q.o_chirp = ap.o_chirp;
    q.o_melw = ap.o_melw; q.o_mstart = ap.o_mstart; q.o_mlen = ap.o_mlen; q.o_moff = ap.o_moff;
    // p->variant on an any-length plan: 1 = the one-frame-per-wave kernel (A/B), 2 .. 4 = frames per batch
    // (where the two-stage register kernel is the default, 2 keeps the batched LDS kernel and 1 the one-frame kernel)
    if (ap.reg2 && p->variant == 0) {
      q.frames_per_group = reg2_frames_per_wave(ap.reg2);
      const int fpbb = 4 * q.frames_per_group;
      const int64_t gridb = batch * ((q.n_frames + fpbb - 1) / fpbb);
      if (gridb > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
      reg2_launch(ap.reg2, ap.reg2_r2, mode, dim3((unsigned)gridb), reg2_lds_bytes(ap.reg2, ap.reg2_r2, ap.tab_floats), st, q);
      HIP_TRY(hipGetLastError());
      return MM_OK;
    }
    const int fb = p->variant == 1 ? 0 : (p->variant >= 2 && p->variant <= 4 && ap.fb ? p->variant : ap.fb);
    if (fb > 0) {
      q.frames_per_group = 16 / fb * fb;                  // frames a wave walks (in batches of fb)
      q.grp_bytes = (unsigned)((size_t)2 * fb * ap.nn * 8);
      const int fpbb = 4 * q.frames_per_group;
      const int64_t gridb = batch * ((q.n_frames + fpbb - 1) / fpbb);
      if (gridb > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
      const size_t ldsb = (size_t)4 * q.grp_bytes + (size_t)ap.tab_floats * 4;
      if (mode == 0) hipLaunchKernelGGL(stft_anyb_kernel<0>, dim3((unsigned)gridb), dim3(256), ldsb, st, q, fb);
      else hipLaunchKernelGGL(stft_anyb_kernel<1>, dim3((unsigned)gridb), dim3(256), ldsb, st, q, fb);
      HIP_TRY(hipGetLastError());
      return MM_OK;
    }
    const int G = 256 / ap.tpf, fpb = G * q.frames_per_group;
    const int64_t grid = batch * ((q.n_frames + fpb - 1) / fpb);
    if (grid > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    const size_t lds = (size_t)G * ap.grp_bytes + (ap.lds_tab ? (size_t)ap.tab_floats * 4 : 0);
    const dim3 gd((unsigned)grid), bd(256);
#define MM_ANY_GO(MM, TT, LL) hipLaunchKernelGGL((stft_any_kernel<MM, TT, LL>), gd, bd, lds, st, q)
    if (ap.lds_tab) {
      if (ap.tpf == 64) { if (mode == 0) MM_ANY_GO(0, 64, true); else MM_ANY_GO(1, 64, true); }
      else { if (mode == 0) MM_ANY_GO(0, 256, true); else MM_ANY_GO(1, 256, true); }
    } else {
      if (ap.tpf == 64) { if (mode == 0) MM_ANY_GO(0, 64, false); else MM_ANY_GO(1, 64, false); }
      else { if (mode == 0) MM_ANY_GO(0, 256, false); else MM_ANY_GO(1, 256, false); }
    }
#undef MM_ANY_GO
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (kern == MM_K_H16) {
    Logmel512hParams q;
    q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
    q.n_frames = mm_num_frames(&p->cfg, n_samples);
    q.tiles_per_clip = (q.n_frames + 31) / 32;
    q.n_tiles = batch * q.tiles_per_clip;
    q.hop = p->cfg.hop_length; q.n_mels = p->cfg.n_mels; q.amin = p->cfg.amin; q.db_offset = p->db_offset;
    q.lane_tab = p->d_lane_tab; q.pair_tab = (const float4*)p->d_h16_tab; q.n_pairs = p->h16_n_pairs; q.n_tab16 = p->h16_n_tab16;
    q.wave_part = p->d_h16_part; q.out_logmel = o.logmel; q.clip_key = o.key_max; q.key_nmin = o.key_nmin;
    if (q.n_tiles > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    const int64_t grid = std::min<int64_t>(q.n_tiles, 2 * (int64_t)p->num_cus);
    hipLaunchKernelGGL(logmel512h_kernel, dim3((unsigned)grid), dim3(512), p->h16_lds_bytes, st, q);
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (kern == MM_K_WPF) {
    WpfParams q;
    const int R = p->wpf_r, F = 4 / R;
    q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
    q.n_frames = mm_num_frames(&p->cfg, n_samples);
    q.groups_per_clip = (q.n_frames + F - 1) / F;
    q.total_groups = batch * q.groups_per_clip;
    q.hop = p->cfg.hop_length; q.n_mels = p->cfg.n_mels; q.amin = p->cfg.amin; q.db_offset = p->db_offset;
    // The sixteen-wave form (W16) is built and correct (the GPU suite passes on it) but measured 1.5 % SLOWER than the
    // twelve-wave form on configs[3] (1.90 - 1.92 vs 1.875 - 1.90 ms: its LDS reads of the mel weights and three spilled
    // registers cost what the fourth wave per SIMD hides; docs/experiments.md R4) -- off unless a side build asks for it.
#ifdef MM_WPF_W16_DEFAULT
    const bool w16 = p->wpf_w16 && mode == 1 && p->cfg.preemph == 0.0f && p->variant != MM_K_WPF;
#else
    const bool w16 = false;
#endif
    const bool half = mode == 1 && p->cfg.preemph == 0.0f && R >= 2 && p->wpf_half && p->wpf_pairs >= 4 && p->wpf_pairs <= 7 && !w16;
    q.macc_stride = (p->cfg.n_mels + 2 + 63) / 64 * 64; q.waves_per_wg = w16 ? 16 : (half ? p->wpf_waves_half : p->wpf_waves);
    q.lane_tab = p->d_k2_lane_tab; q.mel_lane = p->d_k2_mel_lane; q.group_max = p->wpf_group_max;
    q.out_logmel = o.logmel; q.clip_key = o.key_max; q.out_power = o.power;
    if (o.frame_major) { q.sB = q.n_frames * q.n_mels; q.sT = q.n_mels; q.sM = 1; }
    else { q.sB = q.n_frames * q.n_mels; q.sT = 1; q.sM = q.n_frames; }
    o.is_fm = o.frame_major;
    int64_t grid = (q.total_groups + q.waves_per_wg - 1) / q.waves_per_wg;
    if (grid > p->num_cus) grid = p->num_cus;
    const dim3 blk(64 * q.waves_per_wg);
    const size_t lds = w16 ? p->wpf_lds16 : (half ? p->wpf_lds_half : p->wpf_lds_bytes);
    q.preemph = p->cfg.preemph;
    const bool pre = p->cfg.preemph != 0.0f;
#define MM_WPF_LAUNCH(RR, MM) do { if (pre) hipLaunchKernelGGL((logmel_wpf_kernel<RR, MM, true>), dim3((unsigned)grid), blk, lds, st, q); \
                                   else hipLaunchKernelGGL((logmel_wpf_kernel<RR, MM, false>), dim3((unsigned)grid), blk, lds, st, q); } while (0)
    if (w16) {
      if (R == 2) hipLaunchKernelGGL((logmel_wpf_kernel<2, 1, false, true>), dim3((unsigned)grid), blk, lds, st, q);
      else hipLaunchKernelGGL((logmel_wpf_kernel<4, 1, false, true>), dim3((unsigned)grid), blk, lds, st, q);
    } else if (half) {
      // n_fft 1024 / 2048, a mel bank that ends below sr / 4: the output-pruned instantiations (with or without the input pruning)
#define MM_WPF_NI(RR, ZZ, NN) hipLaunchKernelGGL((logmel_wpf_kernel<RR, 1, false, false, ZZ, NN>), dim3((unsigned)grid), blk, lds, st, q)
#define MM_WPF_NIS(RR, ZZ) switch (p->wpf_pairs) { case 4: MM_WPF_NI(RR, ZZ, 4); break; case 5: MM_WPF_NI(RR, ZZ, 5); break; \
                                                   case 6: MM_WPF_NI(RR, ZZ, 6); break; default: MM_WPF_NI(RR, ZZ, 7); }
      if (R == 4) { if (p->wpf_z >= 3) { MM_WPF_NIS(4, 3) } else { MM_WPF_NIS(4, 0) } }
      else { if (p->wpf_z >= 3) { MM_WPF_NIS(2, 3) } else { MM_WPF_NIS(2, 0) } }
#undef MM_WPF_NIS
#undef MM_WPF_NI
    } else if (p->wpf_z >= 3 && mode == 1 && !pre && R >= 2) {
#define MM_WPF_Z(RR, ZZ) hipLaunchKernelGGL((logmel_wpf_kernel<RR, 1, false, false, ZZ>), dim3((unsigned)grid), blk, lds, st, q)
#define MM_WPF_ZS(RR) switch (p->wpf_z) { case 7: MM_WPF_Z(RR, 7); break; case 6: MM_WPF_Z(RR, 6); break; case 5: MM_WPF_Z(RR, 5); break; \
                                          default: MM_WPF_Z(RR, 3); }
      if (R == 2) { MM_WPF_ZS(2) } else { MM_WPF_ZS(4) }
#undef MM_WPF_ZS
#undef MM_WPF_Z
    } else
    if (R == 1) { if (mode == 0) MM_WPF_LAUNCH(1, 0); else MM_WPF_LAUNCH(1, 1); }
    else if (R == 2) { if (mode == 0) MM_WPF_LAUNCH(2, 0); else MM_WPF_LAUNCH(2, 1); }
    else { if (mode == 0) MM_WPF_LAUNCH(4, 0); else MM_WPF_LAUNCH(4, 1); }
#undef MM_WPF_LAUNCH
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (kern == MM_K_M12) {
    Logmel12mParams q;
    q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
    q.n_frames = mm_num_frames(&p->cfg, n_samples);
    q.tiles_per_clip = (q.n_frames + MM_M12_TF - 1) / MM_M12_TF;
    q.n_tiles = batch * q.tiles_per_clip;
    q.hop = p->cfg.hop_length; q.n_mels = p->cfg.n_mels; q.n_mfcc = p->cfg.n_mfcc; q.nb = p->m12_nb;
    q.amin = p->cfg.amin; q.db_offset = p->db_offset; q.preemph = p->cfg.preemph;
    q.lane_tab = p->d_lane_tab; q.a_tab = (const float2*)p->d_m12_a; q.n_a2 = p->m12_n_a2;
    q.window = p->embed > 1 ? p->d_window_e : p->d_window;
    q.dct_tab = p->d_m12_dct;
    std::memcpy(q.units, p->m12_units, sizeof(q.units));
    std::memcpy(q.n_units, p->m12_nunits, sizeof(q.n_units));
    o.fused_dct = o.mfcc != nullptr && p->m12_fused_dct && !p->no_fuse;
    q.out_logmel = (o.fused_dct && p->cfg.top_db < 0.0f) ? nullptr : o.logmel;   // rows only feed the clamp fix-up
    q.out_mfcc = o.fused_dct ? o.mfcc : nullptr;
    q.key_max = o.key_max; q.key_nmin = o.key_nmin;
    q.s_floats = p->m12_s_floats; q.zeros = p->d_zeros;
    q.win_off = p->m12_win_off; q.tw_off = p->m12_tw_off; q.a_off = p->m12_a_off; q.dct_off = p->m12_dct_off; q.part_off = p->m12_part_off; q.cnt_off = p->m12_cnt_off;
    if (q.n_tiles > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    const int64_t grid = q.n_tiles < p->num_cus ? q.n_tiles : p->num_cus;
    const bool pre = p->cfg.preemph != 0.0f;
    const bool odd = p->cfg.hop_length & 1;
    const bool unal = (stride % 4) != 0 || (n_samples % 4) != 0 || (((uintptr_t)d_audio) & 15) != 0;
    launch_m12(p->m12_nr, pre, odd, unal, dim3((unsigned)grid), p->m12_lds_bytes, st, q);
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (kern != MM_K_GENERIC) {
    Logmel512Params q;
    q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
    q.n_frames = mm_num_frames(&p->cfg, n_samples);
    q.tiles_per_clip = (q.n_frames + 63) / 64;
    q.n_tiles = batch * q.tiles_per_clip;
    q.hop = p->cfg.hop_length; q.n_mels = p->cfg.n_mels; q.amin = p->cfg.amin; q.db_offset = p->db_offset;
    q.window = p->embed > 1 ? p->d_window_e : p->d_window; q.tw = p->d_tw; q.mel_tab = (const float4*)p->d_sw_tab; q.n_runs = p->sw_n_runs; q.n_tab16 = p->sw_n_tab16;
    q.wave_part = p->d_sw_part; q.out_logmel = o.logmel; q.clip_key = o.key_max;
    q.out_power = o.power;
    q.out_mfcc = nullptr; q.key_nmin = nullptr; q.dct_a = nullptr; q.n_mfcc = 0; q.dct_nk = q.dct_kb = q.lt_rows = 0;
    q.lt_off = q.dcta_off = 0; q.dct_roles = ~0ull;
    q.out_mod = nullptr; q.dct_t = nullptr; q.n_mod = q.dct_kp = 0; q.top_db = -1.0f; q.red_off = 0;
    q.dct_flags = p->s16_halfwin ? MM_S16F_HALFWIN : 0; q.lt_b2 = 0;
    q.lane_tab = p->d_lane_tab;
    q.preemph = p->cfg.preemph;
    if (q.n_tiles > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
    int64_t grid = q.n_tiles < p->num_cus ? q.n_tiles : p->num_cus;
    if (kern == MM_K_W16S || kern == MM_K_W16) {
      q.mel_tab = (const float4*)p->d_w16_tab; q.n_runs = p->w16_n_runs; q.n_tab16 = p->w16_n_tab16;
      q.wave_part = p->d_w16_part;
      if (kern == MM_K_W16S) {
        const bool pre = p->cfg.preemph != 0.0f;
        const bool odd = p->cfg.hop_length & 1;
        const bool unal = (stride % 4) != 0 || (n_samples % 4) != 0 || (((uintptr_t)d_audio) & 15) != 0;
        size_t lds = p->s16_lds_bytes;
        if (mode == 1 && o.mfcc != nullptr && p->s16f_ok && o.key_nmin != nullptr && !p->no_fuse) {
          // DCT fused in: its own run table (half-size parts for the four DCT waves)
          o.fused_dct = true;
          o.skip_empty = (p->s16f_flags & MM_S16F_SKIP) != 0;
          q.mel_tab = (const float4*)p->d_s16f_tab; q.n_runs = p->s16f_n_runs; q.n_tab16 = p->s16f_n_tab16;
          q.wave_part = p->d_s16f_part;
          q.out_mfcc = o.mfcc; q.key_nmin = o.key_nmin; q.dct_a = p->d_s16f_dcta; q.n_mfcc = p->cfg.n_mfcc;
          q.dct_nk = p->s16f_nk; q.dct_kb = p->s16f_kb; q.lt_rows = p->s16f_lt_rows;
          q.lt_off = p->s16f_lt_off; q.dcta_off = p->s16f_dcta_off; q.dct_roles = p->s16f_roles;
          q.dct_flags = p->s16f_flags | (p->s16_halfwin ? MM_S16F_HALFWIN : 0);
          q.lt_b2 = (p->s16f_flags & MM_S16F_SINGLE) ? 0 : p->s16f_lt_rows;
          if (p->cfg.top_db < 0.0f) q.out_logmel = nullptr;      // the rows only feed the clamp fix-up
          lds = p->s16f_lds_bytes;
          q.red_off = p->s16f_red_off;
          if ((o.mod != nullptr || o.clip_only) && s16_clip_mode_ok(p, batch, o.mod ? o.n_mod : 0)) {
            // whole clips per workgroup: extremes, clamp fix-up and trajectory rFFT inside the launch
            o.fused_tail = true;
            q.out_mod = (float2*)o.mod; q.n_mod = o.mod ? o.n_mod : 0; q.dct_t = p->d_dct_t; q.dct_kp = p->kp; q.top_db = p->cfg.top_db;
            grid = batch < p->num_cus ? batch : p->num_cus;
            const size_t per = (size_t)((batch + grid - 1) / grid);
            lds = std::max((size_t)p->s16f_red_off + per * 128, (size_t)MM_S16_FIN_TAB_OFF + MM_S16_FIN_TAB_BYTES);
            lds = std::max(lds, (size_t)MM_S16_DELTA_OFF((size_t)p->s16f_red_off, per, q.n_mod) + per * 4);
            if (q.n_mod == 2048) {          // the tail runs the 2048-point transform: its lane table rides in q.tw
              q.tw = (const float2*)p->d_rf2k_lane_tab;
              lds = std::max(lds, (size_t)MM_S16_FIN2K_BYTES);
            }
            mode = 2;
          }
        }
        launch_s16(mode, p->s16_nr, pre, odd, unal, dim3((unsigned)grid), lds, st, q);
        HIP_TRY(hipGetLastError());
        return MM_OK;
      }
      if (mode == 0)
        hipLaunchKernelGGL(logmel512w_kernel<0>, dim3((unsigned)grid), dim3(1024), p->w16_lds_bytes, st, q);
      else
        hipLaunchKernelGGL(logmel512w_kernel<1>, dim3((unsigned)grid), dim3(1024), p->w16_lds_bytes, st, q);
      HIP_TRY(hipGetLastError());
      return MM_OK;
    }
    if (mode == 0)
      hipLaunchKernelGGL(logmel512_kernel<0>, dim3((unsigned)grid), dim3(512), p->lm_lds_bytes, st, q);
    else
      hipLaunchKernelGGL(logmel512_kernel<1>, dim3((unsigned)grid), dim3(512), p->lm_lds_bytes, st, q);
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  StftParams q;
  q.audio = d_audio; q.batch = batch; q.n_samples = n_samples; q.stride = stride;
  q.n_frames = mm_num_frames(&p->cfg, n_samples);
  q.n_fft = p->cfg.n_fft; q.log2nc = p->log2nc; q.hop = p->cfg.hop_length; q.n_bins = p->n_bins;
  q.n_mels = p->cfg.n_mels; q.preemph = p->cfg.preemph; q.amin = p->cfg.amin; q.db_offset = p->db_offset;
  q.window = p->d_window; q.tw = p->d_tw; q.mel_start = p->d_mel_start; q.mel_len = p->d_mel_len;
  q.mel_off = p->d_mel_off; q.mel_w = p->d_mel_w; q.out_power = o.power; q.out_logmel = o.logmel;
  q.clip_key = o.key_max; q.frames_per_wave = 4;
  const int nc = 1 << p->log2nc;
  const size_t wave_bytes = ((size_t)nc * 8 + (size_t)(nc + 1) * 4 + 15) & ~(size_t)15;
  const int fpb = 4 * q.frames_per_wave;
  const int64_t tiles = (q.n_frames + fpb - 1) / fpb;
  const int64_t grid = batch * tiles;
  if (grid > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  if (mode == 0)
    hipLaunchKernelGGL(stft_generic_kernel<0>, dim3((unsigned)grid), dim3(256), 4 * wave_bytes, st, q);
  else
    hipLaunchKernelGGL(stft_generic_kernel<1>, dim3((unsigned)grid), dim3(256), 4 * wave_bytes, st, q);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

static int check_audio_args(mm_plan* p, const float* d_audio, int64_t batch, int64_t n_samples,
                            int64_t stride) {
  if (!p || !d_audio || batch < 1 || n_samples < 1 || stride < n_samples) return MM_ERR_INVALID_ARG;
  // the radix-16 kernels index samples with 32-bit offsets (4 * index in the wave-per-frame kernel)
  if (n_samples > MM_MAX_SAMPLES) return MM_ERR_INVALID_ARG;
  return MM_OK;
}

int mm_stft_power_f32(mm_plan* p, const float* d_audio, int64_t batch, int64_t n_samples,
                      int64_t stride, float* d_power, void* stream) {
  int rc = check_audio_args(p, d_audio, batch, n_samples, stride);
  if (rc || !d_power) return rc ? rc : MM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  StageTimer tm(p, MM_STAGE_POWER, st);
  StftOut o;
  o.power = d_power;
  return launch_stft(p, 0, d_audio, batch, n_samples, stride, o, st);
}

int mm_logmel_f32(mm_plan* p, const float* d_audio, int64_t batch, int64_t n_samples, int64_t stride,
                  float* d_logmel, float* d_clipmax, void* stream) {
  int rc = check_audio_args(p, d_audio, batch, n_samples, stride);
  if (rc || !d_logmel || !d_clipmax) return rc ? rc : MM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(d_clipmax, 0x80, (size_t)batch * 4, st));
  {
    StageTimer tm(p, MM_STAGE_LOGMEL, st);
    StftOut o;
    o.logmel = d_logmel; o.key_max = (int*)d_clipmax;
    rc = launch_stft(p, 1, d_audio, batch, n_samples, stride, o, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(decode_keys_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st,
                     (int*)d_clipmax, batch);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

int mm_mfcc_f32(mm_plan* p, const float* d_audio, int64_t batch, int64_t n_samples, int64_t stride,
                float* d_mfcc, void* d_ws, size_t ws_bytes, void* stream) {
  int rc = check_audio_args(p, d_audio, batch, n_samples, stride);
  if (rc || !d_mfcc || !d_ws) return rc ? rc : MM_ERR_INVALID_ARG;
  if (ws_bytes < mm_workspace_bytes(p, batch, n_samples)) return MM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t T = mm_num_frames(&p->cfg, n_samples);
  float* logmel = (float*)d_ws;
  int* keys = (int*)((char*)d_ws + align_up((size_t)batch * p->cfg.n_mels * T * 4, 256));
  const bool clamp = p->cfg.top_db >= 0.0f;
  StftOut o;
  // A plan with empty mel filters (the reference's default maxFreq above Nyquist) sends EVERY clip through the fix-up
  // launch for the empty filters' share E[k] (thr - L0); with whole clips per workgroup that add (and the rare clamp
  // fix-up) runs at the end of the tile kernel's launch instead: no key arrays, no memset, no second launch.
  if (clamp && p->s16f_ok && !p->no_fuse && (p->s16f_flags & MM_S16F_SKIP) &&
      choose_kernel(p, 1, true, d_audio, n_samples, stride) == MM_K_W16S && s16_clip_mode_ok(p, batch, 0)) {
    StageTimer tm(p, MM_STAGE_LOGMEL, st);
    o.logmel = logmel; o.key_max = keys; o.key_nmin = keys + batch; o.mfcc = d_mfcc; o.frame_major = true; o.clip_only = true;
    rc = launch_stft(p, 1, d_audio, batch, n_samples, stride, o, st);
    if (rc) return rc;
    if (o.fused_tail) return MM_OK;
    return MM_ERR_UNSUPPORTED;       // not reached: the predicate above is the one launch_stft applies
  }
  {
    StageTimer tm(p, MM_STAGE_INIT, st);
    HIP_TRY(hipMemsetAsync(keys, 0x80, (size_t)batch * 8, st));   // max keys | keys of -min
  }
  {
    StageTimer tm(p, MM_STAGE_LOGMEL, st);
    o.logmel = logmel; o.key_max = keys; o.key_nmin = keys + batch; o.mfcc = d_mfcc; o.frame_major = true;
    rc = launch_stft(p, 1, d_audio, batch, n_samples, stride, o, st);
    if (rc) return rc;
  }
  if (o.fused_dct && !clamp) return MM_OK;
  {
    StageTimer tm(p, MM_STAGE_DCT, st);
    if (o.fused_dct) {
      // the kernel stored DCT(unclamped rows): only clips with min < max - top_db need the clamped DCT
      const int64_t bpc = (T + 255) / 256;
      if (batch * bpc > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
      // (o.skip_empty: the kernel that ran treated the filters without weights analytically -- the staged-sample kernel)
      hipLaunchKernelGGL(dct_fixup_kernel, dim3((unsigned)(batch * bpc)), dim3(256), 0, st, logmel, keys, keys + batch,
                         p->d_dct_t, d_mfcc, T, p->cfg.n_mels, p->cfg.n_mfcc, p->kp, p->cfg.top_db,
                         o.skip_empty ? p->d_s16f_dcta + (size_t)p->s16f_kb * p->s16f_nk * 64 : nullptr,
                         p->cfg.amin, p->db_offset);
    } else if (o.is_fm && p->d_dctw_a && !p->no_fuse) {
      // frame-major rows of the wave-per-frame kernel: clamp + DCT on the matrix pipe, a wave per 16-frame tile
      const int64_t n_items = batch * ((T + 15) / 16);
      const int pitch = (4 * p->dctw_nk) | 1;
      const size_t lds = ((size_t)p->dctfm_kb * p->dctw_nk * 64 + 4 * 16 * (size_t)pitch) * 4;
      const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(8, 163840 / (int64_t)lds));
      const int64_t grid = std::min<int64_t>((n_items + 3) / 4, per_cu * p->num_cus);      // persistent: A operands loaded once
#define MM_DCTW_GO(CC, UU, SS) hipLaunchKernelGGL((dct_clamp_fm_wave_kernel<CC, UU, SS>), dim3((unsigned)grid), dim3(256), lds, st, logmel, keys, \
                                                  p->d_dctw_a, d_mfcc, T, n_items, p->cfg.n_mels, p->cfg.n_mfcc, p->dctw_nk, p->dctfm_kb, p->cfg.top_db)
      if (p->dctw_ch == 10 && (p->cfg.n_mels & 3) == 0 && 16 * p->cfg.n_mels <= 5 * 256) MM_DCTW_GO(10, 5, false);   // 80 mel and below
      else if (p->dctw_ch == 10) MM_DCTW_GO(10, 8, true);
      else MM_DCTW_GO(8, 8, true);
#undef MM_DCTW_GO
    } else if (o.is_fm) {
      const int64_t bpc = (T + 63) / 64;
      if (batch * bpc > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
      launch_dct_fm(dim3((unsigned)(batch * bpc)), (size_t)64 * (p->cfg.n_mels + 1) * 4, st, logmel, keys, p->d_dct_t,
                    d_mfcc, T, p->cfg.n_mels, p->cfg.n_mfcc, p->kp, p->cfg.top_db);
    } else {
      const int64_t bpc = (T + 255) / 256;
      if (batch * bpc > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
      hipLaunchKernelGGL(dct_clamp_kernel, dim3((unsigned)(batch * bpc)), dim3(256), 0, st, logmel, keys,
                         p->d_dct_t, d_mfcc, T, p->cfg.n_mels, p->cfg.n_mfcc, p->kp, p->cfg.top_db);
    }
    HIP_TRY(hipGetLastError());
  }
  return MM_OK;
}

static int launch_rfft(mm_plan* p, const float* d_in, int64_t rows, int64_t in_len, int64_t in_stride,
                       int n, float* d_out, hipStream_t st) {
  if (n > 8192) return MM_ERR_UNSUPPORTED;      // longer trajectories: mm_hilbert_rfft_f32 (a transform in global memory)
  RfftParams q;
  q.in = d_in; q.rows = rows; q.in_len = in_len; q.in_stride = in_stride; q.n = n;
  q.log2nc = ilog2(n) - 1; q.rows_per_wave = 4; q.tw = p->d_tw; q.out = d_out;
  if (!p->force_generic && (n == 512 || n == 1024)) {
    const int rows_per_wave = (n == 512) ? 4 : 2;
    const int64_t groups = (rows + rows_per_wave - 1) / rows_per_wave;
    int64_t grid = (groups + 3) / 4;
    if (grid > 2048) grid = 2048;
    const bool fast = (in_len == n) && (in_stride % 2 == 0) && (((uintptr_t)d_in & 7) == 0);
    if (n == 512) {
      if (fast) hipLaunchKernelGGL((rfft16_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, st, q);
      else hipLaunchKernelGGL((rfft16_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, st, q);
    } else {
      if (fast) hipLaunchKernelGGL((rfft16_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, st, q);
      else hipLaunchKernelGGL((rfft16_kernel<2, false>), dim3((unsigned)grid), dim3(256), 0, st, q);
    }
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (!p->force_generic && n == 2048 && p->rf2k_ok) {
    // one row per wave, 8 waves per workgroup, two workgroups per CU
    int64_t grid = (rows + 7) / 8;
    if (grid > 512) grid = 512;
    const size_t lds = (size_t)(64 * MM_WPF_LT_PITCH + 8 * WpfGeo<4>::XBUF) * 4;
    const bool fast = (in_len == n) && (in_stride % 2 == 0) && (((uintptr_t)d_in & 7) == 0);
    if (fast) hipLaunchKernelGGL((rfft_wpf_kernel<4, true>), dim3((unsigned)grid), dim3(512), lds, st, q, p->d_rf2k_lane_tab);
    else hipLaunchKernelGGL((rfft_wpf_kernel<4, false>), dim3((unsigned)grid), dim3(512), lds, st, q, p->d_rf2k_lane_tab);
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  const int nc = n / 2;
  const int64_t grid = (rows + 4 * q.rows_per_wave - 1) / (4 * q.rows_per_wave);
  if (grid > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  // (n = 8192 needs 128 KB of dynamic LDS: the function attribute is raised once, in mm_plan_create)
  hipLaunchKernelGGL(rfft_generic_kernel, dim3((unsigned)grid), dim3(256), (size_t)4 * nc * 8, st, q);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

int mm_rfft_f32(mm_plan* p, const float* d_in, int64_t rows, int64_t in_len, int64_t in_stride,
                int32_t n, float* d_out, void* stream) {
  if (!p || !d_in || !d_out || rows < 1 || in_len < 1 || in_stride < in_len) return MM_ERR_INVALID_ARG;
  if (n < 32 || n > 8192 || (n & (n - 1))) return MM_ERR_UNSUPPORTED;
  if (in_len > n) return MM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  StageTimer tm(p, MM_STAGE_RFFT, st);
  return launch_rfft(p, d_in, rows, in_len, in_stride, n, d_out, st);
}

int mm_modspec_f32(mm_plan* p, const float* d_mfcc, int64_t batch, int64_t n_frames, float* d_out,
                   void* stream) {
  if (!p || !d_mfcc || !d_out || batch < 1 || n_frames < 1) return MM_ERR_INVALID_ARG;
  const int n = mm_mod_fft_len(&p->cfg, n_frames);
  if (n < 0) return n;
  hipStream_t st = (hipStream_t)stream;
  StageTimer tm(p, MM_STAGE_MODSPEC, st);
  return launch_rfft(p, d_mfcc, batch * p->cfg.n_mfcc, n_frames, n_frames, n, d_out, st);
}

int mm_mfcc_modspec_f32(mm_plan* p, const float* d_audio, int64_t batch, int64_t n_samples, int64_t stride,
                        float* d_mfcc, float* d_modspec, void* d_ws, size_t ws_bytes, void* stream) {
  int rc = check_audio_args(p, d_audio, batch, n_samples, stride);
  if (rc || !d_mfcc || !d_modspec || !d_ws) return rc ? rc : MM_ERR_INVALID_ARG;
  if (ws_bytes < mm_workspace_bytes(p, batch, n_samples)) return MM_ERR_WORKSPACE;
  const int64_t T = mm_num_frames(&p->cfg, n_samples);
  const int n_mod = mm_mod_fft_len(&p->cfg, T);
  if (n_mod < 0) return n_mod;
  hipStream_t st = (hipStream_t)stream;
  if (choose_kernel(p, 1, true, d_audio, n_samples, stride) == MM_K_W16S && p->s16f_ok && !p->no_fuse &&
      s16_clip_mode_ok(p, batch, n_mod)) {
    // ONE launch: a workgroup owns whole clips, so the clip extremes never leave it (no key arrays, no memset),
    // the clamped DCT of a clip that needs it and the trajectory rFFT of every finished clip run in the kernel
    StageTimer tm(p, MM_STAGE_LOGMEL, st);
    StftOut o;
    int* keys = (int*)((char*)d_ws + align_up((size_t)batch * p->cfg.n_mels * T * 4, 256));
    o.logmel = (float*)d_ws; o.key_max = keys; o.key_nmin = keys + batch; o.mfcc = d_mfcc; o.frame_major = true;
    o.mod = d_modspec; o.n_mod = n_mod;
    rc = launch_stft(p, 1, d_audio, batch, n_samples, stride, o, st);
    if (rc) return rc;
    if (o.fused_tail) return MM_OK;
    return MM_ERR_UNSUPPORTED;       // not reached: the predicate above is the one launch_stft applies
  }
  rc = mm_mfcc_f32(p, d_audio, batch, n_samples, stride, d_mfcc, d_ws, ws_bytes, stream);
  if (rc) return rc;
  return mm_modspec_f32(p, d_mfcc, batch, T, d_modspec, stream);
}

int mm_plan_fused_tail(const mm_plan* p, int64_t batch, int64_t n_samples) {
  if (!p || batch < 1 || n_samples < 1) return MM_ERR_INVALID_ARG;
  const int64_t T = mm_num_frames(&p->cfg, n_samples);
  const int n_mod = mm_mod_fft_len(&p->cfg, T);
  if (n_mod < 0) return 0;
  return (choose_kernel(p, 1, false, nullptr, 0, 0) == MM_K_W16S && n_samples >= 4 && p->s16f_ok && !p->no_fuse &&
          s16_clip_mode_ok(p, batch, n_mod)) ? 1 : 0;
}

int mm_timing_enable(mm_plan* p, int on) {
  if (!p) return MM_ERR_INVALID_ARG;
  p->timing_on = on;
  return MM_OK;
}

int mm_timing_read(mm_plan* p, double* ms_sum, int64_t* count) {
  if (!p || !ms_sum || !count) return MM_ERR_INVALID_ARG;
  for (int i = 0; i < p->ev_used; ++i) {
    HIP_TRY(hipEventSynchronize(p->ev_pool[2 * i + 1]));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[2 * i], p->ev_pool[2 * i + 1]));
    p->t_sum[p->ev_stage[i]] += ms;
    p->t_cnt[p->ev_stage[i]] += 1;
  }
  p->ev_used = 0;
  for (int s = 0; s < MM_NUM_STAGES; ++s) {
    ms_sum[s] = p->t_sum[s];
    count[s] = p->t_cnt[s];
    p->t_sum[s] = 0.0;
    p->t_cnt[s] = 0;
  }
  return MM_OK;
}

}  // extern "C"
