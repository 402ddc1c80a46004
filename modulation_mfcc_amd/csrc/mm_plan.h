// struct mm_plan (the MFCC plan behind include/modmfcc.h) and the per-stage event timer: shared by mm_api.hip (which
// creates and runs plans) and mm_tail.hip (whose change tail reads the plan's configuration and timing state).
#pragma once
#include "mm_common.h"

// any-length STFT (mm_anyfft.hip.inc): host-side description of the transform of an n_fft that is not a power of two
struct AnyPlan {
  int nn = 0, packed = 0, n_pass = 0, radix[16] = {0}, M = 0, log2M = 0, tpf = 64;
  unsigned grp_bytes = 0, b_off = 0, p_off = 0;
  float2 *d_tw = nullptr, *d_split = nullptr, *d_chirp = nullptr, *d_bhat = nullptr;
  // LDSTAB: packed constant tables (window | tw | split | chirp | mel_w | mel_start | mel_len | mel_off)
  float* d_tabpack = nullptr;
  int tab_floats = 0, o_tw = 0, o_split = 0, o_chirp = 0, o_melw = 0, o_mstart = 0, o_mlen = 0, o_moff = 0;
  bool lds_tab = false;
  bool ok = false;
  int fb = 0;                    // > 0: the batched kernel (stft_anyb_kernel) with this many frames per wave at once
  int reg2 = 0, reg2_r2 = 0;     // > 0: the two-stage register kernel (mm_reg2.hip), nn = reg2 x reg2_r2
  unsigned fb_grp_bytes = 0;     // its LDS bytes per wave: two buffers of fb x nn complex points
};

// 12-wave matrix-pipe variant (mm_logmel12m.hip.inc): unit lists kept in the plan
#define MM_M12_MW 4             // mel waves
#define MM_M12_UMAX 12          // units per mel wave

#define MM_MAX_TIMED 16384
#define MM_MAX_SAMPLES (((int64_t)1 << 29) - 8192)

struct mm_plan {
  mm_config cfg;
  int device;
  int n_bins, log2nc, kp;
  float db_offset;
  int path;           // 0 generic, 1 radix-16 register kernels
  int force_generic;
  float* d_window;
  float2* d_tw;
  int *d_mel_start, *d_mel_len, *d_mel_off;
  float* d_mel_w;
  float* d_dct_t;
  float* d_sw_tab;             // mel run table of the fused kernel (headers + groups)
  int* d_sw_part;
  int sw_n_runs, sw_n_tab16;
  size_t lm_lds_bytes;
  float *d_w16_tab, *d_lane_tab;   // 16-wave variant: its own run table + per-lane records
  int* d_w16_part;
  int w16_n_runs, w16_n_tab16, w16_ok;
  size_t w16_lds_bytes;
  int s16_nr;                      // staged-sample variant: 16-byte groups per thread and tile (0: not usable)
  size_t s16_lds_bytes;
  // staged-sample variant with the (unclamped) DCT fused in: its own run table (four half-size parts for the
  // DCT waves), DCT A operands, LDS layout
  float *d_s16f_tab, *d_s16f_dcta;
  int* d_s16f_part;
  int s16f_ok, s16f_n_runs, s16f_n_tab16, s16f_lt_rows, s16f_nk, s16f_kb;
  unsigned s16f_lt_off, s16f_dcta_off;
  unsigned long long s16f_roles;
  size_t s16f_lds_bytes;
  int wpf_waves_half; size_t wpf_lds_half;   // launch geometry of the half-band instantiations (up to sixteen waves)
  int wpf_half, wpf_pairs;         // the mel bank reads no bin >= NC / 2: split pairs the kernel forms (WpfParams::n_pairs / half_band)
  int wpf_z;                       // 3: the window is zero for a lane's first and last three pairs (logmel_wpf_kernel<.., Z = 3>)
  int s16_halfwin;                 // the 512-point window is zero outside [128, 384): the staged kernel prunes its first stage
  int s16f_flags;                  // Logmel512Params::dct_flags (MM_S16F_SINGLE | MM_S16F_SKIP)
  // 12-wave MFMA-mel variant (mm_logmel12m.hip.inc)
  float *d_m12_a, *d_m12_dct, *d_zeros;
  int m12_units[MM_M12_MW * MM_M12_UMAX * 4], m12_nunits[8];
  int m12_ok, m12_nb, m12_nstep8, m12_nr, m12_s_floats, m12_fused_dct;
  unsigned m12_win_off, m12_tw_off, m12_a_off, m12_dct_off, m12_part_off, m12_cnt_off;
  int m12_n_a2;
  size_t m12_lds_bytes;
  float* d_dctfm_a;                // dct_clamp_fm_mfma_kernel: A operands [kb][nk][64] (nullptr: VALU kernel)
  int dctfm_nk, dctfm_kb;
  size_t dctfm_lds;
  float* d_dctw_a;                 // dct_clamp_fm_wave_kernel<CH>: the same, steps padded to a multiple of CH with zeros
  int dctw_nk, dctw_ch;
  int variant;                     // mm_plan_set_variant: 0 = automatic
  int no_fuse;                     // mm_plan_set_fuse_dct(0): always run the separate clamp + DCT kernel
  int no_fuse_tail;                // mm_plan_set_fuse_tail(0): mm_mfcc_modspec_f32 always runs its separate launches
  int fuse_tail_wide;              // mm_plan_set_fuse_tail(2): clip mode also for 2048-point trajectories and for mm_mfcc_f32 (empty filters)
  unsigned s16f_red_off;           // clip mode: LDS offset of the per-wave clip max / min slots

  float* d_window_e;                       // n_fft < 512 embedded in the 512-point kernels: window centred in 512
  int embed;                               // 512 / n_fft for such plans, else 1
  float *d_k2_lane_tab, *d_k2_mel_lane;   // wave-per-frame-group kernel (n_fft 512 / 1024 / 2048)
  float* d_rf2k_lane_tab;                 // rfft_wpf_kernel<4> (stage-isolated rFFT, n = 2048)
  int rf2k_ok;
  int k2_ok, wpf_r, wpf_waves, wpf_group_max;
  size_t wpf_lds_bytes;
  size_t wpf_lds16;                        // sixteen-wave form of the same kernel (W16): LDS bytes, and whether it applies
  int wpf_w16;
  int num_cus;
  float* d_h16_tab; int* d_h16_part;       // 32-frame-tile / two-workgroup experiment (mm_logmel16h.hip.inc)
  int h16_ok, h16_n_pairs, h16_n_tab16;
  size_t h16_lds_bytes;
  AnyPlan any;                             // any-length STFT (mm_anyfft.hip.inc): n_fft that is not a power of two in [32, 4096]
  // timing
  int timing_on;
  std::vector<hipEvent_t> ev_pool;  // pairs
  std::vector<int> ev_stage;
  int ev_used;
  double t_sum[MM_NUM_STAGES];
  int64_t t_cnt[MM_NUM_STAGES];
};

namespace {

struct StageTimer {
  mm_plan* p;
  hipStream_t s;
  int idx;
  StageTimer(mm_plan* plan, int stage, hipStream_t stream) : p(plan), s(stream), idx(-1) {
    if (!p->timing_on || p->ev_used >= MM_MAX_TIMED) return;
    if (p->timing_on != 1 && !((p->timing_on >> (stage + 1)) & 1)) return;   // stage mask
    if ((size_t)(2 * p->ev_used + 1) >= p->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      p->ev_pool.push_back(a);
      p->ev_pool.push_back(b);
    }
    idx = p->ev_used++;
    p->ev_stage.resize(p->ev_used);
    p->ev_stage[idx] = stage;
    (void)hipEventRecord(p->ev_pool[2 * idx], s);
  }
  ~StageTimer() {
    if (idx >= 0) (void)hipEventRecord(p->ev_pool[2 * idx + 1], s);
  }
};

}  // namespace
