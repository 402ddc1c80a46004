/*
 * modmfcc.h -- C ABI of libmodmfcc.so: MI355X (gfx950) MFCC + modulation-spectrum extractor.
 *
 * The reference (aaron-randreth/modulation-mfcc) is pure Python and has no FFI; the
 * interface each entry point replaces is therefore a Python call site:
 *
 *   mm_mfcc_f32            <- librosa.feature.mfcc(y, sr, n_mfcc, win_length, hop_length,
 *                             n_fft, fmin, fmax) as called at script/mfcc.py:387
 *                             (rows A1-A6 of SURVEY.md section 8(a)), batched over clips
 *   mm_num_frames          <- frame count implied by that call, used at script/mfcc.py:390
 *   mm_logmel_f32          <- the melspectrogram + 10*log10 half of the same call (stage test)
 *   mm_stft_power_f32      <- the stft + |.|^2 half of the same call (stage test / roofline)
 *   mm_rfft_f32            <- the batched rFFT stage in isolation (np.fft.rfft rows; the
 *                             "% HBM roofline (rFFT)" metric of BASELINE.json)
 *   mm_modspec_f32         <- row A8: rFFT over every coefficient's time trajectory
 *                             (build-defined; no reference site)
 *   mm_mfcc_change_f64     <- script/mfcc.py:392-427 (drop c0, Butterworth sosfiltfilt,
 *                             gradient or Savitzky-Golay derivative, norm, output filter) -- row N1
 *   mm_sosfiltfilt_f64     <- applyFilter(filt='iir') (script/mfcc.py:29-135): scipy sosfiltfilt on a batch
 *   mm_stencil_f64         <- get_velocity (script/calc.py:593-650): np.gradient / savgol_filter /
 *                             findiff derivative of a curve -- row N2
 *   mm_rms_f32,
 *   mm_hilbert_envelope    <- calculate_amplitude_envelope (script/calc.py:284-343): librosa.feature.rms /
 *                             |scipy.signal.hilbert| -- row N3
 *   mm_pcm_decode_f32,
 *   mm_resample_f32        <- librosa.load(path, sr=sigSr, mono=False) (script/mfcc.py:284,373) -- row N4
 *   mm_build_window/mel/dct<- scipy.signal.get_window('hann'), librosa.filters.mel,
 *                             scipy.fftpack.dct(type=2, norm='ortho') constant tables
 *
 * Contract:
 *   - plain pointers and sizes only; every data/workspace pointer is DEVICE memory owned by
 *     the caller (e.g. torch tensor.data_ptr()); the plan owns only constant device tables;
 *   - every compute call is asynchronous on the caller's hipStream_t (passed as void*), does
 *     no allocation and no synchronisation (graph-capture safe);
 *   - functions return MM_OK (0) or a negative mm_status; nothing throws, nothing exits;
 *   - a plan is bound to the device current at creation.  Its constant tables never change after mm_plan_create;
 *     the few SET-UP calls that do change plan state -- mm_plan_set_variant, mm_plan_set_fuse_dct,
 *     mm_plan_set_fuse_tail, mm_plan_force_generic, mm_timing_enable / mm_timing_read -- are unsynchronised
 *     host-side switches: call them from the thread that owns the plan, between compute calls, never while another
 *     thread is inside a compute call on the same plan.  A plan that is only computed with (no set-up calls, timing
 *     off) may be shared by threads that each pass their own buffers, workspace and stream.
 */
#ifndef MODMFCC_H
#define MODMFCC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_VERSION 123 /* 0.3.3 */

typedef enum mm_status {
  MM_OK = 0,
  MM_ERR_INVALID_ARG = -1,  /* NULL pointer, non-positive size, fmax <= fmin ...              */
  MM_ERR_UNSUPPORTED = -2,  /* n_fft > 8192, center == 0, n_mod_fft not a power of two ...    */
  MM_ERR_HIP = -3,          /* a HIP runtime call failed (see mm_last_hip_error)              */
  MM_ERR_WORKSPACE = -4,    /* workspace smaller than mm_workspace_bytes()                    */
  MM_ERR_ALLOC = -5
} mm_status;

/* POD configuration; mirrors the keyword arguments of the librosa call at script/mfcc.py:387
 * plus the build's extensions (n_mels, preemph, top_db, amin, n_mod_fft). */
typedef struct mm_config {
  double sr;          /* sample rate in Hz (sigSr)                                            */
  int32_t n_fft;      /* FFT length: ANY integer 2..8192 (librosa: any n_fft >= win_length)   */
  int32_t win_length; /* int(winLen*sigSr), 1..n_fft                                          */
  int32_t hop_length; /* int(tStep*sigSr), >= 1                                               */
  int32_t n_mels;     /* librosa default 128; 1..256                                          */
  int32_t n_mfcc;     /* 1..n_mels                                                            */
  double fmin;        /* minFreq                                                              */
  double fmax;        /* maxFreq (may exceed Nyquist: empty filters, as the reference does)   */
  float preemph;      /* 0 = off (reference behaviour); y[n] - a*y[n-1]                       */
  float top_db;       /* 80; < 0 disables the per-clip clamp                                  */
  float amin;         /* 1e-10                                                                */
  int32_t center;     /* must be 1 (librosa center=True, pad_mode='constant')                 */
  int32_t n_mod_fft;  /* trajectory rFFT length; 0 = next power of two >= n_frames (> 8192: mm_hilbert_rfft_f32) */
} mm_config;

typedef struct mm_plan mm_plan;

/* kernels whose device time mm_timing_read() reports */
enum {
  MM_STAGE_LOGMEL = 0,   /* fused frame+window+rFFT+power+mel+log (+per-clip max)             */
  MM_STAGE_DCT = 1,      /* top_db clamp + DCT-II                                             */
  MM_STAGE_MODSPEC = 2,  /* trajectory rFFT                                                   */
  MM_STAGE_RFFT = 3,     /* stage-isolated batched rFFT                                       */
  MM_STAGE_POWER = 4,    /* fused frame+window+rFFT+power                                     */
  MM_STAGE_CHANGE = 5,   /* MFCC-change tail                                                  */
  MM_STAGE_INIT = 6,     /* clip-max reset                                                    */
  MM_NUM_STAGES = 8
};

int mm_version(void);
const char* mm_strerror(int status);
const char* mm_last_hip_error(void);

/* ---- host-only helpers (no GPU needed) ------------------------------------------------ */
int mm_config_default(mm_config* cfg);                           /* reference defaults      */
int mm_config_validate(const mm_config* cfg);
int64_t mm_num_frames(const mm_config* cfg, int64_t n_samples);  /* 1 + (n_samples + 2 (n_fft / 2) - n_fft) / hop: 1 + n_samples / hop for even n_fft */
int32_t mm_num_bins(const mm_config* cfg);                       /* n_fft/2 + 1             */
int32_t mm_mod_fft_len(const mm_config* cfg, int64_t n_frames);  /* resolved n_mod_fft      */
int mm_build_window(const mm_config* cfg, float* out /*[n_fft]*/);
int mm_build_mel(const mm_config* cfg, float* out /*[n_mels][n_fft/2+1]*/);
int mm_build_dct(const mm_config* cfg, float* out /*[n_mfcc][n_mels]*/);
/* Sweep form of the mel matrix used by the lane<->frame kernels (host-only, for tests): bin k
 * feeds filter d[k] with weight wlo[k] and filter d[k]+1 with weight whi[k]; part = per-wave
 * {k_begin, k_end, m_begin, m_end}.  Returns MM_ERR_UNSUPPORTED if the matrix has another shape. */
int mm_build_mel_sweep(const mm_config* cfg, int n_waves, float* wlo /*[n_bins]*/,
                       float* whi /*[n_bins]*/, int32_t* d /*[n_bins]*/, int32_t* part /*[n_waves][4]*/);
/* Run form of the same sweep, exactly as the fused kernel reads it from LDS (host-only, for tests):
 * hdr [n_runs][4] = {first bin (multiple of 4), n_groups, first group, filter d}; grp [n_groups][8] =
 * {weight in filter d x4, weight in filter d+1 x4}; part [n_waves][4] = {run_begin, run_end, m_begin,
 * m_end}.  counts[0] = n_runs, counts[1] = n_groups.  Capacities are in runs / groups. */
int mm_build_mel_runs(const mm_config* cfg, int n_waves, int32_t* hdr, int32_t hdr_cap, float* grp,
                      int32_t grp_cap, int32_t* part, int32_t* counts);
/* Butterworth low-pass as second-order sections, scipy.signal.butter(order, wn, 'low',
 * output='sos') layout [n_sections][6]; returns the number of sections or a negative status. */
int mm_build_butter_sos(int order, double wn, double* sos /*[(order+1)/2][6]*/);

/* ---- plan ----------------------------------------------------------------------------- */
int mm_plan_create(const mm_config* cfg, mm_plan** out);
int mm_plan_destroy(mm_plan* plan);
int mm_plan_config(const mm_plan* plan, mm_config* out);
/* Which fused kernel a log-mel / MFCC call of this plan runs on, for a regular call (aligned rows,
 * n_samples >= 4): 0 = generic LDS radix-2; 1 = register radix-16, 8 waves per workgroup; 2 = register
 * radix-16, 16 waves, direct loads (n_fft 512, even hop, no pre-emphasis); 3 = register radix-16
 * wave-per-frame-group kernel (n_fft 1024 / 2048, and what is left of n_fft 512); 4 = variant 2 with the
 * tile's samples staged through LDS (hop <= 252; any hop parity, row alignment and length, optional
 * pre-emphasis); 5 = 12-wave kernel with the mel contraction (and for n_mfcc <= 16 the DCT-II) on the
 * matrix pipe, 48-frame double-buffered power tiles (n_fft 512, hop <= 170 at 40 mel: the tables must fit
 * the 160 KB of LDS) -- opt-in (mm_plan_set_variant): measured slower than variant 4, see DESIGN.md.  n_fft 64 / 128 / 256 plans use the n_fft 512
 * variants too (frames zero-padded to 512 points: same power at every (512/n_fft)-th bin).  6 = the any-length
 * kernel: every n_fft that is not a power of two in [32, 4096] -- 400, 600, 1000, 1536, a prime, 8192, 16 -- as a
 * mixed-radix (2 / 3 / 5 / 7) Stockham FFT in LDS, Bluestein's chirp-z transform for lengths with other prime factors
 * (such a plan has this one kernel: variants and force_generic do not apply). */
int mm_plan_kernel_path(const mm_plan* plan);
/* 1 if mm_mfcc_f32 of this plan applies the DCT-II inside the log-mel kernel (variants 4 and 5: the DCT
 * of the unclamped rows, followed by a fix-up launch that redoes only the clips whose minimum lies more
 * than top_db under their maximum), 0 if it runs the separate clamp + DCT kernel. */
int mm_plan_fused_dct(const mm_plan* plan);
/* on = 0: always run the separate clamp + DCT kernel (A/B measurements, cross-checks); returns the
 * previous setting.  Default on. */
int mm_plan_set_fuse_dct(mm_plan* plan, int on);
/* mm_mfcc_modspec_f32 in ONE launch (whole clips per workgroup: the clip maximum / minimum never leave it, the
 * clamp fix-up and the trajectory rFFT of the workgroup's clips run at the end of the tile kernel)?  1 when the
 * n_fft 512 staged-sample kernel with its fused DCT takes the call, the trajectory length is 512 or 1024 and
 * `batch` clips spread over the compute units within 4 % (at most 32 per workgroup); 0 = the separate launches.
 * mm_plan_set_fuse_tail(plan, 0) pins the separate launches (A/B measurements, cross-checks; returns the previous
 * setting; default on = 1).  2 widens clip mode to 2048-point trajectories (1025 .. 2048 frames per clip: the 2048-point
 * transform at the end of the launch) and to mm_mfcc_f32 on plans with empty mel filters (their add inside the launch):
 * faster when no clip clamps, slower when a few do (one workgroup pays a clip's whole fix-up) -- see DESIGN.md 4.3.  The same switch selects the device form of mm_mfcc_change_f64 (on: the clip-resident
 * single launch; off: the time-major launches). */
int mm_plan_fused_tail(const mm_plan* plan, int64_t batch, int64_t n_samples);
int mm_plan_set_fuse_tail(mm_plan* plan, int on);
/* force the generic kernels (debug / cross-check); returns previous value */
int mm_plan_force_generic(mm_plan* plan, int on);
/* pin one of the variants above (1..5) for the calls it can take, 0 = automatic choice; returns the
 * previous setting.  Results of all variants agree within float32 round-off (tests/test_gpu_parity.py);
 * this is how the tests and tools/ab.sh select a kernel -- the library reads no environment variable. */
int mm_plan_set_variant(mm_plan* plan, int variant);
size_t mm_workspace_bytes(const mm_plan* plan, int64_t batch, int64_t n_samples);

/* ---- compute (device pointers, async on `stream`) ---------------------------------------
 * d_audio : float32 [batch][audio_stride], n_samples valid per clip
 * d_mfcc  : float32 [batch][n_mfcc][n_frames]   (coefficient-major, librosa layout)          */
int mm_mfcc_f32(mm_plan* plan, const float* d_audio, int64_t batch, int64_t n_samples,
                int64_t audio_stride, float* d_mfcc, void* d_workspace, size_t ws_bytes,
                void* stream);

/* d_logmel : float32 [batch][n_mels][n_frames], 10*log10(max(amin, mel)) BEFORE the clamp;
 * d_clipmax: float32 [batch], per-clip maximum of d_logmel                                   */
int mm_logmel_f32(mm_plan* plan, const float* d_audio, int64_t batch, int64_t n_samples,
                  int64_t audio_stride, float* d_logmel, float* d_clipmax, void* stream);

/* d_power : float32 [batch][n_frames][n_fft/2+1] (frame-major)                               */
int mm_stft_power_f32(mm_plan* plan, const float* d_audio, int64_t batch, int64_t n_samples,
                      int64_t audio_stride, float* d_power, void* stream);

/* rows of `in_len` (<= n, zero padded) real samples -> complex64 [rows][n/2+1] interleaved;
 * n a power of two in [32, 8192].                                                             */
int mm_rfft_f32(mm_plan* plan, const float* d_in, int64_t rows, int64_t in_len,
                int64_t in_stride, int32_t n, float* d_out, void* stream);

/* d_mfcc [batch][n_mfcc][n_frames] -> complex64 [batch][n_mfcc][n_mod/2+1]                   */
int mm_modspec_f32(mm_plan* plan, const float* d_mfcc, int64_t batch, int64_t n_frames,
                   float* d_modspec, void* stream);

/* Rows A1-A6 + A8 in one call: d_mfcc as mm_mfcc_f32 (bit for bit), d_modspec as mm_modspec_f32 of it (to
 * float32 round-off: the in-kernel transform is a second instantiation of the same code); one kernel launch
 * where mm_plan_fused_tail() says so, the two calls' launches otherwise.  Workspace: mm_workspace_bytes(). */
int mm_mfcc_modspec_f32(mm_plan* plan, const float* d_audio, int64_t batch, int64_t n_samples,
                        int64_t audio_stride, float* d_mfcc, float* d_modspec, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* MFCC-change tail (script/mfcc.py:392-427, outFilter 'iir' low-pass or None): d_mfcc [batch][n_mfcc]
 * [n_frames] f32 -> d_change [batch][n_frames] f64.  diff_method 0 = np.gradient (diffMethod='grad',
 * script/mfcc.py:405-407), 1 = savgol_filter(x, 3, 2, deriv=1, mode='interp') (any other diffMethod,
 * script/mfcc.py:409-412; needs n_frames >= 3).
 * sos1/sos2: HOST pointers to [n_sec][6] Butterworth sections (first / output filter).
 * Float64 recursion with fused multiply-adds (5 operations per section and sample), the odd extension of the float32
 * MFCC rows formed in float32 as scipy does on a float32 array: within ~1e-11 of scipy (tests: 1e-9 of the curve's maximum).
 * Three device forms, the same arithmetic per sample (they differ by rounding, ~1e-13 relative at most):
 *   - clip-resident (default for both filters <= 4 sections and clips up to ~5000 frames): one launch, a workgroup per clip, the clip's rows and the curve
 *     in LDS (rows in groups when they do not fit at once), the filters time-parallel over 64 chunks per row;
 *     of the workspace only ~10 KB of filter tables are used;
 *   - segmented rows (clips so long that LDS holds less than a quarter of their rows at once, about 5000 frames at
 *     12 rows -- one recording at the reference's default 1 ms step is 10 001 frames per ten seconds): both filters
 *     through the kernels of
 *     mm_sosfiltfilt_f64 (a wave per 1088 samples of a row), the derivative + norm between them;
 *   - time-major (more than 4 sections, and after mm_plan_set_fuse_tail(plan, 0)): eight launches over a float64
 *     workspace of [frames][rows of all clips].                                                */
int mm_mfcc_change_f64(mm_plan* plan, const float* d_mfcc, int64_t batch, int64_t n_frames,
                       int32_t remove_first, int32_t diff_method, const double* sos1, int32_t n_sec1,
                       const double* sos2, int32_t n_sec2, double* d_change,
                       void* d_workspace, size_t ws_bytes, void* stream);
/* Workspace: mm_change_workspace_bytes_for() = what the form THIS call takes needs (same arguments as the call: a few KB
 * for the clip-resident form, ~2 x rows x frames doubles for the other two); mm_change_workspace_bytes() = the upper
 * bound over all forms and filters, for callers that size their buffer before they know the filters.  The call checks
 * ws_bytes against the former. */
size_t mm_change_workspace_bytes(const mm_plan* plan, int64_t batch, int64_t n_frames);
size_t mm_change_workspace_bytes_for(const mm_plan* plan, int64_t batch, int64_t n_frames, int32_t remove_first,
                                     const double* sos1, int32_t n_sec1, const double* sos2, int32_t n_sec2);

/* Zero-phase IIR filter of float64 curves: scipy.signal.sosfiltfilt(sos, x) with its defaults (odd
 * extension by 3 * ntaps samples, sosfilt_zi initial state), i.e. the 'iir' branch of applyFilter
 * (script/mfcc.py:29-135, script/calc.py:23-129) on a batch.  d_x [rows][x_stride] -> d_y [rows][n];
 * sos: HOST pointer to [n_sec][6] sections; n must exceed the padding length.  Needs no plan.
 * Up to 4 sections (Butterworth order <= 8): rows of any length in segments of 64 x 17 samples, a wave per segment,
 * the recursion closed over chunk / segment / row levels (three launches per direction, 24 bytes of traffic per
 * sample and direction; from 128 rows on a workgroup walks a row, sixteen segments per round: one launch per direction,
 * 16 bytes per sample: 256 rows x 160 000 samples in 0.39 ms); more sections: a lane per row, eight chunks.
 * Differs from scipy's sequential recursion by rounding only. */
int mm_sosfiltfilt_f64(const double* d_x, int64_t rows, int64_t n, int64_t x_stride, const double* sos,
                       int32_t n_sec, double* d_y, void* d_workspace, size_t ws_bytes, void* stream);
size_t mm_sosfiltfilt_workspace_bytes(int64_t rows, int64_t n);
/* The same for FLOAT32 rows (librosa's RMS envelope, MFCC rows): scipy.signal.sosfiltfilt forms the odd extension
 * `2 x[0] - x[k]` in the array's own type before its recursion upcasts, so the padded samples of a float32 curve are
 * rounded to float32 -- this entry point does exactly that (the float64 entry point on an upcast copy differs from
 * scipy's result on the float32 array by ~3e-9).  Same workspace. */
int mm_sosfiltfilt_f32_f64(const float* d_x, int64_t rows, int64_t n, int64_t x_stride, const double* sos,
                           int32_t n_sec, double* d_y, void* d_workspace, size_t ws_bytes, void* stream);

/* Derivative transforms of get_velocity (script/calc.py:593-650; row N2) as ONE banded linear operator
 * along time on float64 rows: interior output i = (sum_k c[k] x[i + off[k]]) / den_c; the n_edge first
 * outputs = (el[i] . x[0 .. edge_w)) / den_e, the n_edge last ones = (er[i] . x[n - edge_w .. n)) / den_e.
 * np.gradient (n_c 2, off {-1, +1}, c {-1, 1}, den_c 2h; one edge row {-1, 1}, den_e h) comes out bit for
 * bit; Savitzky-Golay (mode='interp') and the findiff stencils to float64 round-off.  The host builds the
 * taps (modulation_mfcc_amd/calc.py: velocity_stencil).  d_x [rows][x_stride] -> d_y [rows][n], n >=
 * max(2 n_edge, edge_w, spread of off).  Needs no plan. */
#define MM_ST_MAXW 16
#define MM_ST_MAXE 8
typedef struct mm_stencil {
  int32_t n_c, n_edge, edge_w, reserved;
  int32_t off[MM_ST_MAXW];
  double c[MM_ST_MAXW];
  double el[MM_ST_MAXE][MM_ST_MAXW];
  double er[MM_ST_MAXE][MM_ST_MAXW];
  double den_c, den_e;
} mm_stencil;
int mm_stencil_f64(const mm_stencil* st, const double* d_x, int64_t rows, int64_t n, int64_t x_stride,
                   double* d_y, void* stream);

/* Framewise RMS (row N3): librosa.feature.rms(y, frame_length, hop_length, center, pad_mode=
 * 'constant') as called at script/calc.py:331 / script/mfcc.py:247.  d_audio [batch][audio_stride]
 * -> d_rms [batch][n_out], n_out = 1 + (n_samples + 2*(center ? frame_length/2 : 0) - frame_length)
 * / hop_length.  Needs no plan. */
int64_t mm_rms_num_frames(int64_t n_samples, int32_t frame_length, int32_t hop_length, int32_t center);
int mm_rms_f32(const float* d_audio, int64_t batch, int64_t n_samples, int64_t audio_stride,
               int32_t frame_length, int32_t hop_length, int32_t center, float* d_rms, void* stream);

/* Hilbert envelope (row N3): np.abs(scipy.signal.hilbert(x)) as called at script/calc.py:286 -- the DFT of the
 * clip at its own length n (any integer 1 .. 2^24), negative frequencies zeroed, positive ones doubled, the
 * inverse DFT, the magnitude -- for a batch of clips of one length.  A length of the form 2^a 3^b 5^c 7^d (>= 16:
 * sample counts such as 160 000, 441 000, 480 000) is transformed DIRECTLY by the library's mixed-radix Stockham FFT
 * (radix 16 / 8 / 4 / 2 / 9 / 3 / 25 / 5 / 49 / 7 passes, pairs of radix-16 or radix-25 passes fused into one
 * launch): mm_hilbert_fft_size() == n.  Any other length goes through Bluestein's chirp-z identity over a
 * power-of-two FFT of mm_hilbert_fft_size() = M >= 2n - 1 points.  Transforms run in the clips' own precision like
 * scipy (dtype 0: float32 / complex64, 1: float64 / complex128).
 * Two clips share one complex transform (clip 2r in the real part, 2r + 1 in the imaginary part: the spectrum mask is
 * linear, so the masked inverse transform returns (a - H b) + i (H a + b), and with a and b at hand both analytic
 * signals follow); each clip is scaled by a power of two first (exact), so a quiet clip next to a loud one keeps its
 * own relative accuracy, and a clip of zeros comes out as zeros.  A call with a single clip transforms it alone.
 * mm_hilbert_create allocates and fills the constant device tables on the current device (and synchronises);
 * mm_hilbert_envelope is asynchronous on `stream`: d_x [rows][x_stride] -> d_env [rows][env_stride], rows <=
 * 65535 per call, workspace = mm_hilbert_workspace_bytes(rows) = two [ceil(rows / 2)][fft_size] complex buffers and the
 * clips' scale factors -- size it with that call, not from n (fft_size is n on the direct path, M on the Bluestein
 * path). */
typedef struct mm_hilbert mm_hilbert;
int mm_hilbert_create(int64_t n, int32_t dtype, mm_hilbert** out);
void mm_hilbert_destroy(mm_hilbert* h);
int64_t mm_hilbert_fft_size(const mm_hilbert* h);
size_t mm_hilbert_workspace_bytes(const mm_hilbert* h, int64_t rows);
int mm_hilbert_envelope(mm_hilbert* h, const void* d_x, int64_t rows, int64_t x_stride, void* d_env,
                        int64_t env_stride, void* d_workspace, size_t ws_bytes, void* stream);
/* rFFT of real float32 rows zero-padded to the plan's length n (a plan of dtype 0 whose length is even and 2 / 3 / 5 / 7-
 * smooth, i.e. mm_hilbert_fft_size() == n): d_x [rows][x_stride] with n_valid <= n samples each -> d_out complex64
 * [rows][n / 2 + 1] -- the trajectory rFFT (np.fft.rfft(mfcc, n, axis=-1), row A8 of the hot path) for clips with more
 * than 8192 frames, which mm_modspec_f32 does not take: one recording at the reference's default 1 ms step
 * (script/mfcc.py:296) is 10 001 frames per ten seconds.  rows <= 65535 per call; asynchronous on `stream`. */
size_t mm_hilbert_rfft_workspace_bytes(const mm_hilbert* h, int64_t rows);
int mm_hilbert_rfft_f32(mm_hilbert* h, const float* d_x, int64_t rows, int64_t x_stride, int64_t n_valid, float* d_out,
                        void* d_ws, size_t ws_bytes, void* stream);

/* Input side (row N4): what librosa.load(path, sr=sigSr, mono=False) does before the hot path
 * (script/mfcc.py:284,373).
 * mm_pcm_decode_f32: interleaved little-endian PCM frames (device copy of a WAV data chunk) -> planar
 *   float32 [channels][out_stride]; fmt 1 = u8, 2 = s16, 3 = packed s24, 4 = s32, 5 = f32, 6 = f64; scaled to
 *   [-1, 1) as libsndfile does.
 * mm_resample_f32: rational-ratio (L / M) polyphase FIR sample-rate conversion, zero-phase, n_out =
 *   ceil(n_in * L / M) as librosa.resample returns; d_taps = DEVICE pointer to the low-pass taps (DC gain L) in
 *   OUTPUT-phase order, records of four: taps[j / 4][t][j % 4] = h[ph_t + j * L] with ph_t = (t * M + half_len)
 *   mod L the polyphase branch of the t-th output of a period (taps_per_phase a multiple of 4, zero padded; 16-byte
 *   aligned), half_len = (len(h) - 1) / 2; rows <= 65535.  The host
 *   designs h (modulation_mfcc_amd/audio_io.py: a Kaiser-windowed sinc of soxr-HQ class: pass band to 0.913 of
 *   the lower Nyquist, > 120 dB stop band); the reference's resampler is soxr_hq, whose coefficients are not
 *   public API, so outputs agree with it to the quality of both filters, not bit for bit. */
int mm_pcm_decode_f32(const void* d_raw, int32_t fmt, int32_t channels, int64_t n_frames, float* d_out,
                      int64_t out_stride, void* stream);
int mm_resample_f32(const float* d_x, int64_t rows, int64_t n_in, int64_t x_stride, const float* d_taps,
                    int32_t L, int32_t M, int32_t taps_per_phase, int64_t half_len, float* d_y, int64_t n_out,
                    void* stream);
/* mm_resample_banded_f32: the same conversion as a banded GEMM on the matrix pipe (v_mfma_f32_16x16x4_f32: exact float32
 *   products, float32 accumulation in tap order; measured within 8e-7 of full scale of the float64 accumulation of mm_resample_f32) -- the
 *   default of audio_io.resample_batch.  Outputs m = q F + 16 b + r (F >= 16 a multiple of L: a period of outputs with
 *   the same taps; b < NB = ceil(F / 16); r < 16, 16 b + r < F) read the window x[q S + lo_min + lo_off[b] + k], S = F M / L, k < 4 ksteps, through
 *   d_atab [NB][ksteps / 4][64][4] (16-byte aligned; ksteps a multiple of 8): entry (b, ks / 4, lane, ks % 4) = the tap
 *   of row r = lane % 16 at window sample k = 32 (ks / 8) + KOFF[lane / 16] + ks % 8, KOFF = {0, 16, 8, 24} (zero
 *   outside the row's taps_per_phase taps); win = max_b lo_off[b] + 4 ksteps.  The host builds both tables
 *   (audio_io.banded_tables).  Returns MM_ERR_UNSUPPORTED when one tile of 16 periods does not fit the LDS
 *   (S + win > ~40 k samples): use mm_resample_f32 then. */
int mm_resample_banded_f32(const float* d_x, int64_t rows, int64_t n_in, int64_t x_stride, const float* d_atab,
                           const int32_t* d_lo_off, int32_t F, int32_t S, int32_t NB, int32_t ksteps, int32_t lo_min,
                           int32_t win, float* d_y, int64_t n_out, void* stream);

/* Measurement aid: float4 grid-stride device-to-device copy of n_floats (a multiple of 4; 16-byte
 * aligned pointers) on `stream` -- the practical HBM ceiling bench.py quotes beside the stage-isolated
 * rFFT figure.  No reference counterpart. */
int mm_devcopy_f32(const float* d_src, float* d_dst, int64_t n_floats, void* stream);

/* ---- per-kernel device timing (hipEvents on the launch stream) ------------------------- */
/* on = 0: off; 1: every stage; otherwise a mask with bit (MM_STAGE_x + 1) set for each stage to time
 * (two hipEventRecord per timed launch: timing fewer stages perturbs the stream less). */
int mm_timing_enable(mm_plan* plan, int on);
/* synchronises the recorded events; ms_sum[s]/count[s] = average launch duration of stage s;
 * resets the accumulators. Arrays of MM_NUM_STAGES. */
int mm_timing_read(mm_plan* plan, double* ms_sum, int64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* MODMFCC_H */
