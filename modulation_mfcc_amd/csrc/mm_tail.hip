// libmodmfcc: the rows after the MFCCs (SURVEY.md 8(f) N1, N2 and the 'iir' filter of N3) -- the MFCC-change tail of
// script/mfcc.py:392-427 in its three device forms, scipy.signal.sosfiltfilt on batches of curves, the banded stencil
// operator behind get_velocity (script/calc.py:593-650) and the 'sg' / 'fir' output filters.  gfx950 only.
#include "mm_common.h"
#include "mm_plan.h"

#include "mm_change.hip.inc"
#include "mm_change_clip.hip.inc"
#include "mm_sos_rows.hip.inc"

extern "C" {

// scipy.signal.sosfilt_zi + the padlen rule of sosfiltfilt, host side
static int make_sosfilt(const double* sos, int n_sec, SosFilt* f) {
  if (n_sec < 0 || n_sec > MM_MAX_SEC || (n_sec > 0 && !sos)) return MM_ERR_INVALID_ARG;
  f->n_sec = n_sec;
  f->padlen = 0;
  if (n_sec == 0) return MM_OK;
  double scale = 1.0;
  int zb = 0, za = 0;
  for (int s = 0; s < n_sec; ++s) {
    const double* r = sos + 6 * s;
    if (r[3] == 0.0) return MM_ERR_INVALID_ARG;
    double b0 = r[0] / r[3], b1 = r[1] / r[3], b2 = r[2] / r[3], a1 = r[4] / r[3], a2 = r[5] / r[3];
    f->c[s][0] = b0; f->c[s][1] = b1; f->c[s][2] = b2; f->c[s][3] = 1.0; f->c[s][4] = a1; f->c[s][5] = a2;
    // lfilter_zi: (I - companion(a)^T) zi = b[1:] - a[1:] b0
    const double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
    const double den = 1.0 + a1 + a2;
    const double z0 = (B0 + B1) / den;
    f->zi[s][0] = scale * z0;
    f->zi[s][1] = scale * (B1 - a2 * z0);
    scale *= (b0 + b1 + b2) / (1.0 + a1 + a2);
    if (r[2] == 0.0) ++zb;
    if (r[5] == 0.0) ++za;
  }
  const int ntaps = 2 * n_sec + 1 - (zb < za ? zb : za);
  f->padlen = 3 * ntaps;
  return MM_OK;
}

static int change_pads(int n_sec1, const double* sos1, int n_sec2, const double* sos2, SosFilt* f1, SosFilt* f2) {
  int rc = make_sosfilt(sos1, n_sec1, f1);
  if (rc) return rc;
  if (n_sec1 < 1) return MM_ERR_INVALID_ARG;
  return make_sosfilt(sos2, n_sec2, f2);
}

static int64_t round64(int64_t v) { return (v + 63) / 64 * 64; }

// Upper bound over the three forms and every filter of up to MM_MAX_SEC sections (a caller that does not know its
// filters yet); mm_change_workspace_bytes_for() sizes the form a given call takes.
size_t mm_change_workspace_bytes(const mm_plan* p, int64_t batch, int64_t n_frames) {
  if (!p || batch < 1 || n_frames < 1) return 0;
  // two time-major buffers [T + 2 pad][columns padded to 64]; worst-case padding 3 * (2 * MM_MAX_SEC + 1)
  const int64_t padmax = 3 * (2 * MM_MAX_SEC + 1);
  const int64_t n = n_frames + 2 * padmax;
  const size_t tm = (size_t)n * (size_t)(round64(batch * p->cfg.n_mfcc) + round64(batch));
  // ... or the segmented rows' buffers (mm_sos_rows.hip.inc), whichever is larger
  const int pads = 3 * (2 * MM_CLIP_NS + 1);
  const size_t sg = chg_seg_workspace_doubles(batch, p->cfg.n_mfcc, n_frames, pads, pads);
  return std::max(tm, sg) * sizeof(double);
}

// Which of the three device forms a change-tail call takes, and the workspace (in doubles) THAT form needs -- one
// routine for the size query and for the call, so the two cannot disagree.
enum { MM_CHG_TIME_MAJOR = 0, MM_CHG_CLIP = 1, MM_CHG_SEGMENTED = 2 };
struct ChgForm { int form; size_t need; ClipShape cs; };
static ChgForm change_form(const mm_plan* p, int64_t batch, int64_t n_frames, int n_rows, const SosFilt& f1, const SosFilt& f2) {
  ChgForm r;
  const int64_t p1 = f1.padlen, p2 = f2.n_sec > 0 ? f2.padlen : 0;
  const int64_t n1 = n_frames + 2 * p1, n2 = n_frames + 2 * p2;
  r.cs = clip_shape(n_rows, n1, n2);
  const bool small_sec = f1.n_sec <= MM_CLIP_NS && f2.n_sec <= MM_CLIP_NS;
  // Long clips (one recording at a 1 ms step is 10 001 frames per ten seconds): the clip form holds fewer and fewer rows
  // at once and walks its groups one after the other -- from five groups on (about 5000 frames at 12 rows) a wave per
  // 1088 samples of a row (mm_sos_rows.hip.inc) is faster at every clip count (tools/chg_forms.py: 8001 frames 0.80 ms
  // against 0.11 - 0.45 ms for 1 - 256 clips; 4001 frames, three groups: 0.12 against 0.11 - 0.27 ms)
  bool few_long = r.cs.G >= 1 && (n_rows + r.cs.G - 1) / r.cs.G > 4;
#ifdef MM_DEV
  if (const char* e = getenv("MM_CHG_FORM")) few_long = e[0] == 's';      // side build only: A/B of the two forms (tools/chg_forms.py)
#endif
  if (!p->no_fuse_tail && small_sec && (r.cs.G < 1 || few_long)) {
    r.form = MM_CHG_SEGMENTED;
    r.need = chg_seg_workspace_doubles(batch, n_rows, n_frames, (int)p1, (int)p2);
  } else if (!p->no_fuse_tail && small_sec && r.cs.G >= 1) {
    r.form = MM_CHG_CLIP;
    r.need = (size_t)r.cs.tab_n;        // filter tables only (a few KB): the clip lives in LDS
  } else {
    r.form = MM_CHG_TIME_MAJOR;         // two time-major buffers [T + 2 pad][columns padded to 64]
    r.need = (size_t)n1 * (size_t)round64(batch * n_rows) + (size_t)n2 * (size_t)round64(batch);
  }
  return r;
}

size_t mm_change_workspace_bytes_for(const mm_plan* p, int64_t batch, int64_t n_frames, int32_t remove_first, const double* sos1,
                                     int32_t n_sec1, const double* sos2, int32_t n_sec2) {
  if (!p || batch < 1 || n_frames < 1 || remove_first < 0 || remove_first >= p->cfg.n_mfcc) return 0;
  SosFilt f1, f2;
  if (change_pads(n_sec1, sos1, n_sec2, sos2, &f1, &f2)) return 0;
  const int n_rows = p->cfg.n_mfcc - (remove_first ? 1 : 0);
  return std::max<size_t>(change_form(p, batch, n_frames, n_rows, f1, f2).need, 1) * sizeof(double);
}

int mm_mfcc_change_f64(mm_plan* p, const float* d_mfcc, int64_t batch, int64_t n_frames,
                       int32_t remove_first, int32_t diff_method, const double* sos1, int32_t n_sec1, const double* sos2,
                       int32_t n_sec2, double* d_change, void* d_ws, size_t ws_bytes, void* stream) {
  if (!p || !d_mfcc || !d_change || !d_ws || batch < 1 || n_frames < 1) return MM_ERR_INVALID_ARG;
  if (diff_method < 0 || diff_method > 1 || (diff_method == 1 && n_frames < 3)) return MM_ERR_INVALID_ARG;
  if (remove_first < 0 || remove_first >= p->cfg.n_mfcc) return MM_ERR_INVALID_ARG;
  if (batch > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  SosFilt f1, f2;
  int rc = change_pads(n_sec1, sos1, n_sec2, sos2, &f1, &f2);
  if (rc) return rc;
  // scipy: "The length of the input vector x must be greater than padlen"
  if (n_frames <= f1.padlen || n_frames <= f2.padlen) return MM_ERR_INVALID_ARG;
  ChangeParams q;
  q.mfcc = d_mfcc; q.n_frames = n_frames; q.batch = batch; q.n_mfcc = p->cfg.n_mfcc;
  q.first_row = remove_first ? 1 : 0; q.n_rows = q.n_mfcc - q.first_row;
  q.p1 = f1.padlen; q.p2 = f2.n_sec > 0 ? f2.padlen : 0; q.sg = diff_method;
  q.R = batch * q.n_rows; q.Rp = round64(q.R); q.Bp = round64(batch);
  const int64_t n1 = n_frames + 2 * q.p1, n2 = n_frames + 2 * q.p2;
  // the workspace of the form that runs (mm_change_workspace_bytes_for); mm_change_workspace_bytes is the bound over all forms
  const ChgForm cf = change_form(p, batch, n_frames, q.n_rows, f1, f2);
  if (ws_bytes < cf.need * sizeof(double)) return MM_ERR_WORKSPACE;
  q.ws1 = (double*)d_ws; q.ws2 = q.ws1 + n1 * q.Rp; q.out = d_change;
  const int64_t tblocks = (n_frames + 63) / 64;
  if (tblocks > 65535 || 2 * (int64_t)q.p1 * q.Rp / 256 + 1 > 0x7FFFFFFF || n_frames * q.Bp / 256 + 1 > 0x7FFFFFFF ||
      q.Bp / 64 > 65535 || n_frames > 0x7FFFFFFF)
    return MM_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  StageTimer tm(p, MM_STAGE_CHANGE, st);
  if (cf.form == MM_CHG_SEGMENTED) {       // mm_sos_rows.hip.inc: a wave per 1088 samples of a row
    rc = launch_chg_segmented(q, f1, f2, q.ws1, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  if (cf.form == MM_CHG_CLIP) {            // mm_change_clip.hip.inc: one launch, no workspace traffic
    const ClipShape& cs = cf.cs;
    const int ns = std::max(f1.n_sec, f2.n_sec);
    rc = ns <= 2 ? launch_chg_clip<2>(q, f1, f2, cs, n1, n2, q.ws1, st)
       : ns == 3 ? launch_chg_clip<3>(q, f1, f2, cs, n1, n2, q.ws1, st)
                 : launch_chg_clip<4>(q, f1, f2, cs, n1, n2, q.ws1, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  hipLaunchKernelGGL(chg_pack_kernel, dim3((unsigned)(q.Rp / 64), (unsigned)tblocks), dim3(256), 0, st, q);
  hipLaunchKernelGGL(chg_pad_kernel, dim3((unsigned)((2 * (int64_t)q.p1 * q.Rp + 255) / 256)), dim3(256), 0, st,
                     q.ws1, n_frames, q.p1, q.Rp, 1);
  launch_sos_any(f1, q.ws1, n1, q.Rp, st);
  if (q.n_rows <= MM_CHG_MAXROWS)
    hipLaunchKernelGGL(chg_norm_kernel, dim3((unsigned)n_frames, (unsigned)(q.Bp / 64)), dim3(256), 0, st, q);
  else
    hipLaunchKernelGGL(chg_norm_rows_kernel, dim3((unsigned)((n_frames * q.Bp + 255) / 256)), dim3(256), 0, st, q);
  if (f2.n_sec > 0) {
    hipLaunchKernelGGL(chg_pad_kernel, dim3((unsigned)((2 * (int64_t)q.p2 * q.Bp + 255) / 256)), dim3(256), 0, st,
                       q.ws2, n_frames, q.p2, q.Bp, 0);
    launch_sos_any(f2, q.ws2, n2, q.Bp, st);
  }
  hipLaunchKernelGGL(chg_unpack_kernel, dim3((unsigned)(q.Bp / 64), (unsigned)tblocks), dim3(256), 0, st, q);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

size_t mm_sosfiltfilt_workspace_bytes(int64_t rows, int64_t n) {
  if (rows < 1 || n < 1) return 0;
  // the larger of the two device forms: time-major [n + 2 pad][rows padded to 64] | segmented rows (mm_sos_rows.hip.inc)
  const size_t tm = (size_t)(n + 2 * 3 * (2 * MM_MAX_SEC + 1)) * (size_t)round64(rows);
  const size_t sg = seg_workspace_doubles(rows, n, 3 * (2 * MM_CLIP_NS + 1));
  return std::max(tm, sg) * sizeof(double);
}

// d_x (float64 rows) or d_xf (float32 rows: odd extension in float32 arithmetic, see odd_ext_f32)
static int sosfiltfilt_impl(const double* d_x, const float* d_xf, int64_t rows, int64_t n, int64_t x_stride, const double* sos,
                            int32_t n_sec, double* d_y, void* d_ws, size_t ws_bytes, void* stream) {
  if ((!d_x && !d_xf) || !d_y || !d_ws || rows < 1 || n < 1 || x_stride < n || n_sec < 1) return MM_ERR_INVALID_ARG;
  SosFilt f;
  int rc = make_sosfilt(sos, n_sec, &f);
  if (rc) return rc;
  if (n <= f.padlen) return MM_ERR_INVALID_ARG;      // scipy: "The length of the input vector x must be greater than padlen"
  if (ws_bytes < mm_sosfiltfilt_workspace_bytes(rows, n)) return MM_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (f.n_sec <= MM_CLIP_NS) {     // segmented rows: a wave per 1088 samples of a row, any length
    const SegSrc src = {d_x, x_stride, d_xf, 0, 0, 0};
    rc = launch_sos_rows_any(f, src, rows, n, d_y, (double*)d_ws, st);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return MM_OK;
  }
  const int64_t Wp = round64(rows), tb = (n + 63) / 64;
  if (tb > 65535 || Wp / 64 > 0x7FFFFFFF || 2 * (int64_t)f.padlen * Wp / 256 + 1 > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  double* ws = (double*)d_ws;
  if (d_xf)
    hipLaunchKernelGGL((sos_rows_pack_kernel<float>), dim3((unsigned)(Wp / 64), (unsigned)tb), dim3(256), 0, st, d_xf, rows, n,
                       x_stride, f.padlen, Wp, ws);
  else
    hipLaunchKernelGGL((sos_rows_pack_kernel<double>), dim3((unsigned)(Wp / 64), (unsigned)tb), dim3(256), 0, st, d_x, rows, n,
                       x_stride, f.padlen, Wp, ws);
  hipLaunchKernelGGL(chg_pad_kernel, dim3((unsigned)((2 * (int64_t)f.padlen * Wp + 255) / 256)), dim3(256), 0, st, ws, n,
                     f.padlen, Wp, d_xf ? 1 : 0);
  launch_sos_any(f, ws, n + 2 * f.padlen, Wp, st);
  hipLaunchKernelGGL(sos_rows_unpack_kernel, dim3((unsigned)(Wp / 64), (unsigned)tb), dim3(256), 0, st, ws, rows, n, f.padlen,
                     Wp, d_y);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

int mm_sosfiltfilt_f64(const double* d_x, int64_t rows, int64_t n, int64_t x_stride, const double* sos, int32_t n_sec,
                       double* d_y, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_x) return MM_ERR_INVALID_ARG;
  return sosfiltfilt_impl(d_x, nullptr, rows, n, x_stride, sos, n_sec, d_y, d_ws, ws_bytes, stream);
}

int mm_sosfiltfilt_f32_f64(const float* d_x, int64_t rows, int64_t n, int64_t x_stride, const double* sos, int32_t n_sec,
                           double* d_y, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_x) return MM_ERR_INVALID_ARG;
  return sosfiltfilt_impl(nullptr, d_x, rows, n, x_stride, sos, n_sec, d_y, d_ws, ws_bytes, stream);
}

int mm_stencil_f64(const mm_stencil* st, const double* d_x, int64_t rows, int64_t n, int64_t x_stride, double* d_y,
                   void* stream) {
  if (!st || !d_x || !d_y || rows < 1 || n < 1 || x_stride < n) return MM_ERR_INVALID_ARG;
  if (st->n_c < 1 || st->n_c > MM_ST_MAXW || st->n_edge < 0 || st->n_edge > MM_ST_MAXE || st->edge_w < 0 ||
      st->edge_w > MM_ST_MAXW || (st->n_edge > 0 && st->edge_w < 1) || st->den_c == 0.0 ||
      (st->n_edge > 0 && st->den_e == 0.0))
    return MM_ERR_INVALID_ARG;
  int lo = 0, hi = 0;
  for (int k = 0; k < st->n_c; ++k) { lo = std::min(lo, st->off[k]); hi = std::max(hi, st->off[k]); }
  // every interior output must find its taps inside the row, the edge rows their inputs
  if (n < 2 * (int64_t)st->n_edge || n < st->edge_w || -lo > st->n_edge || hi > st->n_edge) return MM_ERR_INVALID_ARG;
  const int64_t total = rows * n;
  if ((total + 255) / 256 > 0x7FFFFFFF) return MM_ERR_INVALID_ARG;
  hipLaunchKernelGGL(stencil_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *st, d_x,
                     rows, n, x_stride, d_y);
  HIP_TRY(hipGetLastError());
  return MM_OK;
}

}  // extern "C"
