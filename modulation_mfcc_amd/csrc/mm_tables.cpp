// Host-side constant tables for libmodmfcc: window, Slaney mel filterbank, DCT-II matrix,
// twiddles, Butterworth sections.  Everything is computed in float64 and rounded to float32
// exactly where the reference's third-party stack does:
//   window : scipy.signal.get_window('hann', win_length, fftbins=True), centred in n_fft
//   mel    : librosa.filters.mel(htk=False, norm='slaney', dtype=float32)
//   dct    : scipy.fftpack.dct(type=2, norm='ortho') written as a matrix
// These are what script/mfcc.py:387 reaches through librosa.feature.mfcc.
#include "mm_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace mm {

static const double kPi = 3.14159265358979323846;

int validate(const mm_config* c) {
  if (!c) return MM_ERR_INVALID_ARG;
  if (!(c->sr > 0.0)) return MM_ERR_INVALID_ARG;
  if (c->n_fft < 2) return MM_ERR_INVALID_ARG;
  if (c->n_fft > 8192) return MM_ERR_UNSUPPORTED;     // any integer up to 8192 (librosa: any n_fft >= win_length)
  if (c->win_length < 1 || c->win_length > c->n_fft) return MM_ERR_INVALID_ARG;
  if (c->hop_length < 1) return MM_ERR_INVALID_ARG;
  if (c->n_mels < 1 || c->n_mels > 256) return MM_ERR_INVALID_ARG;
  if (c->n_mfcc < 1 || c->n_mfcc > c->n_mels) return MM_ERR_INVALID_ARG;
  if (!(c->fmin >= 0.0) || !(c->fmax > c->fmin)) return MM_ERR_INVALID_ARG;
  if (!(c->amin > 0.0f)) return MM_ERR_INVALID_ARG;
  if (c->center != 1) return MM_ERR_UNSUPPORTED;
  if (c->n_mod_fft != 0 &&
      (c->n_mod_fft < 32 || c->n_mod_fft > (1 << 24) || (c->n_mod_fft & (c->n_mod_fft - 1))))
    return MM_ERR_UNSUPPORTED;
  return MM_OK;
}

// periodic Hann: scipy general_cosine(M+1, [0.5, 0.5])[:-1] with fac = linspace(-pi, pi, M+1)
void build_window(const mm_config& c, float* out) {
  const int M = c.win_length, N = c.n_fft;
  const int lpad = (N - M) / 2;
  std::memset(out, 0, sizeof(float) * N);
  const double step = (kPi - (-kPi)) / (double)M;  // linspace(-pi, pi, M+1): div = M
  for (int n = 0; n < M; ++n) {
    double fac = (double)n * step + (-kPi);
    if (M == 1) fac = -kPi;
    double w = 0.0;
    w += 0.5 * std::cos(0.0 * fac);
    w += 0.5 * std::cos(1.0 * fac);
    if (M == 1) w = 1.0;  // scipy: M == 1 -> ones
    out[lpad + n] = (float)w;
  }
}

static const double kFsp = 200.0 / 3;
static const double kMinLogHz = 1000.0;
static const double kMinLogMel = (1000.0 - 0.0) / (200.0 / 3);
static double logstep() { return std::log(6.4) / 27.0; }

static double hz_to_mel(double f) {
  double mel = (f - 0.0) / kFsp;
  if (f >= kMinLogHz) mel = kMinLogMel + std::log(f / kMinLogHz) / logstep();
  return mel;
}
static double mel_to_hz(double m) {
  double f = 0.0 + kFsp * m;
  if (m >= kMinLogMel) f = kMinLogHz * std::exp(logstep() * (m - kMinLogMel));
  return f;
}

void build_mel(const mm_config& c, float* out) {
  const int n_mels = c.n_mels, n_bins = c.n_fft / 2 + 1;
  std::vector<double> mel_f(n_mels + 2), fftf(n_bins);
  const double min_mel = hz_to_mel(c.fmin), max_mel = hz_to_mel(c.fmax);
  const int npts = n_mels + 2;
  const double step = (max_mel - min_mel) / (double)(npts - 1);
  for (int i = 0; i < npts; ++i) mel_f[i] = mel_to_hz((double)i * step + min_mel);
  mel_f[npts - 1] = mel_to_hz(max_mel);
  const double val = 1.0 / ((double)c.n_fft * (1.0 / c.sr));  // np.fft.rfftfreq
  for (int k = 0; k < n_bins; ++k) fftf[k] = (double)k * val;
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int k = 0; k < n_bins; ++k) {
      const double lower = -(mel_f[i] - fftf[k]) / fd0;
      const double upper = (mel_f[i + 2] - fftf[k]) / fd1;
      const float w32 = (float)std::max(0.0, std::min(lower, upper));
      out[(size_t)i * n_bins + k] = (float)((double)w32 * enorm);  // float32 *= float64
    }
  }
}

void build_dct(const mm_config& c, float* out) {
  const int N = c.n_mels;
  for (int k = 0; k < c.n_mfcc; ++k) {
    const double f = (k == 0) ? std::sqrt(1.0 / (4.0 * N)) : std::sqrt(1.0 / (2.0 * N));
    for (int m = 0; m < N; ++m)
      out[(size_t)k * N + m] = (float)(2.0 * f * std::cos(kPi * k * (2.0 * m + 1.0) / (2.0 * N)));
  }
}

// tw[k] = exp(-2 pi i k / n), k < n (full circle), float2 interleaved
void build_twiddles(int n, float* out) {
  for (int k = 0; k < n; ++k) {
    const double a = -2.0 * kPi * (double)k / (double)n;
    out[2 * k] = (float)std::cos(a);
    out[2 * k + 1] = (float)std::sin(a);
  }
}

// Butterworth low-pass, digital, as SOS -- the algorithm of scipy.signal.butter(N, Wn, 'low',
// output='sos'): buttap -> lp2lp_zpk(warped) -> bilinear_zpk(fs=2) -> zpk2sos(pairing='nearest').
// For a low-pass Butterworth all zeros sit at z = -1, so zpk2sos reduces to ordering the pole
// pairs: scipy emits sections with the poles CLOSEST to the unit circle LAST, and for odd
// order the real pole shares the first section (b = [1, 1, 0]).
int build_butter_sos(int order, double wn, double* sos) {
  if (order < 1 || order > 64 || !(wn > 0.0) || !(wn < 1.0) || !sos) return MM_ERR_INVALID_ARG;
  typedef std::complex<double> cd;
  const double fs = 2.0;
  const double warped = 2.0 * fs * std::tan(kPi * wn / fs);
  std::vector<cd> p(order);
  for (int i = 0; i < order; ++i) {
    const int m = -order + 1 + 2 * i;  // buttap: m = arange(-N+1, N, 2)
    cd pa = -std::exp(cd(0.0, kPi * m / (2.0 * order)));
    pa *= warped;                               // lp2lp_zpk
    p[i] = (2.0 * fs + pa) / (2.0 * fs - pa);   // bilinear_zpk
  }
  // gain: k_analog = warped^N ; k_digital = k * real(prod(1)/prod(fs2 - p_analog)) ; all zeros at -1
  cd den(1.0, 0.0);
  for (int i = 0; i < order; ++i) {
    const int m = -order + 1 + 2 * i;
    cd pa = -std::exp(cd(0.0, kPi * m / (2.0 * order)));
    pa *= warped;
    den *= (2.0 * fs - pa);
  }
  const double k = std::pow(warped, order) * (cd(1.0, 0.0) / den).real();
  // collect one pole per conjugate pair (imag >= 0) plus the real pole for odd order
  std::vector<cd> pairs;
  bool has_real = false;
  cd real_pole(0, 0);
  for (int i = 0; i < order; ++i) {
    if (std::fabs(p[i].imag()) < 1e-14 * std::max(1.0, std::abs(p[i]))) {
      has_real = true;
      real_pole = cd(p[i].real(), 0.0);
    } else if (p[i].imag() > 0) {
      pairs.push_back(p[i]);
    }
  }
  // distance to unit circle ascending = closest first; scipy fills from the LAST section backwards
  std::sort(pairs.begin(), pairs.end(),
            [](const cd& a, const cd& b) { return std::fabs(1.0 - std::abs(a)) < std::fabs(1.0 - std::abs(b)); });
  const int n_sec = (order + 1) / 2;
  std::vector<cd> sec_pole(n_sec);
  std::vector<int> sec_real(n_sec, 0);
  int si = n_sec - 1;
  size_t pi_ = 0;
  if (has_real) {
    // for odd order the single real pole forms the FIRST emitted section (index 0)
    for (; si >= 1; --si) sec_pole[si] = pairs[pi_++];
    sec_pole[0] = real_pole;
    sec_real[0] = 1;
  } else {
    for (; si >= 0; --si) sec_pole[si] = pairs[pi_++];
  }
  // zeros: `order` zeros at z = -1.  For odd order scipy's zpk2sos pads one pole and one zero at
  // the origin and pairs 'nearest', visiting poles worst-first (last section first): the first
  // complex pair that lies closer to 0 than to -1 (Re p > -1/2) takes the origin zero and gets
  // the numerator [1, 1, 0]; if none does, the real-pole section (index 0) gets it.
  int single_zero_sec = -1;
  if (has_real) {
    single_zero_sec = 0;
    for (int s = n_sec - 1; s >= 1; --s)
      if (std::abs(sec_pole[s]) < std::abs(sec_pole[s] + 1.0)) { single_zero_sec = s; break; }
  }
  for (int s = 0; s < n_sec; ++s) {
    double* r = sos + 6 * s;
    const bool single_zero = (s == single_zero_sec);
    r[0] = 1.0; r[1] = single_zero ? 1.0 : 2.0; r[2] = single_zero ? 0.0 : 1.0;
    if (sec_real[s]) {
      r[3] = 1.0; r[4] = -sec_pole[s].real(); r[5] = 0.0;
    } else {
      r[3] = 1.0; r[4] = -2.0 * sec_pole[s].real(); r[5] = std::norm(sec_pole[s]);
    }
  }
  sos[0] *= k; sos[1] *= k; sos[2] *= k;  // scipy: sos[0, :3] *= k
  return n_sec;
}

// CSR view of the (1-5 % dense) mel matrix: every filter is a contiguous run of bins.
void build_mel_csr(const mm_config& c, const float* dense, MelCsr* csr) {
  const int n_mels = c.n_mels, n_bins = c.n_fft / 2 + 1;
  csr->start.assign(n_mels, 0);
  csr->len.assign(n_mels, 0);
  csr->off.assign(n_mels, 0);
  csr->w.clear();
  for (int m = 0; m < n_mels; ++m) {
    int lo = n_bins, hi = -1;
    for (int k = 0; k < n_bins; ++k)
      if (dense[(size_t)m * n_bins + k] != 0.0f) { lo = std::min(lo, k); hi = std::max(hi, k); }
    csr->off[m] = (int)csr->w.size();
    if (hi >= lo) {
      csr->start[m] = lo;
      csr->len[m] = hi - lo + 1;
      for (int k = lo; k <= hi; ++k) csr->w.push_back(dense[(size_t)m * n_bins + k]);
    }
  }
  if (csr->w.empty()) csr->w.push_back(0.0f);
}

// Returns false when the matrix does not have the <= 2 adjacent filters per bin structure
// (cannot happen for triangular filters with shared edges; checked anyway).
bool build_mel_sweep(const mm_config& c, const float* dense, int n_waves, MelSweep* out, const double* weights) {
  const int n_mels = c.n_mels, n_bins = c.n_fft / 2 + 1;
  out->wlo.assign(n_bins, 0.0f);
  out->whi.assign(n_bins, 0.0f);
  out->d.assign(n_bins, -1);
  int prev = -1;
  for (int k = 0; k < n_bins; ++k) {
    int nz[3], cnt = 0;
    for (int m = 0; m < n_mels && cnt < 3; ++m)
      if (dense[(size_t)m * n_bins + k] != 0.0f) nz[cnt++] = m;
    int d = prev;
    if (cnt == 1) {
      const int m = nz[0];
      if (m - 1 >= prev) { d = m - 1; out->whi[k] = dense[(size_t)m * n_bins + k]; }
      else if (m >= prev) { d = m; out->wlo[k] = dense[(size_t)m * n_bins + k]; }
      else return false;
    } else if (cnt == 2) {
      if (nz[1] != nz[0] + 1 || nz[0] < prev) return false;
      d = nz[0];
      out->wlo[k] = dense[(size_t)nz[0] * n_bins + k];
      out->whi[k] = dense[(size_t)nz[1] * n_bins + k];
    } else if (cnt > 2) {
      return false;
    }
    out->d[k] = d;
    prev = d;
  }
  // partition the filters over the waves: cost = bins swept + 6 per emitted filter
  auto range_of = [&](int m0, int m1, int* kb, int* ke) {
    int b = n_bins, e = 0;
    for (int k = 0; k < n_bins; ++k)
      if (out->d[k] >= m0 - 1 && out->d[k] <= m1 - 1 && (out->wlo[k] != 0.0f || out->whi[k] != 0.0f)) {
        b = std::min(b, k);
        e = std::max(e, k + 1);
      }
    if (b >= e) { b = 0; e = 0; }
    *kb = b; *ke = e;
  };
  int kb_all, ke_all;
  range_of(0, n_mels, &kb_all, &ke_all);
  const double total = (double)(ke_all - kb_all) + 6.0 * n_mels;
  out->part.assign((size_t)n_waves * 4, 0);
  double wsum = 0.0, wcum = 0.0;
  for (int w = 0; w < n_waves; ++w) wsum += weights ? weights[w] : 1.0;
  int m = 0;
  for (int w = 0; w < n_waves; ++w) {
    wcum += weights ? weights[w] : 1.0;
    const int m0 = m;
    int m1 = m0;
    if (w == n_waves - 1) {
      m1 = n_mels;
    } else {
      const double target = total * wcum / wsum;
      // advance while the cumulative cost of [0, m1) stays below the target
      while (m1 < n_mels) {
        int kb, ke;
        range_of(0, m1 + 1, &kb, &ke);
        const double cum = (double)(ke - kb_all > 0 ? ke - kb_all : 0) + 6.0 * (m1 + 1);
        if (cum > target && m1 > m0) break;
        ++m1;
        if (cum > target) break;
      }
      const int left_waves = n_waves - 1 - w;
      if (n_mels - m1 < left_waves) m1 = std::max(m0, n_mels - left_waves);  // keep >= 1 each if possible
    }
    int kb, ke;
    range_of(m0, m1, &kb, &ke);
    if (m1 <= m0) { kb = 0; ke = 0; }
    out->part[w * 4 + 0] = kb;
    out->part[w * 4 + 1] = ke;
    out->part[w * 4 + 2] = m0;
    out->part[w * 4 + 3] = m1;
    m = m1;
  }
  return true;
}

void build_mel_runs(const mm_config& c, const MelSweep& sw, int n_waves, MelRuns* out) {
  const int n_bins = c.n_fft / 2 + 1;
  out->hdr.clear(); out->grp.clear();
  out->part.assign((size_t)n_waves * 4, 0);
  for (int w = 0; w < n_waves; ++w) {
    const int m0 = sw.part[w * 4 + 2], m1 = sw.part[w * 4 + 3];
    out->part[w * 4 + 0] = (int)(out->hdr.size() / 4);
    out->part[w * 4 + 2] = m0;
    out->part[w * 4 + 3] = m1;
    if (m1 > m0) {
      for (int d = m0 - 1; d <= m1 - 1; ++d) {
        int ks = n_bins, ke = 0;
        for (int k = 0; k < n_bins; ++k)
          if (sw.d[k] == d && (sw.wlo[k] != 0.0f || sw.whi[k] != 0.0f)) { ks = std::min(ks, k); ke = std::max(ke, k + 1); }
        int g0 = 0, g1 = 0;
        if (ke > ks) { g0 = ks / 4; g1 = (ke + 3) / 4; }
        out->hdr.push_back(4 * g0);
        out->hdr.push_back(g1 - g0);
        out->hdr.push_back((int)(out->grp.size() / 8));
        out->hdr.push_back(d);
        for (int g = g0; g < g1; ++g) {
          float wl[4], wh[4];
          for (int u = 0; u < 4; ++u) {
            const int k = 4 * g + u;
            const bool in = (k >= ks && k < ke && k < n_bins && sw.d[k] == d);
            wl[u] = in ? sw.wlo[k] : 0.0f;
            wh[u] = in ? sw.whi[k] : 0.0f;
            // filter m0-1 (first run) and filter m1 (last run) belong to other waves
            if (d == m0 - 1) wl[u] = 0.0f;
            if (d == m1 - 1) wh[u] = 0.0f;
          }
          for (int u = 0; u < 4; ++u) out->grp.push_back(wl[u]);
          for (int u = 0; u < 4; ++u) out->grp.push_back(wh[u]);
        }
      }
    }
    out->part[w * 4 + 1] = (int)(out->hdr.size() / 4);
  }
}

}  // namespace mm
