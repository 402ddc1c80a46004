"""CPU oracle for the MFCC + modulation-spectrum hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The shipped package (``modulation_mfcc_amd``) never
imports it and has no CPU fallback: its ops raise when the HIP library is absent.

PARITY UNPINNED.  The reference (aaron-randreth/modulation-mfcc) delegates the
whole MFCC chain to ``librosa.feature.mfcc`` (script/mfcc.py:387).  librosa is an
un-vendored, un-pinned dependency (requirements.txt:3) that is not installed in
this image; the reference ships no tests, fixtures or golden vectors for this
path, and script/mfcc.py itself cannot be imported (``from typing import
override`` needs Python >= 3.12, script/mfcc.py:1).  This file therefore restates
librosa's published algorithm (0.10.x semantics, the line the reference's
``numpy<2`` pin resolves to) and anchors it on the reference's own call site
arguments.  It was cross-checked in the build container against an independent
librosa-compatible implementation (``transformers.audio_utils``), see
``oracle/crosscheck_transformers.py``; that check is evidence, not a pin.

Every function cites the reference line whose behaviour it follows; where the
arithmetic lives inside librosa the docstring says ``[librosa, not in tree]``.
scipy pieces the reference itself calls (butter, sosfiltfilt, savgol_filter,
fftpack.dct, get_window) are CALLED here, not restated.
"""
from __future__ import annotations

import numpy as np
import scipy.fftpack
import scipy.signal

__all__ = [
    "OracleConfig", "hz_to_mel", "mel_to_hz", "mel_frequencies", "mel_filterbank",
    "hann_window_padded", "num_frames", "frame_signal", "stft_power", "rfft_rows",
    "mel_power", "power_to_db", "mfcc_from_logmel", "mfcc", "modspec",
    "mfcc_change_tail", "get_MFCCS_change", "time_anchors", "synth_clip",
    "rms_envelope",
]


# --------------------------------------------------------------------------------------
# Slaney mel scale and filterbank  [librosa.filters.mel / librosa.core.convert, not in tree]
# (reached from script/mfcc.py:387 with fmin=minFreq, fmax=maxFreq, n_mels default 128)
# --------------------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = (_MIN_LOG_HZ - 0.0) / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    mels = (f - 0.0) / _F_SP
    if f.ndim:
        log_t = f >= _MIN_LOG_HZ
        mels[log_t] = _MIN_LOG_MEL + np.log(f[log_t] / _MIN_LOG_HZ) / _LOGSTEP
    elif f >= _MIN_LOG_HZ:
        mels = _MIN_LOG_MEL + np.log(f / _MIN_LOG_HZ) / _LOGSTEP
    return mels


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    freqs = 0.0 + _F_SP * m
    if m.ndim:
        log_t = m >= _MIN_LOG_MEL
        freqs[log_t] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m[log_t] - _MIN_LOG_MEL))
    elif m >= _MIN_LOG_MEL:
        freqs = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return freqs


def mel_frequencies(n_mels, fmin, fmax):
    min_mel = hz_to_mel(fmin)
    max_mel = hz_to_mel(fmax)
    mels = np.linspace(min_mel, max_mel, n_mels)
    return mel_to_hz(mels)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """Slaney-scale, Slaney-area-normalised triangular filterbank, float32 [n_mels, 1+n_fft//2].

    [librosa.filters.mel(htk=False, norm='slaney', dtype=float32), not in tree].
    The weights are computed in float64, stored to float32, and the Slaney
    normalisation is applied by an in-place float32 *= float64 (double rounding),
    which this restatement reproduces.
    """
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    n_bins = int(1 + n_fft // 2)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# --------------------------------------------------------------------------------------
# Framing + window + rFFT + power   [librosa.stft / _spectrogram, not in tree]
# --------------------------------------------------------------------------------------
def hann_window_padded(win_length, n_fft):
    """Periodic Hann(win_length), float64, zero-padded centrally to n_fft (A1)."""
    w = scipy.signal.get_window("hann", int(win_length), fftbins=True)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft, dtype=np.float64)
    out[lpad:lpad + win_length] = w
    return out


def num_frames(n_samples, hop_length):
    """center=True frame count, 1 + n // hop."""
    return 1 + int(n_samples) // int(hop_length)


def frame_signal(y, n_fft, hop_length, preemph=0.0):
    """[T, n_fft] view of the centre-padded signal (zeros, pad_mode='constant').

    ``preemph`` (build extension, default 0 = reference behaviour): y[n] - a*y[n-1],
    y[-1] := 0, applied in float32 before padding.
    """
    y = np.asarray(y)
    if preemph:
        a = np.float32(preemph)
        y = y.astype(np.float32)
        y = np.concatenate([y[:1], y[1:] - a * y[:-1]]).astype(np.float32)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    T = 1 + (yp.shape[0] - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(T)[:, None]
    return yp[idx]


def stft_power(y, n_fft, hop_length, win_length, preemph=0.0, fft_dtype=np.float64):
    """|STFT|^2, float32 [T, 1+n_fft/2] (frame-major; librosa's is the transpose).

    librosa multiplies float32 frames by the float64 window (-> float64), takes
    np.fft.rfft (float64 under the reference's numpy<2 pin) and stores complex64;
    then np.abs(S)**2 in float32.  ``fft_dtype=np.float32`` gives the numpy>=2 /
    scipy.fft single-precision variant (differs by ~1e-7 relative).
    """
    y = np.asarray(y, dtype=np.float32)
    frames = frame_signal(y, n_fft, hop_length, preemph)
    win = hann_window_padded(win_length, n_fft)
    if fft_dtype == np.float32:
        spec = scipy.fft.rfft((frames * win.astype(np.float32)).astype(np.float32), axis=-1)
    else:
        spec = np.fft.rfft(frames.astype(np.float64) * win, axis=-1)
    spec = spec.astype(np.complex64)
    return (np.abs(spec) ** 2).astype(np.float32)


def rfft_rows(x, n):
    """Stage-isolated batched rFFT: rows [R, L<=n] float32 -> complex64 [R, n/2+1]."""
    x = np.asarray(x, dtype=np.float32)
    return np.fft.rfft(x.astype(np.float64), n=n, axis=-1).astype(np.complex64)


def mel_power(power, mel_w):
    """[T, n_bins] x [n_mels, n_bins] -> [T, n_mels] float32 (librosa einsum '...ft,mf->...mt')."""
    return np.einsum("tf,mf->tm", power.astype(np.float32), mel_w.astype(np.float32),
                     optimize=True).astype(np.float32)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """[librosa.power_to_db(ref=1.0), not in tree]; the max is over the WHOLE clip array."""
    S = np.asarray(S, dtype=np.float32)
    log_spec = 10.0 * np.log10(np.maximum(np.float32(amin), S))
    log_spec = log_spec - np.float32(10.0 * np.log10(max(amin, 1.0)))
    log_spec = log_spec.astype(np.float32)
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - np.float32(top_db))
    return log_spec


def mfcc_from_logmel(logmel_tm, n_mfcc):
    """DCT-II ortho along the mel axis, first n_mfcc rows; returns [n_mfcc, T] float32."""
    M = scipy.fftpack.dct(logmel_tm.T.astype(np.float32), axis=-2, type=2, norm="ortho")
    return np.ascontiguousarray(M[:n_mfcc, :]).astype(np.float32)


class OracleConfig:
    """Same knobs as the C ABI's mm_config (include/modmfcc.h)."""

    def __init__(self, sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=128,
                 n_mfcc=13, fmin=100.0, fmax=10000.0, preemph=0.0, top_db=80.0,
                 amin=1e-10, n_mod_fft=0):
        self.sr = sr
        self.n_fft = n_fft
        self.win_length = win_length
        self.hop_length = hop_length
        self.n_mels = n_mels
        self.n_mfcc = n_mfcc
        self.fmin = fmin
        self.fmax = fmax
        self.preemph = preemph
        self.top_db = top_db
        self.amin = amin
        self.n_mod_fft = n_mod_fft

    @classmethod
    def from_reference_call(cls, sigSr, tStep, winLen, n_mfcc, n_fft, minFreq, maxFreq):
        """Host arithmetic of script/mfcc.py:382-387 (int() truncation, n_mels=128 default)."""
        return cls(sr=sigSr, n_fft=n_fft, win_length=int(winLen * sigSr),
                   hop_length=int(tStep * sigSr), n_mels=128, n_mfcc=n_mfcc,
                   fmin=minFreq, fmax=maxFreq)


def logmel_unclamped(y, cfg, fft_dtype=np.float64):
    """10*log10(max(amin, mel)) [T, n_mels] before the top_db clamp (kernel-1 output)."""
    P = stft_power(y, cfg.n_fft, cfg.hop_length, cfg.win_length, cfg.preemph, fft_dtype)
    W = mel_filterbank(cfg.sr, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax)
    return power_to_db(mel_power(P, W), amin=cfg.amin, top_db=None)


def mfcc(y, cfg, fft_dtype=np.float64):
    """librosa.feature.mfcc restated: [n_mfcc, T] float32 (script/mfcc.py:387)."""
    P = stft_power(y, cfg.n_fft, cfg.hop_length, cfg.win_length, cfg.preemph, fft_dtype)
    W = mel_filterbank(cfg.sr, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax)
    S = power_to_db(mel_power(P, W), amin=cfg.amin, top_db=cfg.top_db)
    return mfcc_from_logmel(S, cfg.n_mfcc)


def mfcc_f64(y, cfg):
    """librosa.feature.mfcc for a FLOAT64 signal: [n_mfcc, T] float64.

    librosa keeps the precision of its input (util.dtype_r2c: a float64 signal gives a complex128 STFT and
    float64 power, mel, dB and DCT arrays; the mel basis keeps its float32 VALUES).  librosa.load returns float32,
    so the reference's own path never takes this branch -- an ndarray caller can.  The build computes every
    input in float32 (stated in modulation_mfcc_amd/mfcc.py); the tests bound the difference with this function."""
    y = np.asarray(y, dtype=np.float64)
    if cfg.preemph:
        y = np.concatenate([y[:1], y[1:] - cfg.preemph * y[:-1]])
    frames = frame_signal(y, cfg.n_fft, cfg.hop_length, 0.0)
    win = hann_window_padded(cfg.win_length, cfg.n_fft)
    P = np.abs(np.fft.rfft(frames * win, axis=-1)) ** 2
    W = mel_filterbank(cfg.sr, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax).astype(np.float64)
    S = np.einsum("tf,mf->tm", P, W, optimize=True)
    log_spec = 10.0 * np.log10(np.maximum(cfg.amin, S)) - 10.0 * np.log10(max(cfg.amin, 1.0))
    if cfg.top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - cfg.top_db)
    M = scipy.fftpack.dct(log_spec.T, axis=-2, type=2, norm="ortho")
    return np.ascontiguousarray(M[:cfg.n_mfcc, :])


def next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


def modspec(mfcc_kt, n_mod_fft=0):
    """Row A8 (build-defined, no reference site): rFFT of every coefficient's time
    trajectory, zero-padded to n_mod_fft (default next_pow2(T)); complex64 [n_mfcc, n/2+1]."""
    T = mfcc_kt.shape[-1]
    n = int(n_mod_fft) if n_mod_fft else next_pow2(T)
    return np.fft.rfft(mfcc_kt.astype(np.float64), n=n, axis=-1).astype(np.complex64)


# --------------------------------------------------------------------------------------
# The reference's own post-processing around the librosa call (script/mfcc.py:390-427)
# --------------------------------------------------------------------------------------
def time_anchors(n_frames, tStep, winLen):
    """script/mfcc.py:390."""
    return np.round(np.multiply(np.arange(1, n_frames + 1), tStep) + winLen / 2, 4)


def apply_filter(x, sr, *, filt="iir", cutOff=(None,), filtLen=6, filtType="low", polyOrd=3):
    """script/mfcc.py:29-135 (iir / fir / sg branches, same exceptions)."""
    if (filt is None) | (cutOff is None):
        if cutOff is None:
            raise Exception("Cannot apply filter without specifying a cut Off freq. (CutOff is None).")
        raise Exception("Cannot apply filter without specifying a filter method among iir, fir and  sg (filt is None).")
    filtTypes = np.array(["bandpass", "lowpass", "highpass"])
    try:
        filtType = filtTypes[np.argwhere([s.startswith(filtType) for s in filtTypes]).flatten()][0]
    except Exception:
        raise Exception("filtType must be one among: lowpass, highpass, bandpass. Partial matches allowed.")
    if any((sr / 2) <= np.array(cutOff)):
        raise Exception("Cut off frequencies must be smaller than the half of the sampling freq. of the signal submitted to the filter")
    if (len(cutOff) > 0) & (any(np.diff(cutOff) <= 0)):
        raise Exception("If two cut off freqs are provided: cutOff[0]<cutOff[1]")
    cutOff = np.array(cutOff)
    ok = ((len(cutOff) == 1) and (filtType in ("lowpass", "highpass"))) or \
         ((len(cutOff) == 2) and (filtType == "bandpass"))
    if filt == "iir":
        if not ok:
            raise Exception("only one or two cut off frequencies allowed. If two freqs are provided, filtType must be bandpass")
        sos = scipy.signal.butter(filtLen, cutOff / (sr / 2), btype=filtType, output="sos")
        return scipy.signal.sosfiltfilt(sos, x)
    if filt == "fir":
        if not ok:
            raise Exception("only one or two cut off frequencies allowed. If two freqs are provided, filtType must be bandpass")
        b = scipy.signal.firwin(filtLen, cutOff / (sr / 2), window=("kaiser", 7.4), pass_zero=filtType)
        return scipy.signal.filtfilt(b, 1, x)
    if filt == "sg":
        if len(cutOff) != 1:
            raise Exception("sg (savitsky Golay) filters can only be lowpass (one cutOff freq allowed)")
        return scipy.signal.savgol_filter(x, filtLen, polyOrd, deriv=0, mode="interp")
    raise Exception("unknown filt")


def mfcc_change_tail(myMfccs, *, tStep, removeFirst=1, filtCutoff=12, filtOrd=6,
                     diffMethod="grad", outFilter="iir", outFiltType="low",
                     outFiltCutOff=(None,), outFiltLen=6, outFiltPolyOrd=3):
    """script/mfcc.py:392-427: drop c0 -> Butterworth sosfiltfilt -> gradient -> norm -> filter."""
    if removeFirst:
        myMfccs = myMfccs[1:, :]
    cutOffNorm = filtCutoff / ((1 / tStep) / 2)
    sos = scipy.signal.butter(filtOrd, cutOffNorm, btype="low", output="sos")
    filt = scipy.signal.sosfiltfilt(sos, myMfccs)
    if diffMethod == "grad":
        d = np.gradient(filt, axis=1)
    else:
        d = scipy.signal.savgol_filter(filt, 3, 2, deriv=1, axis=1, mode="interp")
    tot = np.sqrt(np.sum(d ** 2, 0)) / np.shape(myMfccs)[0]
    if outFilter is None:
        tot = scipy.signal.sosfiltfilt(sos, tot)
    else:
        tot = apply_filter(tot, 1 / tStep, filt=outFilter, filtType=outFiltType,
                           cutOff=outFiltCutOff, filtLen=outFiltLen, polyOrd=outFiltPolyOrd)
    return tot


def get_MFCCS_change(audioIn, sigSr, *, channelN=0, tStep=0.001, winLen=0.025, n_mfcc=13,
                     n_fft=512, minFreq=100, maxFreq=10000, removeFirst=1, filtCutoff=12,
                     filtOrd=6, diffMethod="grad", outFilter="iir", outFiltType="low",
                     outFiltCutOff=(None,), outFiltLen=6, outFiltPolyOrd=3):
    """script/mfcc.py:291-427 for ndarray input (file decoding is out of scope, row N4)."""
    myAudio = np.asarray(audioIn)
    y = myAudio[channelN, :] if myAudio.ndim > 1 else myAudio
    cfg = OracleConfig.from_reference_call(sigSr, tStep, winLen, n_mfcc, n_fft, minFreq, maxFreq)
    m = mfcc(y, cfg)
    T = time_anchors(m.shape[1], tStep, winLen)
    tot = mfcc_change_tail(m, tStep=tStep, removeFirst=removeFirst, filtCutoff=filtCutoff,
                           filtOrd=filtOrd, diffMethod=diffMethod, outFilter=outFilter,
                           outFiltType=outFiltType, outFiltCutOff=outFiltCutOff,
                           outFiltLen=outFiltLen, outFiltPolyOrd=outFiltPolyOrd)
    return tot, T


def rms_envelope(x, frame_length, hop_length, center=True):
    """[librosa.feature.rms(pad_mode='constant'), not in tree] (script/calc.py:331)."""
    x = np.asarray(x, dtype=np.float32)
    if center:
        pad = frame_length // 2
        x = np.pad(x, (pad, pad), mode="constant")
    T = 1 + (x.shape[0] - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(T)[:, None]
    fr = x[idx]
    return np.sqrt(np.mean(np.abs(fr) ** 2, axis=-1)).astype(np.float32)


# --------------------------------------------------------------------------------------
# Synthetic input; seed = clip index
# --------------------------------------------------------------------------------------
def synth_clip(seed, n_samples, sr, kind="am"):
    rng = np.random.default_rng(int(seed))
    t = np.arange(n_samples, dtype=np.float64) / sr
    if kind == "am":
        x = 0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 4 * t)) \
            + 0.05 * rng.standard_normal(n_samples)
    elif kind == "silence":
        x = np.zeros(n_samples)
    elif kind == "impulse":
        x = np.zeros(n_samples)
        x[n_samples // 3] = 1.0
    elif kind == "noise":
        x = rng.uniform(-1.0, 1.0, n_samples)
    elif kind == "quiet_tail":      # loud first half, near-silent second half: forces the top_db clamp
        x = 0.5 * rng.standard_normal(n_samples)
        x[n_samples // 2:] *= 1e-6
    elif kind == "chirp":
        x = 0.4 * np.sin(2 * np.pi * (100 + 0.5 * (sr / 2 - 200) * t / max(t[-1], 1e-9)) * t)
    else:
        raise ValueError(kind)
    return x.astype(np.float32)
