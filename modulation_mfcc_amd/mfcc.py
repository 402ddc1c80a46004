"""Drop-in for the numeric functions of the reference's ``script/mfcc.py``.

Same names, argument meaning, positional-only / keyword-only split, return values and error
behaviour as the reference (script/mfcc.py:29-39, 262-264, 291-311), so the Qt UI's call sites
(script/main.py:750-769, 1049-1066) work unchanged after ``from modulation_mfcc_amd.mfcc import
get_MFCCS_change, load_channel``.  The librosa.feature.mfcc call at script/mfcc.py:387 is replaced
by the HIP path (``MfccPlan.mfcc`` -> libmodmfcc.so); there is no CPU fallback for it.

Host side, as in the reference: filter DESIGN (scipy.signal.butter / firwin) and the 'fir' / 'sg'
branches of applyFilter (they are scipy calls in the reference too).  The MFCC-change tail
(script/mfcc.py:392-427) is in ``tail.py``.
"""
from __future__ import annotations

import numpy as np

from . import filters as _filters
from . import tail as _tail
from .plan import MfccConfig, get_plan

__all__ = ["applyFilter", "get_amplitude", "load_channel", "get_MFCCS_change", "mfcc_array"]

applyFilter = _filters.applyFilter


def _decode_wav(path):
    """PCM / float WAV -> (float32 [ch, n] in [-1, 1), sr).  Host-side input decoding (row N4)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        x = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    else:
        x = data.astype(np.float32)
    x = x.T if x.ndim == 2 else x
    return np.ascontiguousarray(x), float(sr)


def _load_audio(path, sr):
    """Stand-in for librosa.load(path, sr=sr, mono=False) (script/mfcc.py:284,373).

    librosa resamples with soxr_hq, which is not available offline; this uses a polyphase FIR
    (scipy.signal.resample_poly), so path-string inputs that need resampling match the reference
    only approximately.  Arrays passed directly (the batch API, the tests) are unaffected.
    """
    x, file_sr = _decode_wav(path)
    if sr is not None and float(sr) != file_sr:
        from fractions import Fraction
        from scipy.signal import resample_poly
        fr = Fraction(float(sr) / file_sr).limit_denominator(1000)
        x = resample_poly(x, fr.numerator, fr.denominator, axis=-1).astype(np.float32)
    return x


def load_channel(file_path: str, signal_sample_rate: float = 10_000, channel_nb: int = 0):
    """script/mfcc.py:262-289: every channel of the file at ``signal_sample_rate`` (the reference
    ignores ``channel_nb`` as well -- its channel pick is commented out)."""
    return _load_audio(file_path, signal_sample_rate)


def mfcc_array(y, cfg: MfccConfig) -> np.ndarray:
    """One clip through the HIP path: numpy [n] -> float32 [n_mfcc, T] (== librosa.feature.mfcc)."""
    import torch
    y = np.ascontiguousarray(np.asarray(y), dtype=np.float32)
    if y.ndim != 1:
        raise ValueError("expected a 1-D signal")
    plan = get_plan(cfg)
    d = torch.from_numpy(y).to(plan.device)
    return plan.mfcc(d)[0].cpu().numpy()


def get_amplitude(x, sr, /, *, method: str = "RMS", winLen: float = 0.1, hopLen: float = 0.01,
                  center: bool = True, outFilter=None, outFiltType: str = "low",
                  outFiltCutOff=[12], outFiltLen: int = 6, outFiltPolyOrd: int = 3):
    """script/mfcc.py:137-259 -- same function as calc.calculate_amplitude_envelope."""
    from .calc import calculate_amplitude_envelope
    return calculate_amplitude_envelope(x, sr, method=method, winLen=winLen, hopLen=hopLen,
                                        center=center, outFilter=outFilter, outFiltType=outFiltType,
                                        outFiltCutOff=outFiltCutOff, outFiltLen=outFiltLen,
                                        outFiltPolyOrd=outFiltPolyOrd)


def get_MFCCS_change(audioIn, sigSr, /, *, channelN: int = 0, tStep: float = 0.001,
                     winLen: float = 0.025, n_mfcc: int = 13, n_fft: int = 512, minFreq: int = 100,
                     maxFreq: int = 10000, removeFirst: int = 1, filtCutoff: int = 12,
                     filtOrd: int = 6, diffMethod: str = "grad", outFilter: str = "iir",
                     outFiltType: str = "low", outFiltCutOff=[None], outFiltLen: int = 6,
                     outFiltPolyOrd: int = 3):
    """Amount of change in the MFCCs over time -- script/mfcc.py:291-427.

    ``audioIn`` is a path or an array ([n] or [channels, n]); returns ``(totChange, T)``: the
    change curve and the time anchor of every frame, as the reference does.  The MFCCs come from
    the HIP kernels (float32 arithmetic; arrays of any float dtype are cast to float32).
    """
    signal = _load_audio(audioIn, sigSr) if isinstance(audioIn, str) else audioIn
    if np.ndim(signal) > 1:
        signal = signal[channelN, :]

    cfg = MfccConfig.from_reference_call(sigSr, tStep=tStep, winLen=winLen, n_mfcc=n_mfcc,
                                         n_fft=n_fft, minFreq=minFreq, maxFreq=maxFreq)
    if _tail.device_path_applies(diffMethod, outFilter):
        import torch
        y = np.ascontiguousarray(np.asarray(signal), dtype=np.float32)
        if y.ndim != 1:
            raise ValueError("expected a 1-D signal")
        plan = get_plan(cfg)
        coeffs_dev = plan.mfcc(torch.from_numpy(y).to(plan.device))
        anchors = _tail.time_anchors(coeffs_dev.shape[2], tStep, winLen)
        change = _tail.mfcc_change_device(plan, coeffs_dev, tStep=tStep, removeFirst=removeFirst,
                                          filtCutoff=filtCutoff, filtOrd=filtOrd, diffMethod=diffMethod,
                                          outFilter=outFilter,
                                          outFiltType=outFiltType, outFiltCutOff=outFiltCutOff,
                                          outFiltLen=outFiltLen)[0].cpu().numpy()
        return change, anchors
    coeffs = mfcc_array(signal, cfg)
    anchors = _tail.time_anchors(coeffs.shape[1], tStep, winLen)
    change = _tail.mfcc_change(coeffs, tStep=tStep, removeFirst=removeFirst, filtCutoff=filtCutoff,
                               filtOrd=filtOrd, diffMethod=diffMethod, outFilter=outFilter,
                               outFiltType=outFiltType, outFiltCutOff=outFiltCutOff,
                               outFiltLen=outFiltLen, outFiltPolyOrd=outFiltPolyOrd)
    return change, anchors
