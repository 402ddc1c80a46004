"""Drop-in for the numeric functions of the reference's ``script/mfcc.py``.

Same names, argument meaning, positional-only / keyword-only split, return values and error
behaviour as the reference (script/mfcc.py:29-39, 262-264, 291-311), so the Qt UI's call sites
(script/main.py:750-769, 1049-1066) work unchanged after ``from modulation_mfcc_amd.mfcc import
get_MFCCS_change, load_channel``.  The librosa.feature.mfcc call at script/mfcc.py:387 is replaced
by the HIP path (``MfccPlan.mfcc`` -> libmodmfcc.so); there is no CPU fallback for it.

Host side, as in the reference: filter DESIGN (scipy.signal.butter / firwin) and the 'fir' branch of
applyFilter (a scipy call in the reference too).  The MFCC-change tail (script/mfcc.py:392-427) is in
``tail.py``, the input side (WAVE decode + resampling, script/mfcc.py:284,373) in ``audio_io.py``.
"""
from __future__ import annotations

import numpy as np

from . import filters as _filters
from . import tail as _tail
from .plan import MfccConfig, get_plan

__all__ = ["applyFilter", "get_amplitude", "load_channel", "get_MFCCS_change", "mfcc_array"]

applyFilter = _filters.applyFilter


def _load_audio_device(path, sr):
    """librosa.load(path, sr=sr, mono=False) (script/mfcc.py:284,373) for WAVE files, on the device: float32
    CUDA(HIP) tensor, [n] for a mono file and [channels, n] otherwise -- RIFF header parsed on the host, PCM
    decode and sample-rate conversion by the kernels of row N4 (modulation_mfcc_amd/audio_io.py, which also
    states how the resampler relates to librosa's soxr_hq)."""
    from .audio_io import load_audio
    x = load_audio(path, sr)
    return x[0] if x.shape[0] == 1 else x


def _load_audio(path, sr):
    """Same, as the numpy array the reference's callers get."""
    return _load_audio_device(path, sr).cpu().numpy()


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
    signal = _load_audio_device(audioIn, sigSr) if isinstance(audioIn, str) else audioIn   # a path never leaves the GPU
    if signal.ndim > 1:
        signal = signal[channelN, :]

    cfg = MfccConfig.from_reference_call(sigSr, tStep=tStep, winLen=winLen, n_mfcc=n_mfcc,
                                         n_fft=n_fft, minFreq=minFreq, maxFreq=maxFreq)
    if _tail.device_path_applies(diffMethod, outFilter):
        import torch
        plan = get_plan(cfg)
        if isinstance(signal, torch.Tensor):
            y = signal.to(device=plan.device, dtype=torch.float32)
        else:
            y = torch.from_numpy(np.ascontiguousarray(np.asarray(signal), dtype=np.float32)).to(plan.device)
        if y.dim() != 1:
            raise ValueError("expected a 1-D signal")
        coeffs_dev = plan.mfcc(y)
        anchors = _tail.time_anchors(coeffs_dev.shape[2], tStep, winLen)
        change = _tail.mfcc_change_device(plan, coeffs_dev, tStep=tStep, removeFirst=removeFirst,
                                          filtCutoff=filtCutoff, filtOrd=filtOrd, diffMethod=diffMethod,
                                          outFilter=outFilter,
                                          outFiltType=outFiltType, outFiltCutOff=outFiltCutOff,
                                          outFiltLen=outFiltLen, outFiltPolyOrd=outFiltPolyOrd)[0].cpu().numpy()
        return change, anchors
    coeffs = mfcc_array(signal.cpu().numpy() if hasattr(signal, "is_cuda") else signal, cfg)
    anchors = _tail.time_anchors(coeffs.shape[1], tStep, winLen)
    change = _tail.mfcc_change(coeffs, tStep=tStep, removeFirst=removeFirst, filtCutoff=filtCutoff,
                               filtOrd=filtOrd, diffMethod=diffMethod, outFilter=outFilter,
                               outFiltType=outFiltType, outFiltCutOff=outFiltCutOff,
                               outFiltLen=outFiltLen, outFiltPolyOrd=outFiltPolyOrd)
    return change, anchors
