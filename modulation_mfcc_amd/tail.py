"""The MFCC-change tail of the reference (script/mfcc.py:390-427): what happens to the MFCC matrix
after the librosa call.  SURVEY.md 8(f) row N1.

    drop c0 -> Butterworth low-pass (zero-phase) per coefficient -> time derivative ->
    L2 norm over coefficients / n_coef -> low-pass again (or the caller's output filter)

The filter DESIGN is host arithmetic in the reference (scipy.signal.butter) and stays so.  The
filtering runs on the device (``mm_mfcc_change_f64``, float64 like scipy) for both differentiators
(np.gradient, and the Savitzky-Golay derivative every other ``diffMethod`` selects) and every output filter:
IIR or none inside ``mm_mfcc_change_f64``, 'fir' / 'sg' as a banded operator on its result (``mm_stencil_f64``;
tap counts beyond that struct go through the host, filters._apply_filter_device).
"""
from __future__ import annotations

import functools

import numpy as np
from scipy import signal as _sig

from .filters import applyFilter, iir_sos


def time_anchors(n_frames: int, tStep: float, winLen: float) -> np.ndarray:
    """Frame centre times, script/mfcc.py:390: round(k * tStep + winLen / 2, 4), k = 1..n_frames."""
    k = np.arange(1, n_frames + 1)
    return np.round(k * tStep + winLen / 2, 4)


@functools.lru_cache(maxsize=64)
def _design_lowpass(filtOrd, filtCutoff, tStep):
    sos = _sig.butter(filtOrd, filtCutoff / ((1 / tStep) / 2), btype="low", output="sos")
    return sos


def design_lowpass(filtOrd: int, filtCutoff: float, tStep: float) -> np.ndarray:
    """script/mfcc.py:398-400: cutoff normalised by the frame-rate Nyquist (1/tStep)/2.  (The design is host
    arithmetic -- scipy.signal.butter, ~0.1 ms -- and is kept per argument set: with it in every call the device
    tail of 1024 clips would wait for the host.)"""
    try:
        return _design_lowpass(filtOrd, filtCutoff, tStep).copy()
    except TypeError:          # unhashable arguments: design as the reference does, every time
        return _sig.butter(filtOrd, filtCutoff / ((1 / tStep) / 2), btype="low", output="sos")


def device_path_applies(diffMethod, outFilter) -> bool:
    return outFilter is None or outFilter in ("iir", "fir", "sg")


def mfcc_change_device(plan, mfcc_dev, *, tStep: float, removeFirst=1, filtCutoff=12, filtOrd=6,
                       diffMethod="grad", outFilter="iir", outFiltType="low", outFiltCutOff=(None,), outFiltLen=6,
                       outFiltPolyOrd=3):
    """Device version of ``mfcc_change`` for [B, n_mfcc, T] MFCCs already on the GPU -> [B, T] f64
    tensor.  Raises exactly what the host version raises for bad filter arguments."""
    sos1 = design_lowpass(filtOrd, filtCutoff, tStep)
    if outFilter in ("fir", "sg"):
        change = plan.mfcc_change(mfcc_dev, sos1, None, remove_first=bool(removeFirst), diff_method=diffMethod,
                                  out_filter=False)
        return applyFilter(change, 1 / tStep, filt=outFilter, filtType=outFiltType, cutOff=outFiltCutOff,
                           filtLen=outFiltLen, polyOrd=outFiltPolyOrd)
    sos2 = None if outFilter is None else iir_sos(1 / tStep, cutOff=outFiltCutOff, filtLen=outFiltLen,
                                                  filtType=outFiltType)
    return plan.mfcc_change(mfcc_dev, sos1, sos2, remove_first=bool(removeFirst), diff_method=diffMethod)


def mfcc_change(coeffs: np.ndarray, *, tStep: float, removeFirst=1, filtCutoff=12, filtOrd=6,
                diffMethod="grad", outFilter="iir", outFiltType="low", outFiltCutOff=(None,),
                outFiltLen=6, outFiltPolyOrd=3) -> np.ndarray:
    """[n_mfcc, T] MFCCs -> [T] amount-of-change curve (script/mfcc.py:392-427)."""
    rows = coeffs[1:, :] if removeFirst else coeffs
    sos = design_lowpass(filtOrd, filtCutoff, tStep)
    smooth = _sig.sosfiltfilt(sos, rows)
    if diffMethod == "grad":
        slope = np.gradient(smooth, axis=1)
    else:
        slope = _sig.savgol_filter(smooth, 3, 2, deriv=1, axis=1, mode="interp")
    change = np.sqrt((slope ** 2).sum(axis=0)) / rows.shape[0]
    if outFilter is None:
        return _sig.sosfiltfilt(sos, change)
    return applyFilter(change, 1 / tStep, filt=outFilter, filtType=outFiltType, cutOff=outFiltCutOff,
                       filtLen=outFiltLen, polyOrd=outFiltPolyOrd)
