"""The MFCC-change tail of the reference (script/mfcc.py:390-427): what happens to the MFCC matrix
after the librosa call.  SURVEY.md 8(f) row N1.

    drop c0 -> Butterworth low-pass (zero-phase) per coefficient -> time derivative ->
    L2 norm over coefficients / n_coef -> low-pass again (or the caller's output filter)

The filter design is host arithmetic in the reference (scipy.signal.butter) and stays so; the
filtering itself goes through scipy's sosfiltfilt / gradient on the [n_coef, T] matrix the device
returned (float64, as scipy upcasts in the reference).
"""
from __future__ import annotations

import numpy as np
from scipy import signal as _sig

from .filters import applyFilter


def time_anchors(n_frames: int, tStep: float, winLen: float) -> np.ndarray:
    """Frame centre times, script/mfcc.py:390: round(k * tStep + winLen / 2, 4), k = 1..n_frames."""
    k = np.arange(1, n_frames + 1)
    return np.round(k * tStep + winLen / 2, 4)


def design_lowpass(filtOrd: int, filtCutoff: float, tStep: float) -> np.ndarray:
    """script/mfcc.py:398-400: cutoff normalised by the frame-rate Nyquist (1/tStep)/2."""
    return _sig.butter(filtOrd, filtCutoff / ((1 / tStep) / 2), btype="low", output="sos")


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
