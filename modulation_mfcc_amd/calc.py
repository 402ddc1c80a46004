"""Drop-in for the part of the reference's ``script/calc.py`` that touches the MFCC path.

script/calc.py holds no MFCC arithmetic; what the north star names as the
drop-in surface is ``applyFilter`` (script/calc.py:23-129), ``get_velocity`` (:593-650, applied by
the UI to the MFCC-change curve, script/main.py:668-713) and the RMS / Hilbert amplitude envelope
(:221-343).  Praat-backed functions (f0, formants, RMSpraat) and the EMA reader are out of scope.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import hilbert, savgol_filter

from .filters import applyFilter

__all__ = ["applyFilter", "get_velocity", "calculate_amplitude_envelope"]


def _frame_rms(x, frame_length, hop_length, center):
    """librosa.feature.rms(center=..., pad_mode='constant') as called at script/calc.py:331."""
    x = np.asarray(x, dtype=np.float32)
    if center:
        half = frame_length // 2
        x = np.pad(x, (half, half), mode="constant")
    n = 1 + (x.shape[0] - frame_length) // hop_length
    gather = hop_length * np.arange(n)[:, None] + np.arange(frame_length)[None, :]
    return np.sqrt(np.mean(np.abs(x[gather]) ** 2, axis=-1))


def calculate_amplitude_envelope(x, sr, /, *, method: str = "RMS", winLen: float = 0.1,
                                 hopLen: float = 0.01, center: bool = True, outFilter=None,
                                 outFiltType: str = "low", outFiltCutOff=[12], outFiltLen: int = 6,
                                 outFiltPolyOrd: int = 3):
    """script/calc.py:221-343.  'RMS' and 'Hilb'; 'RMSpraat' needs Praat (out of scope)."""
    if method == "RMSpraat":
        raise NotImplementedError("method='RMSpraat' calls Praat through parselmouth; not part of this build")
    if method == "Hilb":
        env = np.abs(hilbert(x))
    elif method == "RMS":
        env = _frame_rms(x, int(winLen * sr), int(hopLen * sr), center).flatten()
    else:
        raise UnboundLocalError(f"unknown amplitude method {method!r}")   # reference: `amp` unbound
    # the reference tests method != 'hilb' (lower case), so 'Hilb' ALSO gets hop-spaced time stamps
    # and the 1/hopLen rate for its output filter (script/calc.py:333-337); kept as is.
    times, env_sr = np.arange(len(env)) * hopLen, 1 / hopLen
    if outFilter is not None:
        env = applyFilter(env, env_sr, filt=outFilter, filtType=outFiltType, cutOff=outFiltCutOff,
                          filtLen=outFiltLen, polyOrd=outFiltPolyOrd)
    return env, times


def _findiff_first_axis(x, h, order, acc):
    """Central finite differences of accuracy ``acc`` with one-sided stencils at the edges --
    the behaviour of findiff.FinDiff(0, h, order, acc=acc) used at script/calc.py:636."""
    x = np.asarray(x, dtype=float)
    n = x.shape[0]
    half = (order + 1) // 2 - 1 + acc // 2        # central stencil half-width
    width_edge = order + acc                      # one-sided stencil points

    def weights(offsets):
        offsets = np.asarray(offsets, dtype=float)
        m = len(offsets)
        A = np.vander(offsets, m, increasing=True).T
        rhs = np.zeros(m)
        rhs[order] = float(np.prod(np.arange(1, order + 1)))
        return np.linalg.solve(A, rhs)

    if n < max(2 * half + 1, width_edge):
        raise ValueError("signal too short for the requested finite-difference stencil")
    out = np.empty_like(x)
    c_off = np.arange(-half, half + 1)
    c_w = weights(c_off)
    for i in range(n):
        if i < half:
            off = np.arange(0, width_edge)
        elif i >= n - half:
            off = np.arange(-(width_edge - 1), 1)
        else:
            off = None
        if off is None:
            out[i] = np.tensordot(c_w, x[i + c_off], axes=(0, 0))
        else:
            out[i] = np.tensordot(weights(off), x[i + off], axes=(0, 0))
    return out / h ** order


def get_velocity(x: np.ndarray, sr: float, difference: int = 1, method: str = "gradient",
                 width: int = 3, accOrder: int = 2, polyOrder: int = 2):
    """First or second derivative of ``x`` (sampled at ``sr``) -- script/calc.py:593-650.

    method 'gradient' (np.gradient, repeated ``difference`` times), 'sg' (Savitzky-Golay with
    ``width`` points and polynomial order ``polyOrder``) or 'finDiff' (finite-difference stencils
    of accuracy ``accOrder``).  Unknown methods raise the reference's ValueError.
    """
    if method == "finDiff":
        return _findiff_first_axis(x, 1 / sr, difference, accOrder)
    if method == "sg":
        return savgol_filter(x, width, polyOrder, deriv=difference, axis=0, mode="interp")
    if method == "gradient":
        for _ in range(difference):
            x = np.gradient(x, 1 / sr)
        return x
    raise ValueError("Méthode inconnue. Utilisez 'gradient', 'sg' ou 'finDiff'.")
