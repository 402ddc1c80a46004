"""``applyFilter`` -- shared by the reference's script/mfcc.py:29-135 and script/calc.py:23-129
(the two copies are identical in behaviour).  Filter DESIGN and the fir / sg variants are scipy
calls on the host exactly as in the reference; this is post-processing of short curves
(one value per frame), not the hot path.
"""
from __future__ import annotations

import numpy as np
from scipy import signal as _sig

_KINDS = ("bandpass", "lowpass", "highpass")

_MSG_NO_CUTOFF = "Cannot apply filter without specifying a cut Off freq. (CutOff is None)."
_MSG_NO_FILT = ("Cannot apply filter without specifying a filter method among iir, fir and  sg "
                "(filt is None).")
_MSG_KIND = "filtType must be one among: lowpass, highpass, bandpass. Partial matches allowed."
_MSG_NYQ = ("Cut off frequencies must be smaller than the half of the sampling freq. of the signal "
            "submitted to the filter")
_MSG_ORDER = "If two cut off freqs are provided: cutOff[0]<cutOff[1]"
_MSG_COUNT = ("only one or two cut off frequencies allowed. If two freqs are provided, filtType "
              "must be bandpass")
_MSG_SG = "sg (savitsky Golay) filters can only be lowpass (one cutOff freq allowed)"


def _resolve_kind(prefix):
    for kind in _KINDS:           # reference order: first match in (bandpass, lowpass, highpass)
        try:
            if kind.startswith(prefix):
                return kind
        except TypeError:
            break
    raise Exception(_MSG_KIND)


def _band_edges(cut, sr, kind):
    n = len(cut)
    if not ((n == 1 and kind in ("lowpass", "highpass")) or (n == 2 and kind == "bandpass")):
        raise Exception(_MSG_COUNT)
    return np.asarray(cut, dtype=float) / (sr / 2)


def _validate(filt, cutOff, filtType, sr):
    if filt is None or cutOff is None:
        raise Exception(_MSG_NO_CUTOFF if cutOff is None else _MSG_NO_FILT)
    kind = _resolve_kind(filtType)
    if any((sr / 2) <= np.array(cutOff)):
        raise Exception(_MSG_NYQ)
    if len(cutOff) > 0 and any(np.diff(cutOff) <= 0):
        raise Exception(_MSG_ORDER)
    return kind


def iir_sos(sr, *, cutOff, filtLen=6, filtType="low"):
    """The Butterworth sections applyFilter(filt='iir') would use (same checks, same exceptions)."""
    kind = _validate("iir", cutOff, filtType, sr)
    return _sig.butter(filtLen, _band_edges(cutOff, sr, kind), btype=kind, output="sos")


def applyFilter(x, sr, /, *, filt: str = "iir", cutOff=[None], filtLen: int = 6,
                filtType: str = "low", polyOrd: int = 3, coeffs=None):
    """Zero-phase low / high / band-pass of ``x`` (sampled at ``sr`` Hz).

    filt      'iir' Butterworth of order ``filtLen`` (sosfiltfilt), 'fir' Kaiser(7.4) window of
              ``filtLen`` taps (filtfilt), 'sg' Savitzky-Golay window ``filtLen`` / order ``polyOrd``
    cutOff    one frequency (low/high-pass) or two increasing ones (band-pass), in Hz
    filtType  'low' | 'high' | 'band' (prefix match, as in the reference)
    coeffs    optional ready-made coefficients: SOS array for 'iir', taps for 'fir'.  (The
              reference accepts the argument but then dies on an unbound name; here it works.)

    Raises the reference's bare ``Exception`` messages for the same bad arguments.
    """
    kind = _validate(filt, cutOff, filtType, sr)

    if filt == "iir":
        sos = np.asarray(coeffs) if coeffs is not None else \
            _sig.butter(filtLen, _band_edges(cutOff, sr, kind), btype=kind, output="sos")
        return _sig.sosfiltfilt(sos, x)
    if filt == "fir":
        taps = np.asarray(coeffs) if coeffs is not None else \
            _sig.firwin(filtLen, _band_edges(cutOff, sr, kind), window=("kaiser", 7.4), pass_zero=kind)
        return _sig.filtfilt(taps, 1, x)
    if filt == "sg":
        if len(cutOff) != 1:
            raise Exception(_MSG_SG)
        return _sig.savgol_filter(x, filtLen, polyOrd, deriv=0, mode="interp")
    # unknown method: the reference falls through to `return y` with y never assigned
    raise UnboundLocalError(f"applyFilter: unknown filt {filt!r} (expected 'iir', 'fir' or 'sg')")
