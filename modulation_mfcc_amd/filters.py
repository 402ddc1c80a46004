"""``applyFilter`` -- shared by the reference's script/mfcc.py:29-135 and script/calc.py:23-129
(the two copies are identical in behaviour).  Filter DESIGN (butter, firwin, savgol_coeffs) is scipy on the
host exactly as in the reference.  numpy input is filtered by the reference's own scipy calls; float64 curves
that live on the GPU are filtered there (mm_sosfiltfilt_f64, mm_stencil_f64) -- SURVEY.md 8(f) row N1.
"""
from __future__ import annotations

import functools

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


@functools.lru_cache(maxsize=64)
def _iir_sos_cached(sr, cut, filtLen, kind):
    sos = _sig.butter(filtLen, _band_edges(list(cut), sr, kind), btype=kind, output="sos")
    return sos


def _iir_design(sr, cutOff, filtLen, kind):
    """The Butterworth design of applyFilter(filt='iir'), kept per argument set -- but ONLY for arguments the cache key
    represents exactly: an integral order (int / np.integer, not bool) and numeric cut-offs.  Anything else (filtLen =
    6.5, a string, ...) goes straight to the reference's own scipy.signal.butter call, so scipy's validation and message
    apply ('Filter order must be a nonnegative integer') instead of a silently truncated order."""
    if isinstance(filtLen, (int, np.integer)) and not isinstance(filtLen, (bool, np.bool_)):
        try:
            return _iir_sos_cached(float(sr), tuple(float(c) for c in cutOff), int(filtLen), kind)
        except (TypeError, ValueError):
            pass
    return _sig.butter(filtLen, _band_edges(cutOff, sr, kind), btype=kind, output="sos")


def iir_sos(sr, *, cutOff, filtLen=6, filtType="low"):
    """The Butterworth sections applyFilter(filt='iir') would use (same checks, same exceptions); the design is
    kept per argument set (host arithmetic, scipy.signal.butter)."""
    kind = _validate("iir", cutOff, filtType, sr)
    return np.array(_iir_design(sr, cutOff, filtLen, kind))


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def sosfiltfilt_batch(x, sos):
    """scipy.signal.sosfiltfilt(sos, x) along the last axis of a float64 or float32 CUDA(HIP) tensor [rows, n] (or [n])
    on the device (mm_sosfiltfilt_f64 / _f32_f64): the recursion of applyFilter(filt='iir') for a whole batch of
    curves; float64 result, as scipy returns for either input type."""
    import ctypes as C
    import torch
    from . import _lib
    if not (_is_device_tensor(x) and x.dtype in (torch.float64, torch.float32)):
        raise TypeError("x must be a float64 or float32 CUDA(HIP) tensor")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(0) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be [n] or [rows, n]")
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, n = x2.shape
    s = np.ascontiguousarray(np.asarray(sos, dtype=np.float64).reshape(-1, 6))
    ntaps = 2 * s.shape[0] + 1 - min(int((s[:, 2] == 0).sum()), int((s[:, 5] == 0).sum()))
    if n <= 3 * ntaps:   # scipy.signal.sosfiltfilt's own check and message
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {3 * ntaps}.")
    lib = _lib.load()
    out = torch.empty((rows, n), dtype=torch.float64, device=x.device)
    ws = torch.empty(int(lib.mm_sosfiltfilt_workspace_bytes(rows, n)), dtype=torch.uint8, device=x.device)
    # float32 rows (librosa's RMS envelope ...): scipy forms their odd extension in float32 before it upcasts -- so does
    # mm_sosfiltfilt_f32_f64; the result is float64 either way
    fn, name = (lib.mm_sosfiltfilt_f64, "mm_sosfiltfilt_f64") if x2.dtype == torch.float64 else \
        (lib.mm_sosfiltfilt_f32_f64, "mm_sosfiltfilt_f32_f64")
    with torch.cuda.device(x.device):
        _lib.check(fn(x2.data_ptr(), rows, n, x2.stride(0), s.ctypes.data, s.shape[0], out.data_ptr(), ws.data_ptr(), ws.numel(),
                      C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), name)
    return out[0] if squeeze else out


def fir_filtfilt_stencil(taps):
    """scipy.signal.filtfilt(taps, 1, x) -- the 'fir' branch of applyFilter (script/mfcc.py:113-126) -- as the
    banded operator mm_stencil_f64 applies.  filtfilt pads x by an odd extension of 3 * len(taps) samples, runs
    the filter forwards and backwards from lfilter_zi states and crops the padding; for an FIR filter the two
    start-up transients (len(taps) - 1 samples each) lie inside the cropped padding, so every kept output is
    the correlation of the extended signal with h = taps (*) reversed taps.  Inside that is the symmetric stencil
    h; the first / last len(taps) - 1 outputs fold the taps that reach the extension x_ext[-m] = 2 x[0] - x[m]
    back onto the signal.  Raises NotImplementedError when it does not fit the C struct (more than 8 taps; or a single tap,
    which scipy rejects)."""
    from . import _lib
    b = np.asarray(taps, dtype=np.float64).ravel()
    L = len(b)
    if L < 2 or 2 * L - 1 > _lib.MM_ST_MAXW or L - 1 > _lib.MM_ST_MAXE:     # one tap: scipy's lfilter_zi raises
        raise NotImplementedError("FIR filter too long (or too short) for the device stencil")
    h = np.convolve(b, b[::-1])                      # h[d + L - 1], d = -(L-1) .. L-1 (symmetric)
    half = L - 1
    ew = 2 * half
    el = np.zeros((half, ew))
    for i in range(half):
        for d in range(-half, half + 1):
            j = i + d
            if j >= 0:
                el[i, j] += h[d + half]
            else:                                    # odd extension about x[0]
                el[i, 0] += 2.0 * h[d + half]
                el[i, -j] -= h[d + half]
    er = el[::-1, ::-1]                              # the same fold about x[n - 1]
    return dict(off=list(range(-half, half + 1)), c=[float(v) for v in h], den_c=1.0, n_edge=half, edge_w=ew,
                el=el.tolist(), er=np.ascontiguousarray(er).tolist(), den_e=1.0)


def _stencil_rows(x, st):
    import torch
    from .calc import apply_stencil
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(0) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be [n] or [rows, n]")
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    y = apply_stencil(x2, st, 1)
    return y[0] if squeeze else y


def _apply_filter_device(x, sr, kind, *, filt, cutOff, filtLen, polyOrd, coeffs):
    """applyFilter for float64 CUDA(HIP) curves ([n] or [rows, n], along the last axis): 'iir' through
    mm_sosfiltfilt_f64, 'sg' and 'fir' through the banded-operator kernel (mm_stencil_f64).  Windows / tap
    counts beyond the C struct (Savitzky-Golay windows over 16 samples, FIR filters over 8 taps) make the round
    trip through the host (the reference's own scipy calls)."""
    import torch
    was_f32 = x.dtype == torch.float32
    x_in = x                                # (the host round trip below hands scipy the caller's own type)
    if filt == "iir" and was_f32:
        pass                                # float32 curves keep their type up to the kernel (odd extension in float32)
    elif x.dtype != torch.float64:
        x = x.double()                      # scipy filters in float64 whatever the input type
    if filt == "iir":
        if coeffs is not None:
            sos = np.asarray(coeffs)
        else:                               # the design (host, ~0.15 ms) is kept per argument set: the device filter of
            _band_edges(cutOff, sr, kind)   # 1024 envelope rows takes less than designing it (count check first)
            sos = _iir_design(sr, cutOff, filtLen, kind)
        return sosfiltfilt_batch(x, sos)
    if filt == "sg":
        if len(cutOff) != 1:
            raise Exception(_MSG_SG)
        from .calc import velocity_batch
        try:
            y = velocity_batch(x, 1.0, 0, "sg", filtLen, 2, polyOrd)
            # scipy.signal.savgol_filter keeps a float32 curve float32 (it correlates in double and rounds once): so here
            return y.float() if was_f32 else y
        except NotImplementedError:
            pass
    if filt == "fir":
        taps = np.asarray(coeffs) if coeffs is not None else \
            _sig.firwin(filtLen, _band_edges(cutOff, sr, kind), window=("kaiser", 7.4), pass_zero=kind)
        try:
            st = fir_filtfilt_stencil(taps)
        except NotImplementedError:
            st = None
        if st is not None:
            pad = 3 * len(taps)
            if x.shape[-1] <= pad:             # scipy.signal.filtfilt's own check and message
                raise ValueError(f"The length of the input vector x must be greater than padlen, which is {pad}.")
            if was_f32:
                # scipy forms the odd extension of a float32 curve in float32 before lfilter upcasts (as sosfiltfilt does):
                # extend explicitly in float32, run the operator over the extended curve (its own edge rows then only touch
                # samples that are cropped again: pad > taps - 1), crop
                xi = x_in if x_in.dim() == 2 else x_in.unsqueeze(0)
                n = xi.shape[1]
                ext = torch.cat((2 * xi[:, :1] - xi[:, 1:pad + 1].flip(1), xi, 2 * xi[:, -1:] - xi[:, n - pad - 1:n - 1].flip(1)), dim=1)
                y = _stencil_rows(ext.double(), st)[:, pad:pad + n]
                return y[0] if x_in.dim() == 1 else y
            return _stencil_rows(x, st)
    if filt in ("fir", "sg"):
        y = applyFilter(x_in.cpu().numpy(), sr, filt=filt, cutOff=cutOff, filtLen=filtLen,
                        filtType=kind[:-4], polyOrd=polyOrd, coeffs=coeffs)
        return torch.from_numpy(np.ascontiguousarray(y)).to(x.device)
    raise UnboundLocalError(f"applyFilter: unknown filt {filt!r} (expected 'iir', 'fir' or 'sg')")


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
    if _is_device_tensor(x):       # a batch of curves that lives on the GPU stays there
        return _apply_filter_device(x, sr, kind, filt=filt, cutOff=cutOff, filtLen=filtLen, polyOrd=polyOrd, coeffs=coeffs)

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
