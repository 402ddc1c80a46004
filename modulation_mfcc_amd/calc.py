"""Drop-in for the part of the reference's ``script/calc.py`` that touches the MFCC path.

script/calc.py holds no MFCC arithmetic; what the north star names as the
drop-in surface is ``applyFilter`` (script/calc.py:23-129), ``get_velocity`` (:593-650, applied by
the UI to the MFCC-change curve, script/main.py:668-713) and the RMS / Hilbert amplitude envelope
(:221-343).  Praat-backed functions (f0, formants, RMSpraat) and the EMA reader are out of scope.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import savgol_filter

from .filters import applyFilter

__all__ = ["applyFilter", "get_velocity", "calculate_amplitude_envelope", "velocity_stencil", "velocity_batch", "apply_stencil",
           "hilbert_envelope_batch", "amplitude_envelope_batch"]


def _is_device_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


import collections

_HILBERT_PLANS = collections.OrderedDict()     # (n, dtype, device) -> _HilbertPlan, least recently used first
HILBERT_MAX_PLANS = 8           # every clip length has its own tables (a Bluestein length: ~13 n complex values)
HILBERT_WS_BYTES = 4 << 30      # workspace bound of one mm_hilbert_envelope call: batches are cut to fit


class _HilbertPlan:
    """Owner of one mm_hilbert handle (constant chirp / twiddle tables of one clip length on one device)."""

    def __init__(self, n, dtype_code, device):
        import ctypes as C
        import torch
        from . import _lib
        self._lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self._lib.mm_hilbert_create(n, dtype_code, C.byref(h)), "mm_hilbert_create")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self._lib.mm_hilbert_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _hilbert_plan(n, code, device):
    """The cached mm_hilbert handle of (length, dtype code, device) -- LRU of HILBERT_MAX_PLANS lengths."""
    key = (int(n), int(code), str(device))
    if key in _HILBERT_PLANS:
        _HILBERT_PLANS.move_to_end(key)
    else:
        import torch
        while len(_HILBERT_PLANS) >= HILBERT_MAX_PLANS:      # files of many different lengths: bounded table memory
            _, old = _HILBERT_PLANS.popitem(last=False)
            torch.cuda.synchronize(device)                    # nothing in flight may still read the tables
            del old
        _HILBERT_PLANS[key] = _HilbertPlan(int(n), int(code), device)
    return _HILBERT_PLANS[key]


def rfft_rows_long(x, n_fft, out=None):
    """np.fft.rfft(x, n_fft, axis=-1) of float32 device rows [R, T] (T <= n_fft, n_fft even and 2 / 3 / 5 / 7-smooth, in
    practice a power of two) -> complex64 [R, n_fft / 2 + 1]: the library's Stockham FFT in global memory
    (mm_hilbert_rfft_f32) -- the trajectory rFFT (row A8) of clips with more than 8192 frames, e.g. one recording at the
    reference's default 1 ms step (script/mfcc.py:296).  Rows are taken in chunks that bound the workspace."""
    import torch
    from . import _lib
    if not (_is_device_tensor(x) and x.dtype == torch.float32 and x.dim() == 2):
        raise TypeError("x must be a float32 CUDA(HIP) tensor [rows, T]")
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, T = x.shape
    n_fft = int(n_fft)
    if T < 1 or T > n_fft:
        raise ValueError("need 1 <= T <= n_fft")
    if out is None:
        out = torch.empty((rows, n_fft // 2 + 1), dtype=torch.complex64, device=x.device)
    plan = _hilbert_plan(n_fft, 0, x.device)
    lib = _lib.load()
    per_row = int(lib.mm_hilbert_rfft_workspace_bytes(plan.h, 1))
    chunk = int(max(1, min(rows, 65535, HILBERT_WS_BYTES // max(per_row, 1))))
    ws = torch.empty(int(lib.mm_hilbert_rfft_workspace_bytes(plan.h, chunk)), dtype=torch.uint8, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        for r0 in range(0, rows, chunk):
            r = min(chunk, rows - r0)
            _lib.check(lib.mm_hilbert_rfft_f32(plan.h, x[r0:].data_ptr(), r, x.stride(0), T, out[r0:].data_ptr(),
                                               ws.data_ptr(), ws.numel(), stream), "mm_hilbert_rfft_f32")
    return out


def hilbert_envelope_batch(x):
    """|scipy.signal.hilbert(x)| along the last axis of a CUDA(HIP) tensor ([n] or [B, n], float32 or float64)
    on the device, in scipy's arithmetic: DFT of length N = len(x) in the input's precision, negative
    frequencies zeroed / positive ones doubled, inverse DFT, magnitude (script/calc.py:286).

    N is the clip length, an arbitrary integer: lengths of the form 2^a 3^b 5^c 7^d (160 000 = 2^8 * 5^4 for a 10 s
    clip, 441 000, 480 000 ...) are transformed directly by the library's mixed-radix Stockham FFT, any other length
    through Bluestein's chirp-z identity over a power-of-two FFT (mm_hilbert_envelope, csrc/mm_hilbert.hip.inc);
    the constant tables of a length are built once and kept (LRU of HILBERT_MAX_PLANS lengths)."""
    import ctypes as C
    import torch
    from . import _lib
    if not (_is_device_tensor(x) and x.dtype in (torch.float32, torch.float64)):
        raise TypeError("x must be a float32 / float64 CUDA(HIP) tensor")
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(0) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be [n] or [B, n]")
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, n = x2.shape
    out = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    if n == 0 or rows == 0:
        return out[0] if squeeze else out
    code = 0 if x.dtype == torch.float32 else 1
    plan = _hilbert_plan(n, code, x.device)
    lib = _lib.load()
    # two clips share one complex transform: calls of an even number of clips, sized by what a PAIR needs
    per_pair = int(lib.mm_hilbert_workspace_bytes(plan.h, 2))
    chunk = max(1, min(rows, 65534, 2 * (HILBERT_WS_BYTES // per_pair)))
    ws = torch.empty(int(lib.mm_hilbert_workspace_bytes(plan.h, chunk)), dtype=torch.uint8, device=x.device)
    stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    with torch.cuda.device(x.device):
        for r0 in range(0, rows, chunk):
            r = min(chunk, rows - r0)
            _lib.check(lib.mm_hilbert_envelope(plan.h, x2[r0:].data_ptr(), r, x2.stride(0), out[r0:].data_ptr(), n,
                                               ws.data_ptr(), ws.numel(), stream), "mm_hilbert_envelope")
    return out[0] if squeeze else out


def amplitude_envelope_batch(x, sr, *, method: str = "RMS", winLen: float = 0.1, hopLen: float = 0.01,
                             center: bool = True):
    """The envelope of calculate_amplitude_envelope (script/calc.py:284-343, before its output filter) for a
    batch of clips on the device: [B, n] (or [n]) CUDA(HIP) tensor -> ([B, n_out], sample rate of the envelope).
    'RMS' = librosa.feature.rms(frame_length=int(winLen*sr), hop_length=int(hopLen*sr), center, pad_mode=
    'constant') through mm_rms_f32, 'Hilb' = |hilbert(x)| (hilbert_envelope_batch)."""
    from .batch import rms_batch
    if method == "Hilb":
        return hilbert_envelope_batch(x), 1 / hopLen      # the reference's rate quirk, see below
    if method == "RMS":
        import torch
        xf = x if x.dtype == torch.float32 else x.float()
        squeeze = xf.dim() == 1
        env = rms_batch(xf, int(winLen * sr), int(hopLen * sr), center)
        return (env[0] if squeeze else env), 1 / hopLen
    if method == "RMSpraat":
        raise NotImplementedError("method='RMSpraat' calls Praat through parselmouth; not part of this build")
    raise UnboundLocalError(f"unknown amplitude method {method!r}")


def calculate_amplitude_envelope(x, sr, /, *, method: str = "RMS", winLen: float = 0.1,
                                 hopLen: float = 0.01, center: bool = True, outFilter=None,
                                 outFiltType: str = "low", outFiltCutOff=[12], outFiltLen: int = 6,
                                 outFiltPolyOrd: int = 3):
    """script/calc.py:221-343.  'RMS' and 'Hilb'; 'RMSpraat' needs Praat (out of scope).

    The envelope is computed on the GPU (framewise RMS: mm_rms_f32; Hilbert: device FFT).  A numpy signal
    returns numpy arrays like the reference (its output filter, a short per-frame curve, then runs through the
    reference's own scipy calls on the host); a CUDA(HIP) tensor ([n] or a batch [B, n]) returns device tensors and
    is filtered on the device too (applyFilter's device branch)."""
    if method == "RMSpraat":
        raise NotImplementedError("method='RMSpraat' calls Praat through parselmouth; not part of this build")
    if method not in ("Hilb", "RMS"):
        raise UnboundLocalError(f"unknown amplitude method {method!r}")   # reference: `amp` unbound
    import torch
    on_device = _is_device_tensor(x)
    if on_device:
        xd = x
    else:
        from .plan import _torch  # noqa: F401
        if not torch.cuda.is_available():
            raise RuntimeError("modulation_mfcc_amd needs an AMD GPU (gfx950); there is no CPU fallback")
        xa = np.asarray(x)
        # librosa.feature.rms / scipy.signal.hilbert keep float32 input in single precision, anything else in double
        xd = torch.from_numpy(np.ascontiguousarray(xa, dtype=np.float32 if (method == "RMS" or xa.dtype == np.float32)
                                                   else np.float64)).cuda()
    env, env_sr = amplitude_envelope_batch(xd, sr, method=method, winLen=winLen, hopLen=hopLen, center=center)
    # the reference tests method != 'hilb' (lower case), so 'Hilb' ALSO gets hop-spaced time stamps
    # and the 1/hopLen rate for its output filter (script/calc.py:333-337); kept as is.
    times = np.arange(env.shape[-1]) * hopLen
    if not on_device:
        env = env.cpu().numpy()
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


def _fd_weights(offsets, order):
    """Finite-difference weights of the ``order``-th derivative on integer ``offsets`` (unit spacing)."""
    offsets = np.asarray(offsets, dtype=float)
    m = len(offsets)
    A = np.vander(offsets, m, increasing=True).T
    rhs = np.zeros(m)
    rhs[order] = float(np.prod(np.arange(1, order + 1)))
    return np.linalg.solve(A, rhs)


def velocity_stencil(sr: float, difference: int = 1, method: str = "gradient", width: int = 3,
                     accOrder: int = 2, polyOrder: int = 2):
    """The derivative of get_velocity (script/calc.py:593-650) as the banded operator mm_stencil_f64 applies:
    (stencil dict, passes).  'gradient' is ONE first-derivative stencil applied ``difference`` times, exactly
    as the reference loops np.gradient.  Raises NotImplementedError when the stencil does not fit the C
    struct (windows wider than 16 samples, edge zones longer than 8): callers then use the host path."""
    from . import _lib
    W, E = _lib.MM_ST_MAXW, _lib.MM_ST_MAXE
    if method == "gradient":
        h = 1 / sr      # np.gradient(x, h): (x[i+1] - x[i-1]) / (2 h) inside, first-order differences at the ends
        return dict(off=[-1, 1], c=[-1.0, 1.0], den_c=2.0 * h, n_edge=1, edge_w=2,
                    el=[[-1.0, 1.0]], er=[[-1.0, 1.0]], den_e=h), int(difference)
    if method == "sg":
        from scipy.signal import savgol_coeffs
        width, polyOrder, difference = int(width), int(polyOrder), int(difference)
        half = width // 2
        if width > W or half > E:
            raise NotImplementedError("Savitzky-Golay window too wide for the device stencil")
        conv = savgol_coeffs(width, polyOrder, deriv=difference, delta=1.0)    # taps of scipy's convolve1d
        c = conv[::-1]                                                         # as a correlation: x[i - half + k]
        # mode='interp': the first / last half samples come from the polynomial fitted to the first / last
        # window (scipy.signal._savitzky_golay._fit_edge); linear in the window -> fit unit vectors
        t = np.arange(width)
        pc = np.polyfit(t, np.eye(width), polyOrder)                           # [polyOrder + 1, width]
        for _ in range(difference):
            pc = pc[:-1] * np.arange(pc.shape[0] - 1, 0, -1)[:, None] if pc.shape[0] > 1 else np.zeros((1, width))
        el = [[float(np.polyval(pc[:, j], i)) for j in range(width)] for i in range(half)]
        er = [[float(np.polyval(pc[:, j], width - half + i)) for j in range(width)] for i in range(half)]
        # scipy's convolve1d puts an even kernel's centre at width // 2: as a correlation the taps sit on
        # x[i - (width - 1) // 2 .. i + width // 2] (odd width: -half .. half; width 6, the reference's default
        # outFiltLen: -2 .. 3)
        off = list(range(-((width - 1) // 2), width // 2 + 1))
        assert len(off) == len(c) == width
        return dict(off=off, c=[float(v) for v in c], den_c=1.0, n_edge=half, edge_w=width,
                    el=el, er=er, den_e=1.0), 1
    if method == "finDiff":
        order, acc = int(difference), int(accOrder)
        half = (order + 1) // 2 - 1 + acc // 2
        we = order + acc
        ew = half - 1 + we if half > 0 else we
        if 2 * half + 1 > W or ew > W or half > E:
            raise NotImplementedError("finite-difference stencil too wide for the device stencil")
        c_off = list(range(-half, half + 1))
        el = [[0.0] * ew for _ in range(half)]
        er = [[0.0] * ew for _ in range(half)]
        for i in range(half):
            w = _fd_weights(np.arange(0, we), order)             # forward stencil on samples i .. i + we - 1
            for k in range(we):
                el[i][i + k] = float(w[k])
            # output n - half + i: backward stencil on samples i' - (we - 1) .. i', as positions in the last ew
            w = _fd_weights(np.arange(-(we - 1), 1), order)
            pos = ew - half + i
            for k in range(we):
                er[i][pos - (we - 1) + k] = float(w[k])
        h = 1 / sr
        return dict(off=c_off, c=[float(v) for v in _fd_weights(c_off, order)], den_c=h ** order, n_edge=half,
                    edge_w=ew, el=el, er=er, den_e=h ** order), 1
    raise ValueError("Méthode inconnue. Utilisez 'gradient', 'sg' ou 'finDiff'.")


def apply_stencil(x2, st, passes: int = 1):
    """Run the banded operator ``st`` (dict: off, c, den_c, n_edge, edge_w, el, er, den_e -- the fields of the C
    struct mm_stencil) ``passes`` times along the last axis of a float64 CUDA(HIP) tensor [rows, n] with unit
    inner stride (mm_stencil_f64)."""
    import ctypes as C
    import torch
    from . import _lib
    rows, n = x2.shape
    cs = _lib.mm_stencil()
    if len(st["off"]) != len(st["c"]):
        raise ValueError("stencil: one offset per tap")
    cs.n_c, cs.n_edge, cs.edge_w = len(st["c"]), st["n_edge"], st["edge_w"]
    for k, (o, c) in enumerate(zip(st["off"], st["c"])):
        cs.off[k], cs.c[k] = o, c
    for i in range(st["n_edge"]):
        for j in range(st["edge_w"]):
            cs.el[i][j], cs.er[i][j] = st["el"][i][j], st["er"][i][j]
    cs.den_c, cs.den_e = st["den_c"], st["den_e"]
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
    with torch.cuda.device(x2.device):
        src = x2
        for _ in range(passes):
            out = torch.empty((rows, n), dtype=torch.float64, device=x2.device)
            # (a single row's stride is arbitrary -- numpy's x[None, :] gives 0 -- and never used: pass n)
            _lib.check(lib.mm_stencil_f64(C.byref(cs), src.data_ptr(), rows, n, src.stride(0) if rows > 1 else n,
                                          out.data_ptr(), stream), "mm_stencil_f64")
            src = out
    return src


def velocity_batch(x, sr: float, difference: int = 1, method: str = "gradient", width: int = 3,
                   accOrder: int = 2, polyOrder: int = 2):
    """get_velocity along the LAST axis of a float64 CUDA(HIP) tensor [rows, n] (or [n]) on the device
    (mm_stencil_f64; row N2) -- e.g. on the [B, T] output of MfccPlan.mfcc_change.  'gradient' equals
    np.gradient(x, 1/sr) bit for bit, 'sg' / 'finDiff' agree with scipy / findiff to float64 round-off.  A float32
    tensor (an RMS envelope) comes back float32 for 'gradient' / 'sg', as numpy / scipy return it, to float32 round-off."""
    import torch
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype in (torch.float64, torch.float32)):
        raise TypeError("x must be a float64 (or float32) CUDA(HIP) tensor")
    if x.dtype == torch.float32:
        # numpy / scipy keep a float32 curve float32 for 'gradient' and 'sg' (np.gradient in float32 arithmetic,
        # savgol_filter in double, rounded once); here: float64 on the device, rounded once -- equal to float32 round-off
        y = velocity_batch(x.double(), sr, difference, method, width, accOrder, polyOrder)
        return y.float() if method in ("gradient", "sg") else y
    st, passes = velocity_stencil(sr, difference, method, width, accOrder, polyOrder)
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(0) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be [n] or [rows, n]")
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, n = x2.shape
    if method == "sg" and n < len(st["c"]):       # scipy.signal.savgol_filter's own check and message
        raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")
    if n < max(2 * st["n_edge"], st["edge_w"]):
        raise ValueError("signal too short for the requested finite-difference stencil")
    src = apply_stencil(x2, st, passes)
    return src[0] if squeeze else src


def get_velocity(x: np.ndarray, sr: float, difference: int = 1, method: str = "gradient",
                 width: int = 3, accOrder: int = 2, polyOrder: int = 2):
    """First or second derivative of ``x`` (sampled at ``sr``) -- script/calc.py:593-650.

    method 'gradient' (np.gradient, repeated ``difference`` times), 'sg' (Savitzky-Golay with
    ``width`` points and polynomial order ``polyOrder``) or 'finDiff' (finite-difference stencils
    of accuracy ``accOrder``).  Unknown methods raise the reference's ValueError.

    A float64 CUDA(HIP) tensor (a curve, or [rows, n] curves along the last axis) is differentiated on
    the device (``velocity_batch``); numpy input keeps the reference's host arithmetic.
    """
    if type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False):
        if method not in ("gradient", "sg", "finDiff"):
            raise ValueError("Méthode inconnue. Utilisez 'gradient', 'sg' ou 'finDiff'.")
        try:
            return velocity_batch(x, sr, difference, method, width, accOrder, polyOrder)
        except NotImplementedError:
            import torch
            y = get_velocity(x.cpu().numpy().T, sr, difference, method, width, accOrder, polyOrder)
            return torch.from_numpy(np.ascontiguousarray(np.asarray(y).T)).to(x.device)
    if method == "finDiff":
        return _findiff_first_axis(x, 1 / sr, difference, accOrder)
    if method == "sg":
        return savgol_filter(x, width, polyOrder, deriv=difference, axis=0, mode="interp")
    if method == "gradient":
        for _ in range(difference):
            x = np.gradient(x, 1 / sr)
        return x
    raise ValueError("Méthode inconnue. Utilisez 'gradient', 'sg' ou 'finDiff'.")
