"""Input side of the path: ``librosa.load(path, sr=sigSr, mono=False)`` as called at
script/mfcc.py:284 and :373 -- decode the file to float32 in [-1, 1), convert it to the requested rate.

Here: a RIFF/WAVE parser on the host (header only), the PCM -> float32 conversion and the sample-rate
conversion on the device (``mm_pcm_decode_f32``, ``mm_resample_f32``).  Both UI call sites of the reference
pass a PATH (script/main.py:750-769, 1049-1066), so this is what the application actually exercises.

Deviation from the reference, stated once: librosa resamples with ``soxr_hq`` (libsoxr's "high quality"
recipe: 20-bit precision, pass band to 0.913 of the lower Nyquist, linear phase, alias-free stop band from the
Nyquist up).  soxr is not installed here and its coefficients are not public API; ``design_taps`` builds a
Kaiser-windowed sinc to the SAME specification (pass-band edge 0.913, stop band from 1.0 x Nyquist, >= 125 dB
attenuation, linear phase, DC gain 1), so the two outputs agree to the ripple / stop-band leakage of two
filters of that class (measured here against the analytic band-limited signal: tests/test_host.py
``test_resampler_quality``) -- NOT bit for bit.  Files already at the requested rate are bit-exact.
Compressed formats (mp3, flac, ogg -- librosa reads them through soundfile / audioread) are out of scope.
"""
from __future__ import annotations

import math
import struct
from fractions import Fraction

import numpy as np

__all__ = ["read_wav_header", "load_wav", "load_audio", "design_taps", "resample_batch", "resample_ratio", "banded_tables"]

_FMT = {("int", 8): 1, ("int", 16): 2, ("int", 24): 3, ("int", 32): 4, ("float", 32): 5, ("float", 64): 6}


def read_wav_header(path):
    """Parse the RIFF chunks of a WAVE file: dict(sr, channels, bits, kind 'int' | 'float', fmt (device decode
    code), data_offset, data_bytes, n_frames).  PCM (1), IEEE float (3) and WAVE_FORMAT_EXTENSIBLE (0xFFFE)
    carrying one of them; anything else raises ValueError."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        while True:
            ck = f.read(8)
            if len(ck) < 8:
                raise ValueError(f"{path}: no data chunk")
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                body = f.read(size + (size & 1))
                if size < 16 or len(body) < 16:
                    raise ValueError(f"{path}: truncated fmt chunk ({size} bytes)")
                tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", body[:16])
                if tag == 0xFFFE and size >= 40 and len(body) >= 26:
                    # WAVE_FORMAT_EXTENSIBLE: wValidBitsPerSample (body[18:20]) may be smaller than the container
                    # `bits`; samples are MSB-justified in the container, and -- like libsndfile -- the decode scales
                    # by the CONTAINER width, so e.g. 20 valid bits in a 24-bit container come out at the same level
                    tag = struct.unpack("<H", body[24:26])[0]        # first two bytes of the sub-format GUID
                fmt = (tag, ch, sr, align, bits)
            elif cid == b"data":
                if fmt is None:
                    raise ValueError(f"{path}: data chunk before fmt chunk")
                offset = f.tell()
                f.seek(0, 2)
                size = min(size, f.tell() - offset)                 # tolerate a stale / streaming size field
                break
            else:
                f.seek(size + (size & 1), 1)
    tag, ch, sr, align, bits = fmt
    kind = {1: "int", 3: "float"}.get(tag)
    if kind is None or (kind, bits) not in _FMT or ch < 1:
        raise ValueError(f"{path}: unsupported WAVE encoding (format tag {tag}, {bits} bits)")
    frame = ch * bits // 8
    return dict(sr=float(sr), channels=ch, bits=bits, kind=kind, fmt=_FMT[(kind, bits)], data_offset=offset,
                data_bytes=size, n_frames=size // frame)


def _stream(torch, dev):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def load_wav(path, device=None):
    """WAVE file -> (float32 CUDA(HIP) tensor [channels, n] in [-1, 1), sample rate).  The data chunk is read
    as bytes, copied to the device and converted there."""
    import torch
    from . import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("modulation_mfcc_amd needs an AMD GPU (gfx950); there is no CPU fallback")
    h = read_wav_header(path)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = h["n_frames"]
    if n < 1:
        return torch.zeros((h["channels"], 0), dtype=torch.float32, device=dev), h["sr"]
    nbytes = n * h["channels"] * h["bits"] // 8
    raw = np.fromfile(path, dtype=np.uint8, count=nbytes, offset=h["data_offset"])
    d_raw = torch.from_numpy(raw).to(dev)
    out = torch.empty((h["channels"], n), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.mm_pcm_decode_f32(d_raw.data_ptr(), h["fmt"], h["channels"], n, out.data_ptr(), n, _stream(torch, dev)),
                   "mm_pcm_decode_f32")
    return out, h["sr"]


# ---- sample-rate conversion ----------------------------------------------------------------------------
PASSBAND = 0.913        # soxr HQ: pass band up to 91.3 % of the (lower) Nyquist frequency
STOP_DB = 125.0         # >= 20-bit precision class


def resample_ratio(sr_in: float, sr_out: float):
    """(L, M) with sr_out / sr_in = L / M in lowest terms (exact for the integer rates audio files carry)."""
    fr = Fraction(sr_out).limit_denominator(1 << 20) / Fraction(sr_in).limit_denominator(1 << 20)
    if fr.numerator > 4096 or fr.denominator > 4096:            # pathological ratios: nearest small fraction
        fr = Fraction(float(sr_out) / float(sr_in)).limit_denominator(4096)
    return fr.numerator, fr.denominator


def design_taps(L: int, M: int, passband: float = PASSBAND, stop_db: float = STOP_DB):
    """Linear-phase low-pass for L / M conversion at the rate L * sr_in: Kaiser-windowed sinc, pass-band edge
    passband * f_N and stop-band edge f_N, f_N = the lower of the two Nyquist frequencies, attenuation stop_db;
    DC gain L.  Returns (h float64 [2 * half_len + 1], half_len)."""
    q = max(L, M)
    f_stop = 0.5 / q                       # cycles per sample at the rate L * sr_in
    f_pass = passband * f_stop
    width = f_stop - f_pass
    beta = 0.1102 * (stop_db - 8.7)        # Kaiser's formula for A > 50 dB
    half_len = int(math.ceil((stop_db - 7.95) / (2.285 * 2 * math.pi * width) / 2))
    n = np.arange(-half_len, half_len + 1, dtype=np.float64)
    fc = 0.5 * (f_pass + f_stop)
    h = 2 * fc * np.sinc(2 * fc * n) * np.kaiser(2 * half_len + 1, beta)
    return h * (L / h.sum()), half_len


import collections

_TAPS = collections.OrderedDict()      # (L, M, device) -> device taps, least recently used first
TAPS_MAX = 8                           # a 44.1 -> 10 kHz design is ~83 k taps; unusual rate pairs must not pile up


def _device_taps(L, M, dev):
    import torch
    key = (L, M, str(dev))
    if key in _TAPS:
        _TAPS.move_to_end(key)
    else:
        while len(_TAPS) >= TAPS_MAX:
            torch.cuda.synchronize(dev)          # nothing in flight may still read the evicted table
            _TAPS.popitem(last=False)
        h, half = design_taps(L, M)
        tpp = -(-(-(-len(h) // L)) // 4) * 4                # taps per output, padded to a multiple of 4 with zeros
        hp = np.zeros((L, tpp), dtype=np.float32)          # polyphase order: hp[p][j] = h[p + j L]
        for p in range(L):
            col = h[p::L]
            hp[p, :len(col)] = col
        # the kernel wants OUTPUT-phase order in records of four, [tpp / 4][L][4]: output index t of a period uses
        # phase (t M + half) mod L, so adjacent outputs (= adjacent threads) read adjacent 16-byte records
        ph = (np.arange(L, dtype=np.int64) * M + half) % L
        hq = np.ascontiguousarray(hp[ph, :].reshape(L, tpp // 4, 4).transpose(1, 0, 2))
        _TAPS[key] = (torch.from_numpy(hq).to(dev), tpp, half)
    return _TAPS[key]


BANDED_KOFF = (0, 16, 8, 24)      # first sample (within a group of 32) of lane quarter kq: see banded_tables


def banded_tables(L: int, M: int, h: np.ndarray, half: int):
    """The polyphase FIR as the banded GEMM mm_resample_banded_f32 runs on the matrix pipe (csrc/mm_resample.hip.inc).

    Output m = q F + 16 b + r (F a multiple of L: outputs F apart share their taps; b < NB = ceil(F / 16); r < 16,
    16 b + r < F) is
    sum_k A[b][r][k] x[q S + lo_b + k] with S = F M / L.  With t = u M + half, ih0(u) = t div L, ph0(u) = t mod L for
    u = 16 b + r:  lo_b = ih0(16 b) - tpp + 1  and  A[b][r][k] = h[ph0(u) + (ih0(u) - lo_b - k) L]  (zero where the tap
    index falls outside the filter).  Returns dict(atab float32 [NB, ksteps / 4, 64, 4] in MFMA lane order -- lane l of
    k-step ks = 8 g + s holds row r = l % 16 at k = 32 g + BANDED_KOFF[l // 16] + s; entry [b][ks // 4][l][ks % 4] --,
    lo_off int32 [NB], F, S, NB, ksteps (a multiple of 8), lo_min, win)."""
    L, M, half = int(L), int(M), int(half)
    # period F = c L outputs: the smallest multiple of L that is >= 16 and wastes at most 13 % of its last 16-row block
    # (c = 16 / gcd(L, 16) wastes nothing; a short period keeps S, hence the LDS tile of 16 periods, small)
    c = next(c for c in range(max(1, -(-16 // L)), 17) if (-(-(c * L) // 16) * 16) <= 1.13 * c * L)
    F = c * L
    S = F * M // L
    NB = -(-F // 16)
    tpp = -(-len(h) // L)
    u = np.arange(16 * NB, dtype=np.int64)                     # rows past F wrap into the next period: computed, not stored
    t = u * M + half
    ih0, ph0 = t // L, t % L
    lo = ih0[0::16] - tpp + 1                                  # [NB]
    hi = ih0[15::16]
    K = int((hi - lo + 1).max())
    ksteps = -(-K // 32) * 8                                   # k-steps of 4 samples, in groups of 8 (the kernel's pipeline unit)
    k = np.arange(4 * ksteps, dtype=np.int64)
    hp = np.concatenate([np.asarray(h, dtype=np.float64), np.zeros(L)])
    A = np.zeros((NB, 16, 4 * ksteps), dtype=np.float32)
    for b in range(NB):
        uu = 16 * b + np.arange(16)
        j = ih0[uu][:, None] - lo[b] - k[None, :]              # [16, K]
        idx = ph0[uu][:, None] + j * L
        ok = (j >= 0) & (idx < len(h))
        A[b] = np.where(ok, hp[np.clip(idx, 0, len(hp) - 1)], 0.0).astype(np.float32)
    # MFMA A-operand order.  The kernel's K order inside a group of 32 samples: lane quarter kq (lane l = 16 kq + r) owns
    # the 8 consecutive samples KOFF[kq] + 0 .. 7, KOFF = 0 / 16 / 8 / 24 (bank-conflict-free LDS reads for odd S), i.e.
    # k-step ks = 8 g + s multiplies x-window sample k = 32 g + KOFF[kq] + s.  Stored [b][ks / 4][lane][ks % 4]: a lane
    # fetches its taps of four consecutive k-steps with one 16-byte load.
    G = ksteps // 8
    Ag = A.reshape(NB, 16, G, 4, 8)                            # [b][r][g][chunk of 8 = k // 8 % 4][s]
    chunk_of_kq = np.array(BANDED_KOFF) // 8                   # kq -> chunk
    steps = Ag[:, :, :, chunk_of_kq, :]                        # [b][r][g][kq][s]
    steps = steps.transpose(0, 2, 4, 3, 1).reshape(NB, ksteps, 64)                               # [b][ks = 8 g + s][lane = 16 kq + r]
    atab = np.ascontiguousarray(steps.reshape(NB, ksteps // 4, 4, 64).transpose(0, 1, 3, 2))     # [b][ks / 4][lane][4]
    lo_off = (lo - lo[0]).astype(np.int32)
    return dict(atab=atab, lo_off=lo_off, F=F, S=S, NB=NB, ksteps=ksteps, lo_min=int(lo[0]),
                win=int(lo_off.max()) + 4 * ksteps, A=A, lo=lo)


def banded_resample_numpy(x, tabs, n_out):
    """NumPy statement of what the kernel computes from banded_tables() (host logic, tests): float64 accumulation."""
    x = np.asarray(x, dtype=np.float64)
    F, S, NB = tabs["F"], tabs["S"], tabs["NB"]
    A, lo = tabs["A"].astype(np.float64), tabs["lo"]
    K = A.shape[2]
    periods = -(-n_out // F)
    pad_l = max(0, -int(lo.min()))
    xp = np.concatenate([np.zeros(pad_l), x, np.zeros(periods * S + int(lo.max()) + K + 1)])
    y = np.zeros(periods * F)
    for q in range(periods):
        for b in range(NB):
            w = xp[pad_l + q * S + int(lo[b]): pad_l + q * S + int(lo[b]) + K]
            nr = min(16, F - 16 * b)                   # rows past the period's F outputs are not stored
            y[q * F + 16 * b: q * F + 16 * b + nr] = (A[b] @ w)[:nr]
    return y[:n_out]


_BANDED = collections.OrderedDict()


def _device_banded(L, M, dev):
    import torch
    key = (L, M, str(dev))
    if key in _BANDED:
        _BANDED.move_to_end(key)
    else:
        while len(_BANDED) >= TAPS_MAX:
            torch.cuda.synchronize(dev)
            _BANDED.popitem(last=False)
        h, half = design_taps(L, M)
        tb = banded_tables(L, M, h.astype(np.float32).astype(np.float64), half)     # the float32 taps both kernels use
        _BANDED[key] = dict(tb, d_atab=torch.from_numpy(tb["atab"]).to(dev), d_lo_off=torch.from_numpy(tb["lo_off"]).to(dev))
        del _BANDED[key]["A"], _BANDED[key]["atab"]
    return _BANDED[key]


def resample_batch(x, sr_in: float, sr_out: float, method: str = "auto"):
    """Sample-rate conversion along the last axis of a float32 CUDA(HIP) tensor ([n] or [rows, n]) on the
    device: ceil(n * sr_out / sr_in) samples per row, zero phase (librosa.resample's length and alignment).

    method 'auto' / 'mfma': the banded GEMM on the matrix pipe (mm_resample_banded_f32: float32 accumulation in tap
    order, ~1e-6 of full scale from the float64 sum); 'f64': the vector-pipe kernel with float64 accumulation
    (mm_resample_f32; also the fall-back for ratios whose period does not fit the LDS tile)."""
    import torch
    from . import _lib
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise TypeError("x must be a float32 CUDA(HIP) tensor")
    if method not in ("auto", "mfma", "f64"):
        raise ValueError("method must be 'auto', 'mfma' or 'f64'")
    L, M = resample_ratio(sr_in, sr_out)
    if L == M:
        return x.clone()
    squeeze = x.dim() == 1
    x2 = x.unsqueeze(0) if squeeze else x
    if x2.dim() != 2:
        raise ValueError("x must be [n] or [rows, n]")
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    rows, n = x2.shape
    n_out = -(-n * L // M)
    out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    done = False
    if method != "f64":
        tb = _device_banded(L, M, x.device)
        with torch.cuda.device(x.device):
            rc = lib.mm_resample_banded_f32(x2.data_ptr(), rows, n, x2.stride(0), tb["d_atab"].data_ptr(), tb["d_lo_off"].data_ptr(),
                                            tb["F"], tb["S"], tb["NB"], tb["ksteps"], tb["lo_min"], tb["win"], out.data_ptr(),
                                            n_out, _stream(torch, x.device))
        if rc == _lib.MM_ERR_UNSUPPORTED and method == "auto":
            done = False
        else:
            _lib.check(rc, "mm_resample_banded_f32")
            done = True
    if not done:
        taps, tpp, half = _device_taps(L, M, x.device)
        with torch.cuda.device(x.device):
            for r0 in range(0, rows, 65535):         # the kernel's grid takes the rows on its y axis
                r = min(65535, rows - r0)
                _lib.check(lib.mm_resample_f32(x2[r0:].data_ptr(), r, n, x2.stride(0), taps.data_ptr(), L, M, tpp, half,
                                               out[r0:].data_ptr(), n_out, _stream(torch, x.device)), "mm_resample_f32")
    return out[0] if squeeze else out


def load_audio(path, sr=None, device=None):
    """librosa.load(path, sr=sr, mono=False) for WAVE files: float32 CUDA(HIP) tensor [channels, n'] at ``sr``
    (the file's own rate when ``sr`` is None)."""
    x, file_sr = load_wav(path, device)
    if sr is not None and float(sr) != file_sr and x.shape[1] > 0:
        x = resample_batch(x, file_sr, float(sr))
    return x
