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

__all__ = ["read_wav_header", "load_wav", "load_audio", "design_taps", "resample_batch", "resample_ratio"]

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


def resample_batch(x, sr_in: float, sr_out: float):
    """Sample-rate conversion along the last axis of a float32 CUDA(HIP) tensor ([n] or [rows, n]) on the
    device: ceil(n * sr_out / sr_in) samples per row, zero phase (librosa.resample's length and alignment)."""
    import torch
    from . import _lib
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise TypeError("x must be a float32 CUDA(HIP) tensor")
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
    taps, tpp, half = _device_taps(L, M, x.device)
    out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    lib = _lib.load()
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
