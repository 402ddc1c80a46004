"""MfccConfig / MfccPlan: host-side owner of an ``mm_plan`` (include/modmfcc.h).

``MfccConfig.from_reference_call`` reproduces the host arithmetic of script/mfcc.py:382-387:
``win_length = int(winLen * sigSr)``, ``hop_length = int(tStep * sigSr)`` (Python truncation) and
librosa's defaults for everything the reference does not pass (n_mels = 128, top_db = 80, ...).
torch is used ONLY to own device buffers and to name the current stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, asdict

import numpy as np

from . import _lib


@dataclass(frozen=True)
class MfccConfig:
    sr: float = 10000.0
    n_fft: int = 512
    win_length: int = 250
    hop_length: int = 50
    n_mels: int = 128
    n_mfcc: int = 13
    fmin: float = 100.0
    fmax: float = 10000.0
    preemph: float = 0.0
    top_db: float = 80.0
    amin: float = 1e-10
    center: bool = True
    n_mod_fft: int = 0

    @classmethod
    def from_reference_call(cls, sigSr, *, tStep=0.001, winLen=0.025, n_mfcc=13, n_fft=512,
                            minFreq=100, maxFreq=10000, **extra):
        """Arguments of get_MFCCS_change (script/mfcc.py:291-311) -> the librosa call at :387."""
        return cls(sr=float(sigSr), n_fft=int(n_fft), win_length=int(winLen * sigSr),
                   hop_length=int(tStep * sigSr), n_mels=int(extra.pop("n_mels", 128)),
                   n_mfcc=int(n_mfcc), fmin=float(minFreq), fmax=float(maxFreq), **extra)

    def to_c(self) -> _lib.mm_config:
        return _lib.mm_config(float(self.sr), int(self.n_fft), int(self.win_length),
                              int(self.hop_length), int(self.n_mels), int(self.n_mfcc),
                              float(self.fmin), float(self.fmax), float(self.preemph),
                              float(self.top_db), float(self.amin), 1 if self.center else 0,
                              int(self.n_mod_fft))

    # ---- host-only helpers (no GPU) -----------------------------------------------------
    def validate(self):
        c = self.to_c()
        _lib.check(_lib.load().mm_config_validate(C.byref(c)), "mm_config_validate")
        return self

    def num_frames(self, n_samples: int) -> int:
        c = self.to_c()
        r = _lib.load().mm_num_frames(C.byref(c), int(n_samples))
        if r < 0:
            _lib.check(int(r), "mm_num_frames")
        return int(r)

    @property
    def n_bins(self) -> int:
        return self.n_fft // 2 + 1

    def mod_fft_len(self, n_frames: int) -> int:
        c = self.to_c()
        r = _lib.load().mm_mod_fft_len(C.byref(c), int(n_frames))
        if r < 0:
            _lib.check(int(r), "mm_mod_fft_len")
        return int(r)

    def window(self) -> np.ndarray:
        c = self.to_c()
        out = np.empty(self.n_fft, dtype=np.float32)
        _lib.check(_lib.load().mm_build_window(C.byref(c), out.ctypes.data), "mm_build_window")
        return out

    def mel_filterbank(self) -> np.ndarray:
        c = self.to_c()
        out = np.empty((self.n_mels, self.n_bins), dtype=np.float32)
        _lib.check(_lib.load().mm_build_mel(C.byref(c), out.ctypes.data), "mm_build_mel")
        return out

    def dct_matrix(self) -> np.ndarray:
        c = self.to_c()
        out = np.empty((self.n_mfcc, self.n_mels), dtype=np.float32)
        _lib.check(_lib.load().mm_build_dct(C.byref(c), out.ctypes.data), "mm_build_dct")
        return out

    def mel_sweep(self, n_waves: int = 8):
        """(wlo, whi, d, part): the sweep form of the mel matrix the fused kernel walks."""
        c = self.to_c()
        wlo = np.empty(self.n_bins, dtype=np.float32)
        whi = np.empty(self.n_bins, dtype=np.float32)
        d = np.empty(self.n_bins, dtype=np.int32)
        part = np.empty((n_waves, 4), dtype=np.int32)
        _lib.check(_lib.load().mm_build_mel_sweep(C.byref(c), n_waves, wlo.ctypes.data, whi.ctypes.data,
                                                  d.ctypes.data, part.ctypes.data), "mm_build_mel_sweep")
        return wlo, whi, d, part

    def mel_runs(self, n_waves: int = 8):
        """(hdr [n_runs,4], grp [n_groups,8], part [n_waves,4]): the LDS tables of the fused kernel."""
        c = self.to_c()
        cap_r, cap_g = self.n_mels + 2 * n_waves + 8, self.n_bins // 4 + 2 * self.n_mels + 4 * n_waves + 8
        hdr = np.zeros((cap_r, 4), dtype=np.int32)
        grp = np.zeros((cap_g, 8), dtype=np.float32)
        part = np.zeros((n_waves, 4), dtype=np.int32)
        cnt = np.zeros(2, dtype=np.int32)
        _lib.check(_lib.load().mm_build_mel_runs(C.byref(c), n_waves, hdr.ctypes.data, cap_r,
                                                 grp.ctypes.data, cap_g, part.ctypes.data,
                                                 cnt.ctypes.data), "mm_build_mel_runs")
        return hdr[:cnt[0]], grp[:cnt[1]], part

    def asdict(self):
        return asdict(self)


def butter_sos(order: int, wn: float) -> np.ndarray:
    """Low-pass Butterworth sections from the C ABI (scipy.signal.butter(..., output='sos') layout)."""
    n_sec = (int(order) + 1) // 2
    out = np.empty((max(n_sec, 1), 6), dtype=np.float64)
    r = _lib.load().mm_build_butter_sos(int(order), float(wn), out.ctypes.data)
    if r < 0:
        _lib.check(int(r), "mm_build_butter_sos")
    return out[:r]


KERNEL_PATHS = ("generic", "radix16-w8", "radix16-w16", "radix16-wpf", "radix16-w16s", "radix16-m12", "any-length", "radix16-h32")


def _torch():
    import torch
    return torch


class MfccPlan:
    """Owns one ``mm_plan`` on the current CUDA(HIP) device.  Raises when no GPU is present.

    One plan = one stream at a time: the log-mel workspace (``workspace()``) is owned by the plan and reused by
    every ``mfcc`` call, so concurrent calls on different streams need one plan each (plans are cheap: a few
    hundred KB of constant tables)."""

    def __init__(self, cfg: MfccConfig, device=None):
        torch = _torch()
        lib = _lib.load()
        cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("modulation_mfcc_amd needs an AMD GPU (gfx950); torch.cuda is not "
                               "available and there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self._lib = lib
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            c = cfg.to_c()
            _lib.check(lib.mm_plan_create(C.byref(c), C.byref(self._h)), "mm_plan_create")
        self._ws = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mm_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ----------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _check_audio(self, audio):
        torch = _torch()
        if not (isinstance(audio, torch.Tensor) and audio.is_cuda):
            raise TypeError("audio must be a CUDA(HIP) torch tensor")
        if audio.dtype != torch.float32:
            raise TypeError("audio must be float32")
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        if audio.dim() != 2:
            raise ValueError("audio must be [n] or [batch, n]")
        if audio.stride(1) != 1:
            audio = audio.contiguous()
        if audio.shape[1] < 1:
            raise ValueError("empty audio")
        if audio.device != self.device:
            raise ValueError(f"audio is on {audio.device}, the plan on {self.device}")
        return audio

    def _check_out(self, out, shape, dtype, what):
        """A caller-provided output must be exactly what the kernel writes: its data_ptr() goes to the C ABI
        unchecked otherwise (a short or strided tensor would be an out-of-bounds device write)."""
        torch = _torch()
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.device == self.device):
            raise TypeError(f"{what}: out must be a tensor on {self.device}")
        if out.dtype != dtype or tuple(out.shape) != tuple(shape) or not out.is_contiguous():
            raise ValueError(f"{what}: out must be a contiguous {dtype} tensor of shape {tuple(shape)}, "
                             f"got {out.dtype} {tuple(out.shape)} (contiguous: {out.is_contiguous()})")
        return out

    def workspace(self, batch, n_samples):
        torch = _torch()
        need = int(self._lib.mm_workspace_bytes(self._h, batch, n_samples))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @property
    def kernel_path(self):
        return KERNEL_PATHS[self._lib.mm_plan_kernel_path(self._h)]

    @property
    def fused_dct(self):
        """True when mfcc() applies the DCT inside the log-mel kernel (plus the clamp fix-up launch)."""
        return bool(self._lib.mm_plan_fused_dct(self._h))

    def set_fuse_dct(self, on=True):
        """Allow (default) or forbid the DCT inside the log-mel kernel; returns the previous setting."""
        return bool(self._lib.mm_plan_set_fuse_dct(self._h, 1 if on else 0))

    def set_variant(self, which):
        """Pin a fused-kernel variant (a name from KERNEL_PATHS, its number, or None / 'auto') for the
        calls it can take; returns the previous setting's name."""
        if which in (None, "auto"):
            v = 0
        elif isinstance(which, str):
            v = KERNEL_PATHS.index(which if which.startswith("radix16-") else "radix16-" + which)
        else:
            v = int(which)
        prev = self._lib.mm_plan_set_variant(self._h, v)
        if prev < 0:
            _lib.check(prev, "mm_plan_set_variant")
        return "auto" if prev == 0 else KERNEL_PATHS[prev]

    def fused_tail(self, batch, n_samples):
        """True when mfcc_modspec() of `batch` clips of `n_samples` runs as ONE launch (mm_plan_fused_tail)."""
        return bool(self._lib.mm_plan_fused_tail(self._h, int(batch), int(n_samples)))

    def set_fuse_tail(self, on=True):
        """on=False pins the separate launches for mfcc_modspec() (default: one launch where the plan can) and the
        time-major kernels for mfcc_change() (default: the clip-resident single launch); on=2 widens the one-launch path to
        2048-point trajectories and to mfcc() on plans with empty mel filters (opt-in: include/modmfcc.h); returns the
        previous setting (0 / 1 / 2)."""
        return self._lib.mm_plan_set_fuse_tail(self._h, 2 if on == 2 else (1 if on else 0))

    def force_generic(self, on=True):
        return self._lib.mm_plan_force_generic(self._h, 1 if on else 0)

    # ---- compute ------------------------------------------------------------------------
    def mfcc(self, audio, out=None):
        """[B, n] float32 device tensor -> [B, n_mfcc, T] float32 (librosa layout per clip)."""
        torch = _torch()
        audio = self._check_audio(audio)
        B, n = audio.shape
        T = self.cfg.num_frames(n)
        if out is None:
            out = torch.empty((B, self.cfg.n_mfcc, T), dtype=torch.float32, device=self.device)
        else:
            self._check_out(out, (B, self.cfg.n_mfcc, T), torch.float32, "mfcc")
        ws = self.workspace(B, n)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mm_mfcc_f32(self._h, audio.data_ptr(), B, n, audio.stride(0),
                                             out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                       "mm_mfcc_f32")
        return out

    def mfcc_modspec(self, audio, out=None, out_mod=None):
        """[B, n] float32 device tensor -> (MFCC [B, n_mfcc, T] float32, modulation spectrum complex64
        [B, n_mfcc, n_mod/2+1]): mfcc() followed by modspec() (MFCC bit for bit, spectrum to float32 round-off), in
        one launch where the plan can (fused_tail())."""
        torch = _torch()
        audio = self._check_audio(audio)
        B, n = audio.shape
        T = self.cfg.num_frames(n)
        nm = self.cfg.mod_fft_len(T)
        if out is None:
            out = torch.empty((B, self.cfg.n_mfcc, T), dtype=torch.float32, device=self.device)
        else:
            self._check_out(out, (B, self.cfg.n_mfcc, T), torch.float32, "mfcc")
        if out_mod is None:
            out_mod = torch.empty((B, self.cfg.n_mfcc, nm // 2 + 1), dtype=torch.complex64, device=self.device)
        else:
            self._check_out(out_mod, (B, self.cfg.n_mfcc, nm // 2 + 1), torch.complex64, "modspec")
        if nm > 8192:
            self.mfcc(audio, out=out)
            return out, self.modspec(out, out=out_mod)
        ws = self.workspace(B, n)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mm_mfcc_modspec_f32(self._h, audio.data_ptr(), B, n, audio.stride(0), out.data_ptr(),
                                                     out_mod.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                       "mm_mfcc_modspec_f32")
        return out, out_mod

    def logmel(self, audio):
        """Unclamped 10*log10(max(amin, mel)) [B, n_mels, T] and the per-clip max [B]."""
        torch = _torch()
        audio = self._check_audio(audio)
        B, n = audio.shape
        T = self.cfg.num_frames(n)
        lm = torch.empty((B, self.cfg.n_mels, T), dtype=torch.float32, device=self.device)
        mx = torch.empty((B,), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.mm_logmel_f32(self._h, audio.data_ptr(), B, n, audio.stride(0),
                                           lm.data_ptr(), mx.data_ptr(), self._stream()),
                   "mm_logmel_f32")
        return lm, mx

    def stft_power(self, audio):
        """|STFT|^2 [B, T, n_fft/2+1] float32 (frame-major)."""
        torch = _torch()
        audio = self._check_audio(audio)
        B, n = audio.shape
        T = self.cfg.num_frames(n)
        out = torch.empty((B, T, self.cfg.n_bins), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.mm_stft_power_f32(self._h, audio.data_ptr(), B, n, audio.stride(0),
                                               out.data_ptr(), self._stream()), "mm_stft_power_f32")
        return out

    def rfft(self, rows, n, out=None):
        """Stage-isolated batched rFFT: [R, L<=n] float32 -> complex64 [R, n/2+1]."""
        torch = _torch()
        if not (isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.float32
                and rows.dim() == 2):
            raise TypeError("rows must be a 2-D float32 CUDA(HIP) tensor")
        if rows.stride(1) != 1:
            rows = rows.contiguous()
        R, L = rows.shape
        if rows.device != self.device:
            raise ValueError(f"rows is on {rows.device}, the plan on {self.device}")
        if out is None:
            out = torch.empty((R, n // 2 + 1), dtype=torch.complex64, device=self.device)
        else:
            self._check_out(out, (R, n // 2 + 1), torch.complex64, "rfft")
        _lib.check(self._lib.mm_rfft_f32(self._h, rows.data_ptr(), R, L, rows.stride(0), int(n),
                                         out.data_ptr(), self._stream()), "mm_rfft_f32")
        return out

    def modspec(self, mfcc, out=None):
        """[B, n_mfcc, T] -> complex64 [B, n_mfcc, n_mod/2+1] (row A8)."""
        torch = _torch()
        if not (isinstance(mfcc, torch.Tensor) and mfcc.is_cuda and mfcc.dtype == torch.float32
                and mfcc.dim() == 3 and mfcc.shape[1] == self.cfg.n_mfcc):
            raise TypeError("mfcc must be a float32 CUDA(HIP) tensor [B, n_mfcc, T]")
        mfcc = mfcc.contiguous()
        B, _, T = mfcc.shape
        n = self.cfg.mod_fft_len(T)
        if mfcc.device != self.device:
            raise ValueError(f"mfcc is on {mfcc.device}, the plan on {self.device}")
        if n > 8192:
            # more than 8192 frames per clip (one recording at the reference's 1 ms step): the transform in global memory
            from .calc import rfft_rows_long
            if out is not None:
                self._check_out(out, (B, self.cfg.n_mfcc, n // 2 + 1), torch.complex64, "modspec")
            res = rfft_rows_long(mfcc.reshape(B * self.cfg.n_mfcc, T), n,
                                 None if out is None else out.reshape(B * self.cfg.n_mfcc, n // 2 + 1))
            return res.reshape(B, self.cfg.n_mfcc, n // 2 + 1) if out is None else out
        if out is None:
            out = torch.empty((B, self.cfg.n_mfcc, n // 2 + 1), dtype=torch.complex64,
                              device=self.device)
        else:
            self._check_out(out, (B, self.cfg.n_mfcc, n // 2 + 1), torch.complex64, "modspec")
        _lib.check(self._lib.mm_modspec_f32(self._h, mfcc.data_ptr(), B, T, out.data_ptr(),
                                            self._stream()), "mm_modspec_f32")
        return out

    def mfcc_change(self, mfcc, sos1, sos2=None, remove_first=True, diff_method="grad", out_filter=True):
        """MFCC-change tail on the device (script/mfcc.py:392-427): [B, n_mfcc, T] float32 -> [B, T]
        float64.  sos1 / sos2: SOS arrays [n_sec, 6] (host); sos2 None applies sos1 again (the reference's
        outFilter=None branch); out_filter=False stops after the norm (the caller applies a 'fir' / 'sg'
        output filter to the result).  diff_method 'grad' = np.gradient, anything else = the reference's
        Savitzky-Golay differentiator savgol_filter(x, 3, 2, deriv=1, mode='interp')."""
        torch = _torch()
        if not (isinstance(mfcc, torch.Tensor) and mfcc.is_cuda and mfcc.dtype == torch.float32
                and mfcc.dim() == 3 and mfcc.shape[1] == self.cfg.n_mfcc):
            raise TypeError("mfcc must be a float32 CUDA(HIP) tensor [B, n_mfcc, T]")
        mfcc = mfcc.contiguous()
        B, _, T = mfcc.shape
        s1 = np.ascontiguousarray(np.asarray(sos1, dtype=np.float64).reshape(-1, 6))
        s2 = s1 if sos2 is None else np.ascontiguousarray(np.asarray(sos2, dtype=np.float64).reshape(-1, 6))
        for s in ((s1, s2) if out_filter else (s1,)):
            ntaps = 2 * s.shape[0] + 1 - min(int((s[:, 2] == 0).sum()), int((s[:, 5] == 0).sum()))
            if T <= 3 * ntaps:   # scipy.signal.sosfiltfilt's own check and message
                raise ValueError("The length of the input vector x must be greater than padlen, "
                                 f"which is {3 * ntaps}.")
        sg = 0 if diff_method == "grad" else 1
        if sg and T < 3:    # scipy.signal.savgol_filter's own check and message
            raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")
        out = torch.empty((B, T), dtype=torch.float64, device=self.device)
        # sized by the form this call takes (a few KB for the clip-resident one), not by the bound over all forms
        need = int(self._lib.mm_change_workspace_bytes_for(self._h, B, T, 1 if remove_first else 0, s1.ctypes.data, s1.shape[0],
                                                           s2.ctypes.data, s2.shape[0] if out_filter else 0))
        if need == 0:
            raise ValueError("mm_change_workspace_bytes_for: invalid arguments")
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.mm_mfcc_change_f64(self._h, mfcc.data_ptr(), B, T, 1 if remove_first else 0, sg,
                                                s1.ctypes.data, s1.shape[0], s2.ctypes.data, s2.shape[0] if out_filter else 0,
                                                out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   "mm_mfcc_change_f64")
        return out

    # ---- per-kernel device timing -------------------------------------------------------
    def timing_enable(self, on=True, stages=None):
        """Record hipEvents around the library's launches; `stages` (names from _lib.STAGES) limits
        the recording to those kernels."""
        flag = 1 if on else 0
        if on and stages:
            flag = 0
            for name in stages:
                flag |= 1 << (_lib.STAGES.index(name) + 1)
        _lib.check(self._lib.mm_timing_enable(self._h, flag), "mm_timing_enable")

    def timing_read(self):
        """{stage: (total_ms, launches)} since the last read (synchronises the recorded events)."""
        ms = (C.c_double * _lib.MM_NUM_STAGES)()
        cnt = (C.c_int64 * _lib.MM_NUM_STAGES)()
        _lib.check(self._lib.mm_timing_read(self._h, ms, cnt), "mm_timing_read")
        return {name: (ms[i], int(cnt[i])) for i, name in enumerate(_lib.STAGES) if cnt[i]}


_PLANS = {}


def get_plan(cfg: MfccConfig) -> MfccPlan:
    """Plan cache keyed by (config, device)."""
    torch = _torch()
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    key = (cfg, dev)
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = MfccPlan(cfg)
    return p
