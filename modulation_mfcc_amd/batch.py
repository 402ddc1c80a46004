"""Batched entry points (new in the build; the reference processes one clip per call from the
GUI thread, script/main.py:750-769).  All tensors are float32 / complex64 on the GPU.
"""
from __future__ import annotations

from .plan import MfccConfig, get_plan


def mfcc_batch(audio, cfg: MfccConfig, out=None):
    """[B, n] device tensor -> MFCC [B, n_mfcc, T]; per clip identical to librosa.feature.mfcc
    as called at script/mfcc.py:387."""
    return get_plan(cfg).mfcc(audio, out=out)


def modspec_batch(mfcc, cfg: MfccConfig, out=None):
    """[B, n_mfcc, T] -> complex64 [B, n_mfcc, n_mod/2+1]: rFFT of every coefficient trajectory,
    zero-padded to n_mod (build-defined)."""
    return get_plan(cfg).modspec(mfcc, out=out)


def mfcc_modspec_batch(audio, cfg: MfccConfig, mfcc_out=None, mod_out=None):
    """The whole hot path: MFCC and its modulation spectrum."""
    return get_plan(cfg).mfcc_modspec(audio, out=mfcc_out, out_mod=mod_out)     # one launch where the plan can


def rfft_batch(rows, n: int, cfg: MfccConfig = None, out=None):
    """Stage-isolated batched rFFT (the kernel the '% HBM roofline (rFFT)' metric is quoted on)."""
    return get_plan(cfg or MfccConfig()).rfft(rows, n, out=out)


def rms_batch(audio, frame_length: int, hop_length: int, center: bool = True):
    """Framewise RMS on the device (row N3): [B, n] float32 CUDA(HIP) tensor -> [B, n_out], equal to
    librosa.feature.rms(y, frame_length=..., hop_length=..., center=..., pad_mode='constant') per
    clip (script/calc.py:331)."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    if not (isinstance(audio, torch.Tensor) and audio.is_cuda and audio.dtype == torch.float32):
        raise TypeError("audio must be a float32 CUDA(HIP) tensor")
    if audio.dim() == 1:
        audio = audio.unsqueeze(0)
    if audio.stride(1) != 1:
        audio = audio.contiguous()
    B, n = audio.shape
    n_out = lib.mm_rms_num_frames(n, int(frame_length), int(hop_length), 1 if center else 0)
    if n_out < 0:
        _lib.check(int(n_out), "mm_rms_num_frames")
    out = torch.empty((B, n_out), dtype=torch.float32, device=audio.device)
    stream = C.c_void_p(torch.cuda.current_stream(audio.device).cuda_stream)
    _lib.check(lib.mm_rms_f32(audio.data_ptr(), B, n, audio.stride(0), int(frame_length), int(hop_length),
                              1 if center else 0, out.data_ptr(), stream), "mm_rms_f32")
    return out
