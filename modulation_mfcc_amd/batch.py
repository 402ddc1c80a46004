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
    plan = get_plan(cfg)
    m = plan.mfcc(audio, out=mfcc_out)
    return m, plan.modspec(m, out=mod_out)


def rfft_batch(rows, n: int, cfg: MfccConfig = None, out=None):
    """Stage-isolated batched rFFT (the kernel the '% HBM roofline (rFFT)' metric is quoted on)."""
    return get_plan(cfg or MfccConfig()).rfft(rows, n, out=out)
