"""dev: the rows around the path on ONE recording (the reference's call shape) instead of a batch of clips (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from modulation_mfcc_amd import calc, audio_io, applyFilter
from modulation_mfcc_amd.batch import rms_batch
dev = torch.device("cuda", 0)
def t(fn, k=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for secs, sr in ((10, 16000), (300, 16000), (300, 44100), (301, 16000)):
    n = secs * sr + (1 if secs == 301 else 0)
    x = torch.randn((1, n), device=dev)
    print(f"--- one recording of {n} samples ({secs} s at {sr} Hz)", flush=True)
    print(f"  hilbert envelope: {t(lambda: calc.hilbert_envelope_batch(x)):.3f} ms", flush=True)
    print(f"  rms envelope:     {t(lambda: rms_batch(x, 400, 160, True)):.3f} ms", flush=True)
    env = calc.hilbert_envelope_batch(x).double()
    print(f"  iir on envelope:  {t(lambda: applyFilter(env, float(sr), filt='iir', cutOff=[12.0], filtLen=6)):.3f} ms", flush=True)
    if sr == 44100:
        print(f"  resample to 16 k: {t(lambda: audio_io.resample_batch(x, 44100, 16000)):.3f} ms", flush=True)
