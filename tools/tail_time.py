"""dev: mfcc + modspec of BASELINE configs[2] as separate launches vs the fused-tail launch (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
kw = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
plan = MfccPlan(MfccConfig(**kw))
dev = torch.device("cuda", 0)
B, n = 1024, 160000
g = torch.Generator(device=dev).manual_seed(0)
audio = 0.05 * torch.randn((B, n), generator=g, device=dev)
audio += (0.3 * torch.sin(2 * np.pi * 220 * torch.arange(n, device=dev) / 16000.0))[None]
m = torch.empty((B, 13, 1001), device=dev)
s = torch.empty((B, 13, 513), device=dev, dtype=torch.complex64)
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for rep in range(3):
    plan.set_fuse_tail(False)
    a = t(lambda: plan.mfcc_modspec(audio, out=m, out_mod=s))
    plan.set_fuse_tail(True)
    b = t(lambda: plan.mfcc_modspec(audio, out=m, out_mod=s))
    fused = plan.fused_tail(B, n)
    c = t(lambda: plan.mfcc(audio, out=m))
    print(f"separate {a:.4f} ms   fused tail {b:.4f} ms   (mfcc alone {c:.4f} ms)  fused={fused}", flush=True)
