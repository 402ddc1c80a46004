"""dev: time the m12 kernel (and w16s) at BASELINE configs[2] size; used with MODMFCC_LIB=<ablated build>"""
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
out = torch.empty((B, 13, 1001), device=dev)
for v in sys.argv[1:] or ["m12"]:
    fuse = not v.endswith("-nofuse")
    plan.set_variant(v.replace("-nofuse", ""))
    plan.set_fuse_dct(fuse)
    for _ in range(3): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    plan.timing_enable(True)
    for _ in range(20): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    plan.timing_enable(False)
    tr = plan.timing_read()
    print(os.environ.get("MODMFCC_LIB", "product"), v, {k: round(a / c, 4) for k, (a, c) in tr.items()}, "sum", round(sum(a / c for a, c in tr.values()), 4), flush=True)
