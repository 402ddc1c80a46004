import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
cfg = MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
plan = MfccPlan(cfg)
g = torch.Generator(device="cuda").manual_seed(0)
B, n = 256, 160000
audio = 0.1 * torch.randn((B, n), generator=g, device="cuda")
lm1, mx1 = plan.logmel(audio)
lm2, mx2 = plan.logmel(audio)
d = (lm1 != lm2)
print("logmel run-to-run differing elements:", int(d.sum()), "of", d.numel(), "max clip diff", float((mx1 - mx2).abs().max()))
if d.any():
    idx = d.nonzero()
    print("first diffs (b, m, t):", idx[:10].tolist())
    bt = idx[:, [0, 2]].unique(dim=0)
    print("distinct (b,t) with diffs:", bt.shape[0], "tile-local t%64 hist:", torch.bincount(bt[:, 1] % 64, minlength=64).tolist())
plan.force_generic(True)
lg, mg = plan.logmel(audio)
plan.force_generic(False)
print("vs generic max abs diff:", float((lm1 - lg).abs().max()))
