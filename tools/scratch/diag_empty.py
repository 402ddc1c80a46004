import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, get_plan
kw = dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0)
plan = get_plan(MfccConfig(**kw))
gpu = torch.device("cuda", 0)
B, n = 256, 50000
g = torch.Generator(device=gpu).manual_seed(11)
t = torch.arange(n, device=gpu, dtype=torch.float64) / 10000
base = (0.3 * torch.sin(2 * np.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 4 * t))).float()
audio = 0.05 * torch.randn((B, n), generator=g, device=gpu) + base[None, :]
audio[1::8] *= 1e-3
audio[2::8, n // 2:] *= 1e-6
audio[3::8] = 0.0
audio[4::8] = 0.0
audio[4::8, n // 3] = 1.0
m = plan.mfcc(audio)
m2, s2 = plan.mfcc_modspec(audio)
d = (m2 - m).abs().amax(dim=(1, 2)).cpu().numpy()
for kind in range(8):
    print("kind", kind, "max diff", d[kind::8].max(), "n differing clips", int((d[kind::8] > 0).sum()))
i = int(np.argmax(d))
dd = (m2[i] - m[i]).abs()
print("clip", i, "rows differing:", (dd.amax(dim=1) > 0).cpu().numpy(), "frames differing:", int((dd.amax(dim=0) > 0).sum()), "of", dd.shape[1])
k = int(dd.amax(dim=1).argmax())
print("row", k, m[i, k, :5].cpu().numpy(), m2[i, k, :5].cpu().numpy())
import dataclasses
plan_nc = get_plan(MfccConfig(**dict(kw, top_db=-1.0)))
mnc = plan_nc.mfcc(audio)
lm, mx = plan.logmel(audio[i:i+1])
W = plan.cfg.mel_filterbank(); D = plan.cfg.dct_matrix()
empty = np.abs(W).sum(1) == 0
E = D[:, empty].astype(np.float64).sum(1).astype(np.float32)
L0 = np.float32(lm.min().item())
thr = np.float32(mx.item()) - np.float32(80.0)
delta = np.float32(thr - L0)
print("mx", mx.item(), "L0", L0, "delta", delta)
for k in (0, 4, 5):
    ek = np.float32(E[k] * delta)
    base = mnc[i, k, :3].cpu().numpy()
    print(k, "E", E[k], "ek", ek, "base", base, "host add", (base + ek), "tile", m[i, k, :3].cpu().numpy(), "clip", m2[i, k, :3].cpu().numpy())
