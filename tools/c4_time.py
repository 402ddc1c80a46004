"""dev: BASELINE configs[3] (48 kHz, n_fft 2048, 80 mel, 40 MFCC, 1024 channel-rows x 10 s) stage times, the
matrix-pipe clamp + DCT kernel against the VALU one (mm_plan_set_fuse_dct(0) selects the latter) on the same box"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
kw = dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0)
plan = MfccPlan(MfccConfig(**kw))
audio = 0.05 * torch.randn((1024, 480000), device="cuda")
out = torch.empty((1024, 40, 1001), device="cuda")
res = {}
for mode in ("mfma", "valu", "mfma", "valu"):
    plan.set_fuse_dct(mode == "mfma")
    for _ in range(2): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    plan.timing_enable(True)
    for _ in range(8): plan.mfcc(audio, out=out)
    plan.timing_enable(False)
    tr = plan.timing_read()
    print(mode, {k: round(a / c, 4) for k, (a, c) in tr.items()}, flush=True)
    res[mode] = out.clone()
print("max rel diff", ((res["mfma"] - res["valu"]).abs().max() / res["valu"].abs().max()).item())
