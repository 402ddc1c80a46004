"""dev: time the MFCC-change tail (mm_mfcc_change_f64) at BASELINE size (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan, tail
kw = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
plan = MfccPlan(MfccConfig(**kw))
m = torch.randn((1024, 13, 1001), device="cuda")
sos1 = tail.design_lowpass(6, 12, 0.01)
sos2 = tail.iir_sos(100.0, cutOff=[12], filtLen=6, filtType="low")
outs = {}
for form in ("clip", "time-major", "clip", "time-major"):
    plan.set_fuse_tail(form == "clip")
    for _ in range(3): out = plan.mfcc_change(m, sos1, sos2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = plan.mfcc_change(m, sos1, sos2)
    torch.cuda.synchronize(); print(f"change tail [{form}] (filters designed once), 1024 clips x 13 x 1001: {(time.perf_counter()-t0)/20*1e3:.4f} ms")
    outs[form] = out.clone()
d = (outs["clip"] - outs["time-major"]).abs().max().item() / outs["time-major"].abs().max().item()
print(f"  clip vs time-major form: max rel diff {d:.2e}")
plan.set_fuse_tail(True)
t0 = time.perf_counter()
for _ in range(20): out = tail.mfcc_change_device(plan, m, tStep=0.01, outFiltCutOff=[12])
torch.cuda.synchronize(); print(f"  through tail.mfcc_change_device (filter designs cached on the host): {(time.perf_counter()-t0)/20*1e3:.4f} ms")
