"""dev: m12 (MFMA mel) kernel vs w16s vs oracle on a few shapes, then both timed at BASELINE configs[2] size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mfcc_oracle as O
from modulation_mfcc_amd import MfccConfig, MfccPlan

kw = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
plan = MfccPlan(MfccConfig(**kw))
print("default path", plan.kernel_path, flush=True)
dev = torch.device("cuda", 0)
bad = 0
for n in (4, 160, 7680, 7681 * 4, 16000, 40964, 160000):
    kinds = ["am", "noise", "quiet_tail", "impulse", "silence"]
    clips = np.stack([O.synth_clip(50 + n + i, n, 16000, k) for i, k in enumerate(kinds)])
    d = torch.from_numpy(clips).to(dev)
    plan.set_variant("m12"); got = plan.mfcc(d).cpu().numpy(); lm, mx = plan.logmel(d)
    plan.set_variant("w16s"); ref = plan.mfcc(d).cpu().numpy(); lmr, mxr = plan.logmel(d)
    plan.set_variant(None)
    dl = (lm - lmr).abs().max().item(); dm = (mx - mxr).abs().max().item()
    for i in range(len(kinds)):
        want = O.mfcc(clips[i], O.OracleConfig(**kw))
        e1 = np.abs(got[i] - want).max() / max(np.abs(want).max(), 1e-30)
        e2 = np.abs(ref[i] - want).max() / max(np.abs(want).max(), 1e-30)
        flag = "" if e1 <= 1e-4 else "  <-- BAD"
        bad += e1 > 1e-4
        print(f"n={n:7d} {kinds[i]:10s} m12 rel {e1:.2e}  w16s rel {e2:.2e}{flag}", flush=True)
    print(f"   logmel |m12-w16s| {dl:.2e}  clipmax diff {dm:.2e}", flush=True)
print("BAD" if bad else "all ok", flush=True)

B, n = 1024, 160000
g = torch.Generator(device=dev).manual_seed(0)
audio = 0.05 * torch.randn((B, n), generator=g, device=dev)
t = torch.arange(n, device=dev, dtype=torch.float64) / 16000
audio += (0.3 * torch.sin(2 * np.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 4 * t))).float()[None]
out = torch.empty((B, 13, 1001), device=dev)
res = {}
for v in ("m12", "w16s", "m12", "w16s"):
    plan.set_variant(v)
    for _ in range(3): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    plan.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(20): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    plan.timing_enable(False)
    tr = plan.timing_read()
    print(v, "wall ms/step %.4f" % (dt * 1e3), {k: round(a / c, 4) for k, (a, c) in tr.items()}, flush=True)
    res[v] = out.clone()
print("full-size m12 vs w16s max rel diff", ((res["m12"] - res["w16s"]).abs().max() / res["w16s"].abs().max()).item())
# clamp-active batch: silence in the second half of every clip
audio[:, n // 2:] = 0
for v in ("m12", "w16s"):
    plan.set_variant(v)
    for _ in range(3): plan.mfcc(audio, out=out)
    torch.cuda.synchronize()
    plan.timing_enable(True)
    for _ in range(10): plan.mfcc(audio, out=out)
    plan.timing_enable(False)
    tr = plan.timing_read()
    print("clamp-active", v, {k: round(a / c, 4) for k, (a, c) in tr.items()}, flush=True)
    res[v] = out.clone()
print("clamp-active m12 vs w16s max rel diff", ((res["m12"] - res["w16s"]).abs().max() / res["w16s"].abs().max()).item())
