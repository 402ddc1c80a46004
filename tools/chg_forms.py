"""dev (GPU box, MODMFCC_LIB = the -DMM_DEV side build): the change tail's clip-resident form vs its segmented-rows form
over a grid of (clips, frames): where does each win?  MM_CHG_FORM=c|s pins the form in the side build."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from modulation_mfcc_amd import MfccConfig, MfccPlan, tail
    plan = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0))
    sos = tail.design_lowpass(6, 12, 0.01)
    for B in (1, 4, 16, 64, 256, 1024):
        for T in (501, 1001, 2001, 4001, 8001):
            if B * T > 1024 * 4001: continue
            m = torch.randn((B, 13, T), device="cuda")
            for _ in range(3): plan.mfcc_change(m, sos, sos)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): plan.mfcc_change(m, sos, sos)
            torch.cuda.synchronize(); print(f"{sys.argv[1]} {B} {T} {(time.perf_counter()-t0)/10*1e3:.4f}", flush=True)
    sys.exit(0)
res = {}
for form in ("c", "s"):
    env = dict(os.environ, MM_CHG_FORM=form)
    out = subprocess.run([sys.executable, __file__, form], env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        f, B, T, ms = line.split()
        res[(int(B), int(T), f)] = float(ms)
print("clips frames  clip-form ms  segmented ms")
for (B, T, f) in sorted(k for k in res if k[2] == "c"):
    print(f"{B:5d} {T:6d}  {res[(B,T,'c')]:10.4f}  {res.get((B,T,'s'), float('nan')):10.4f}  {'<- segmented' if res.get((B,T,'s'), 9e9) < res[(B,T,'c')] else ''}")
