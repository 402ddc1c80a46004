"""dev: the reference's own call shape -- ONE recording, tStep 1 ms -- through MFCC + change tail on the device (GPU box):
the tail's segmented-rows form vs the time-major kernels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, tail
from modulation_mfcc_amd.mfcc import get_plan
dev = torch.device("cuda", 0)
for secs in (10, 60, 300):
    sr = 16000
    y = torch.randn(secs * sr, device=dev) * 0.1
    cfg = MfccConfig.from_reference_call(sr, tStep=0.001, winLen=0.025, n_mfcc=13, n_fft=512, minFreq=100, maxFreq=8000)
    plan = get_plan(cfg)
    def run():
        m = plan.mfcc(y)
        return tail.mfcc_change_device(plan, m, tStep=0.001, outFiltCutOff=[12])
    res = {}
    for form in (True, False):
        prev = plan.set_fuse_tail(form)
        for _ in range(2): out = run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        plan.set_fuse_tail(prev)
        res[form] = (dt, out.clone())
    d = (res[True][1] - res[False][1]).abs().max().item() / res[False][1].abs().max().item()
    print(f"{secs:4d} s recording, {out.shape[-1]} frames: MFCC + change tail {res[True][0]*1e3:.3f} ms (time-major tail: {res[False][0]*1e3:.3f} ms), forms differ by {d:.1e}", flush=True)
