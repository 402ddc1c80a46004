"""dev: time mm_sosfiltfilt_f64 (applyFilter(filt='iir') on device curves): envelope-like shapes (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from modulation_mfcc_amd import applyFilter
dev = torch.device("cuda", 0)
for rows, n, sr, cut in ((256, 160000, 16000.0, 12.0), (1024, 1001, 100.0, 12.0), (64, 441000, 44100.0, 20.0), (4096, 1001, 100.0, 12.0)):
    x = torch.randn((rows, n), dtype=torch.float64, device=dev).cumsum(dim=1)
    for _ in range(2): y = applyFilter(x, sr, filt="iir", cutOff=[cut], filtLen=6)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): y = applyFilter(x, sr, filt="iir", cutOff=[cut], filtLen=6)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    want = applyFilter(x[0].cpu().numpy(), sr, filt="iir", cutOff=[cut], filtLen=6)
    err = np.abs(y[0].cpu().numpy() - want).max() / np.abs(want).max()
    print(f"sosfiltfilt {rows} x {n}: {dt*1e3:.3f} ms  ({rows*n*16/dt/1e9:.0f} GB/s in+out)  max rel err vs scipy {err:.1e}", flush=True)
