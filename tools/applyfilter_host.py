"""dev: host-side cost of applyFilter(filt='iir') on a small device batch (cProfile; GPU box)"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import applyFilter
x = torch.randn((1024, 1001), dtype=torch.float64, device="cuda")
for _ in range(5): applyFilter(x, 100.0, filt="iir", cutOff=[12.0], filtLen=6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): applyFilter(x, 100.0, filt="iir", cutOff=[12.0], filtLen=6)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"50 calls: host {1e3*(t1-t0)/50:.3f} ms per call, + {1e3*(t2-t1):.3f} ms drain")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): applyFilter(x, 100.0, filt="iir", cutOff=[12.0], filtLen=6)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
