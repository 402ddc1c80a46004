"""dev helper: build a side copy with -DMM_DEV (`make -C modulation_mfcc_amd/csrc dev` -> libmodmfcc_dev.so) and point MODMFCC_LIB at it; runs the fused kernel, print per-section cycle totals of workgroup 0.
Sections: 0 window (+wait for samples)  1 DFT-16 #1  2 twiddles  3 exchange + DFT-16 #2  4 split + power rows
5 prefetch issue  6 barrier A->B  7 phase B  8 barrier B->A  9 loop top  10 exchange (then 3 = DFT-16 #2 alone)
w16s: 11 staging loads issued, 0 S reads + window, 5 staging store (after barrier A->B)"""
import sys, ctypes, subprocess, os
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan, _lib
plan = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100., fmax=8000.))
x = torch.randn((1024, 160000), device='cuda') * 0.1
for _ in range(3): plan.mfcc(x)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_uint * 256)()
assert lib.mm_debug_stamps(out) == 0
a = np.array(out[:]).reshape(16, 16)[:, :16]
tiles = 16016 // 256
np.set_printoptions(linewidth=200)
print("cycles per tile, per wave (rows) x section (cols):")
print((a / tiles).round(0).astype(int))
print("mean over waves:", (a.mean(0) / tiles).round(0), "sum", round(a.mean(0).sum() / tiles))
