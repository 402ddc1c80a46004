"""dev helper: like tools/stamps.py for the clip-mode finalisation (s16_finalize_all): build with -DMM_DEV into
../libmodmfcc_stamp.so, MODMFCC_LIB=that; prints cycles per launch (workgroup 0, all of a wave's row pairs), per wave:
row loads arrived | DFT-16 #1 + twiddles + exchange | DFT-16 #2 (+ radix-2) | split + stores issued  (waves 0-6)"""
import sys, ctypes, os
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan, _lib
plan = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100., fmax=8000.))
x = torch.randn((1024, 160000), device='cuda') * 0.1
n = 3
for _ in range(n): plan.mfcc_modspec(x)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_uint * 128)()
assert lib.mm_debug_fin_stamps(out) == 0
a = np.array(out[:]).reshape(16, 8)[:, :4] / float(n)
np.set_printoptions(linewidth=200)
print((a).round(0).astype(int))
