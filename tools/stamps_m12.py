"""dev: per-wave s_memtime totals of the m12 kernel's loop sections (workgroup 0); needs a -DMM_DEV build
(MODMFCC_LIB).  Sections: 0 S reads + barrier B | 1 DMA issue | 2 mel (rank 0) | 3 window + DFT-16 + twiddles
| 4 mel (rank 1) | 5 exchange | 6 mel (rank 2) | 7 DFT-16 #2 + split + power rows | 8 vmcnt wait | 9 barrier A"""
import sys, ctypes, os
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan, _lib
plan = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100., fmax=8000.))
plan.set_variant("m12")
x = torch.randn((1024, 160000), device='cuda') * 0.1
for _ in range(3): plan.mfcc(x)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_uint * 256)()
lib.mm_debug_stamps.restype = ctypes.c_int
assert lib.mm_debug_stamps(out) == 0
a = np.array(out[:]).reshape(16, 16)[:16, :10]
tiles = 21 * 1024 // 256
np.set_printoptions(linewidth=200)
print(os.environ.get("MODMFCC_LIB"), "cycles per tile, per wave (rows) x section (cols):")
print((a / tiles).round(0).astype(int))
print("mean over waves:", (a.mean(0) / tiles).round(0), "sum", round(a.mean(0).sum() / tiles))
