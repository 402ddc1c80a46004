"""dev helper (GPU box, MODMFCC_LIB=libmodmfcc_dev.so from `make -C modulation_mfcc_amd/csrc dev`): s_memtime totals of
the sections of chg_clip_kernel, workgroup 0, per wave.
Sections: 0 load rows  1 odd extension  2/3/4 rows filter A (zero-state run) / B (chain) / C (rerun + store), both
directions summed  5 derivative + norm  6 curve extension  7/8/9 curve filter A / B / C  10 store"""
import sys, ctypes, os
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan, _lib, tail
plan = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100., fmax=8000.))
m = torch.randn((1024, 13, 1001), device="cuda")
sos1 = tail.design_lowpass(6, 12, 0.01)
sos2 = tail.iir_sos(100.0, cutOff=[12], filtLen=6, filtType="low")
plan.mfcc_change(m, sos1, sos2)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_uint * 256)()
assert lib.mm_debug_stamps(out) == 0
a = np.array(out[:]).reshape(16, 16)[:8, :11]
np.set_printoptions(linewidth=200)
print("s_memtime ticks per section (cols), per wave (rows), one clip:")
print(a)
print("share of the total:", (a.mean(0) / a.mean(0).sum()).round(3), "total", int(a.mean(0).sum()))
