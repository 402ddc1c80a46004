"""dev helper (GPU box, MODMFCC_LIB=libmodmfcc_dev.so): s_memtime totals of the sections of sos_row_stream_kernel, last
workgroup, per wave, both directions summed over the rounds of the LAST launch (reverse direction).
Sections: 0 slab written (waits for the prefetched loads)  1 weighted sums (+ next round's loads issued)  2 scan 1
3 barrier 1  4 wave 0's scan over the round's segments  5 barrier 2  6 shift + scan 2  7 recursion  8 stores"""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import applyFilter, _lib
x = torch.randn((256, 160000), dtype=torch.float64, device="cuda").cumsum(dim=1)
applyFilter(x, 16000.0, filt="iir", cutOff=[12.0], filtLen=6)
torch.cuda.synchronize()
lib = _lib.load()
out = (ctypes.c_uint * 256)()
assert lib.mm_debug_stamps(out) == 0
a = np.array(out[:]).reshape(16, 16)[:16, :9]
np.set_printoptions(linewidth=200)
print("ticks per section (cols), per wave (rows), 10 rounds:")
print(a)
print("mean share:", (a.mean(0) / a.mean(0).sum()).round(3), "total per wave", int(a.mean(0).sum()))
