"""dev: a few mm_sosfiltfilt_f64 calls on 256 x 160000 for rocprofv3 --kernel-trace --stats (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import applyFilter
x = torch.randn((256, 160000), dtype=torch.float64, device="cuda").cumsum(dim=1)
for _ in range(10): y = applyFilter(x, 16000.0, filt="iir", cutOff=[12.0], filtLen=6)
torch.cuda.synchronize()
