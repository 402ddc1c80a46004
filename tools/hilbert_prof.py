"""dev: a few Hilbert envelopes of 256 x 160000 for rocprofv3 --kernel-trace --stats (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import calc
x = torch.randn((256, 160000), device="cuda")
for _ in range(10): calc.hilbert_envelope_batch(x)
torch.cuda.synchronize()
