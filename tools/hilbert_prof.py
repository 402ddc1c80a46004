"""dev: a few mm_hilbert_envelope calls for rocprofv3 --kernel-trace --stats (GPU box): argv = B n [f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import calc
B, n = int(sys.argv[1]), int(sys.argv[2])
dt = torch.float64 if len(sys.argv) > 3 else torch.float32
x = torch.randn((B, n), device="cuda", dtype=dt)
for _ in range(6): e = calc.hilbert_envelope_batch(x)
torch.cuda.synchronize()
