"""dev: Hilbert envelope of 256 x 160000 float32 (ms per call) on the library named by MODMFCC_LIB"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modulation_mfcc_amd import calc
x = torch.randn((256, 160000), device="cuda")
for shape in ((256, 160000), (64, 480000)):
    x = torch.randn(shape, device="cuda")
    for _ in range(5): calc.hilbert_envelope_batch(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): calc.hilbert_envelope_batch(x)
    torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("MODMFCC_LIB", "product")), shape, "ms %.4f" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
