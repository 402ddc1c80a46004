"""dev: the reference's UI defaults (10 kHz, win 250, hop 50, 128 mel, fmax 10 kHz) with the n_fft values a user may type,
1024 clips x 10 s (2 049 024 frames): ms per mfcc() call by stage"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
dev = torch.device("cuda", 0)
audio = 0.1 * torch.randn((1024, 100000), device=dev)
for n_fft in (512, 1024, 2048, 400, 1000, 600):
    plan = MfccPlan(MfccConfig(sr=10000, n_fft=n_fft, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0))
    for _ in range(3): plan.mfcc(audio)
    torch.cuda.synchronize(); plan.timing_enable(True)
    for _ in range(5): plan.mfcc(audio)
    torch.cuda.synchronize(); plan.timing_enable(False)
    print(n_fft, plan.kernel_path, {k: round(a / c, 3) for k, (a, c) in plan.timing_read().items()}, flush=True)
