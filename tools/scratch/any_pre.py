"""dev: n_fft 400 / 800 with pre-emphasis on the two-stage register kernel (ms per 1 025 024 frames)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
dev = torch.device("cuda", 0)
audio = 0.1 * torch.randn((1024, 160000), device=dev)
for n_fft in (400, 800):
    for pre in (0.0, 0.97):
        plan = MfccPlan(MfccConfig(sr=16000, n_fft=n_fft, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0, preemph=pre))
        for _ in range(3): plan.mfcc(audio)
        torch.cuda.synchronize(); plan.timing_enable(True)
        for _ in range(5): plan.mfcc(audio)
        torch.cuda.synchronize(); plan.timing_enable(False)
        print(n_fft, pre, {k: round(a / c, 3) for k, (a, c) in plan.timing_read().items()}, flush=True)
