import os, sys, time
sys.path.insert(0, '.')
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
dev = torch.device("cuda", 0)
B, n = 256, 160000
audio = 0.1 * torch.randn((B, n), device=dev)
for n_fft, win in ((400, 400), (1000, 400)):
    plan = MfccPlan(MfccConfig(sr=16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0))
    for _ in range(2): plan.stft_power(audio); plan.logmel(audio)
    torch.cuda.synchronize()
    plan.timing_enable(True)
    for _ in range(3): plan.stft_power(audio); plan.logmel(audio)
    torch.cuda.synchronize()
    plan.timing_enable(False)
    tr = plan.timing_read()
    print(n_fft, {k: round(a / c, 3) for k, (a, c) in tr.items()}, flush=True)
