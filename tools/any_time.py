"""dev: time the any-length STFT path (mm_mfcc_f32 on stft_any_kernel) at BASELINE configs[1] size for a few n_fft"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import MfccConfig, MfccPlan
dev = torch.device("cuda", 0)
B, n = 1024, 160000
audio = 0.1 * torch.randn((B, n), device=dev)
for n_fft, win in ((512, 400), (200, 200), (240, 240), (320, 320), (400, 400), (480, 400), (600, 400), (640, 400), (800, 400), (960, 400), (1000, 400), (1200, 400), (1600, 400), (1536, 400), (502, 400), (499, 400), (2000, 1200)):
    plan = MfccPlan(MfccConfig(sr=16000, n_fft=n_fft, win_length=win, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0))
    # an any-length plan's "variant" selects its form: 1 = one frame per wave, 2 .. 4 = frames per wave at once, 0 = automatic
    for v in ((0,) if plan.kernel_path != "any-length" else (1, 2, 0)):
        plan._lib.mm_plan_set_variant(plan._h, v)
        for _ in range(2): plan.mfcc(audio)
        torch.cuda.synchronize()
        plan.timing_enable(True)
        for _ in range(5): plan.mfcc(audio)
        torch.cuda.synchronize()
        plan.timing_enable(False)
        tr = plan.timing_read()
        print(n_fft, plan.kernel_path, "form", v, {k: round(a / c, 3) for k, (a, c) in tr.items()}, flush=True)
