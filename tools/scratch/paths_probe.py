import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modulation_mfcc_amd import MfccConfig, MfccPlan
for kw in (dict(sr=16000, n_fft=2048, win_length=640, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
           dict(sr=16000, n_fft=2048, win_length=2048, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
           dict(sr=16000, n_fft=2048, win_length=640, hop_length=160, n_mels=80, n_mfcc=13, fmin=100.0, fmax=8000.0),
           dict(sr=16000, n_fft=1024, win_length=321, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
           dict(sr=48000, n_fft=2048, win_length=500, hop_length=240, n_mels=64, n_mfcc=20, fmin=100.0, fmax=10000.0)):
    print(MfccPlan(MfccConfig(**kw)).kernel_path, kw)
