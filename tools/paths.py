import sys; sys.path.insert(0,'.')
from modulation_mfcc_amd import MfccConfig, MfccPlan
for kw in [dict(sr=16000,n_fft=512,win_length=400,hop_length=160,n_mels=40,n_mfcc=13,fmin=100.,fmax=8000.),
           dict(), dict(sr=16000,n_fft=512,win_length=400,hop_length=160,n_mels=256,n_mfcc=13,fmin=0.,fmax=8000.),
           dict(sr=16000,n_fft=512,win_length=400,hop_length=160,n_mels=128,n_mfcc=13,fmin=0.,fmax=8000.)]:
    print(kw.get('n_mels',128), MfccPlan(MfccConfig(**kw)).kernel_path)
