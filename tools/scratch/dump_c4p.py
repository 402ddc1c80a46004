import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, "oracle")
import mfcc_oracle as O
from modulation_mfcc_amd import MfccConfig, MfccPlan
kw = dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0)
plan = MfccPlan(MfccConfig(**kw))
y = O.synth_clip(3, 24000, 48000, "am")
d = torch.from_numpy(y).cuda()
pw = plan.stft_power(d)
lm, mx = plan.logmel(d)
np.save(sys.argv[1], pw.cpu().numpy())
np.save(sys.argv[1] + ".lm.npy", lm.cpu().numpy())
