import sys; sys.path.insert(0,'.')
import numpy as np, scipy.signal, torch
from modulation_mfcc_amd.audio_io import design_taps, resample_ratio, resample_batch
rng=np.random.default_rng(0)
xs=rng.standard_normal((3,200000)).astype(np.float32)
for a,b in ((44100,16000),(44100,10000),(48000,16000),(22050,16000),(16000,10000),(8000,16000),(16000,44100)):
    L,M=resample_ratio(a,b); h,half=design_taps(L,M)
    want=scipy.signal.resample_poly(xs.astype(np.float64),L,M,axis=1,window=h.astype(np.float32).astype(np.float64)/L)
    for m in ("auto","f64"):
        got=resample_batch(torch.from_numpy(xs).cuda(),a,b,method=m).cpu().numpy()
        print(a,b,m,"max err / max|y| = %.2e"%(np.abs(got-want).max()/np.abs(want).max()))
