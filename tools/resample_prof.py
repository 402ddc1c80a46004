"""dev: a few mm_resample_banded_f32 calls (256 ten-second clips 44.1 -> 16 kHz) for rocprofv3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import audio_io
sr_in, sr_out = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (44100, 16000)
x = torch.randn((256, 10 * sr_in), device="cuda")
for _ in range(6):
    y = audio_io.resample_batch(x, sr_in, sr_out)
torch.cuda.synchronize()
print(y.shape)
