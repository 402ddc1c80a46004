"""dev: time mm_pcm_decode_f32 + mm_resample_f32 on a batch (GPU box): 1024 rows x 10 s, 44.1 kHz -> 16 kHz and 48 -> 16"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from modulation_mfcc_amd import audio_io
dev = torch.device("cuda", 0)
for sr_in, sr_out, rows in ((44100, 16000, 256), (44100, 10000, 256), (48000, 16000, 256), (22050, 16000, 256), (16000, 10000, 256), (8000, 16000, 256), (16000, 44100, 64)):
    n = 10 * sr_in
    x = torch.randn((rows, n), device=dev)
    L, M = audio_io.resample_ratio(sr_in, sr_out)
    h, half = audio_io.design_taps(L, M)
    tpp = -(-len(h) // L)
    for method in ("auto", "f64"):
        for _ in range(3): y = audio_io.resample_batch(x, sr_in, sr_out, method=method)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): y = audio_io.resample_batch(x, sr_in, sr_out, method=method)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{sr_in} -> {sr_out} [{method}]: L/M {L}/{M}, {len(h)} taps ({tpp} per output), {rows} rows x {n}: {dt*1e3:.3f} ms, "
              f"{rows*y.shape[1]/dt/1e9:.2f} G out-samples/s, {2*rows*y.shape[1]*tpp/dt/1e12:.2f} TFLOP/s algorithmic", flush=True)
