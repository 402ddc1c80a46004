"""dev: time and check mm_hilbert_envelope on BASELINE-sized clips (GPU box); rocFFT through torch.fft beside it"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, scipy.signal
from modulation_mfcc_amd import calc
dev = torch.device("cuda", 0)
for B, n, dt in ((256, 160000, torch.float32), (256, 131072, torch.float32), (64, 480000, torch.float32), (64, 160000, torch.float64)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((B, n), generator=g, device=dev, dtype=dt)
    for _ in range(2): e = calc.hilbert_envelope_batch(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): e = calc.hilbert_envelope_batch(x)
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 5
    def ref(x):
        X = torch.fft.fft(x, dim=-1); h = torch.zeros(n, dtype=x.dtype, device=dev)
        h[0] = 1; h[1:(n + 1) // 2] = 2
        if n % 2 == 0: h[n // 2] = 1
        return torch.fft.ifft(X * h, dim=-1).abs()
    for _ in range(2): r = ref(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = ref(x)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 5
    want = np.abs(scipy.signal.hilbert(x[:2].double().cpu().numpy(), axis=1))
    err = np.abs(e[:2].cpu().numpy() - want).max() / want.max()
    err_r = np.abs(r[:2].cpu().numpy() - want).max() / want.max()
    print(f"B={B} n={n} {dt}: own {t1*1e3:.2f} ms ({B*n/t1/1e9:.2f} Gsample/s) err {err:.2e} | rocFFT {t2*1e3:.2f} ms err {err_r:.2e}", flush=True)
