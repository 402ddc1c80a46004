"""dev helper (GPU box): randomized check of the matrix-pipe resampler (mm_resample_banded_f32) and its fall-back against
scipy.signal.resample_poly with the same float32 taps: random pairs of common and uncommon sample rates, ragged lengths,
row counts, unaligned row pitches / bases (the scalar staging and store paths).  usage: python tools/fuzz_resample.py [n] [seed]"""
import sys, os
sys.path.insert(0, '.')
import numpy as np, scipy.signal, torch
from modulation_mfcc_amd.audio_io import design_taps, resample_ratio, resample_batch
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
rates = [8000, 10000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 7000, 9600, 37800]
bad = 0
seen = {}
for i in range(n_cfg):
    a, b = (int(v) for v in rng.choice(rates, 2, replace=False))
    L, M = resample_ratio(a, b)
    n = int(rng.choice([1, 2, 3, 17, 255, 1000, 4097, 30001, int(rng.integers(5, 60000))]))
    rows = int(rng.integers(1, 6))
    pad = int(rng.integers(0, 4))                     # row pitch n + pad, base offset: unaligned variants
    off = int(rng.integers(0, 4))
    buf = torch.zeros(rows * (n + pad) + off + 8, device="cuda")
    x = rng.standard_normal((rows, n)).astype(np.float32)
    view = buf[off:off + rows * (n + pad)].view(rows, n + pad)[:, :n]
    view.copy_(torch.from_numpy(x))
    h, half = design_taps(L, M)
    want = scipy.signal.resample_poly(x.astype(np.float64), L, M, axis=1, window=h.astype(np.float32).astype(np.float64) / L)
    for method in ("auto", "f64"):
        got = resample_batch(view, a, b, method=method).cpu().numpy()
        ok = got.shape == want.shape and np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-30)
        if not ok:
            bad += 1
            print("MISMATCH", a, b, "L/M", L, M, "n", n, "rows", rows, "pad", pad, "off", off, method,
                  got.shape, want.shape, np.abs(got - want).max() / max(np.abs(want).max(), 1e-30) if got.shape == want.shape else -1, flush=True)
    seen[(L, M)] = seen.get((L, M), 0) + 1
print("done", n_cfg, "cases,", len(seen), "ratios, mismatches:", bad)
sys.exit(1 if bad else 0)
