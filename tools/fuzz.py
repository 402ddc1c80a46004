"""dev helper (GPU box): randomized parity sweep, larger than the test suite's -- random configurations
(biased towards the radix-16 kernels; a quarter with n_fft that is not a power of two: the any-length kernel) x ragged
clip lengths incl. multiples of 4 (staged-sample kernel),
GPU vs the NumPy oracle with the tests' tolerance.  usage: python tools/fuzz.py [n_configs] [seed]"""
import sys, os, warnings
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import mfcc_oracle as O
from modulation_mfcc_amd import MfccConfig, MfccPlan

def close(a, b):
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    return err.max() <= 1e-4 * scale and not (err > 1e-4 * np.abs(b) + 1e-3).any(), err.max() / scale

def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    rng = np.random.default_rng(seed)
    bad = 0; paths = {}
    for idx in range(n_cfg):
        n_fft = int(rng.choice([256, 512, 512, 512, 512, 1024, 2048, 400, 600, 1000, 1536, int(rng.integers(8, 2400))]))
        win = int(rng.integers(max(2, n_fft // 4), n_fft + 1))
        hop = int(rng.integers(1, max(2, min(win, 300))))
        if rng.random() < 0.8: hop += hop & 1
        if rng.random() < 0.5: hop = max(4, hop // 4 * 4)
        sr = int(rng.choice([8000, 10000, 16000, 22050, 44100, 48000]))
        n_mels = int(rng.integers(2, min(129, n_fft // 2)))
        n_mfcc = int(rng.integers(1, min(n_mels, 40) + 1))
        fmin = float(rng.choice([0.0, 20.0, 100.0, 300.0]))
        fmax = float(rng.choice([sr / 2, sr / 2 * 0.9, sr * 0.7, 3000.0]))
        if fmax <= fmin + 50: continue
        kw = dict(sr=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=n_mels, n_mfcc=n_mfcc, fmin=fmin, fmax=fmax,
                  top_db=float(rng.choice([80.0, 40.0, -1.0])), preemph=float(rng.choice([0.0, 0.0, 0.0, 0.97])))
        okw = dict(kw); okw["top_db"] = None if kw["top_db"] < 0 else kw["top_db"]
        plan = MfccPlan(MfccConfig(**kw))
        n = int(rng.integers(max(4, n_fft // 2), 80 * hop + 6 * n_fft))
        if rng.random() < 0.7: n = max(4, n // 4 * 4)
        B = int(rng.integers(1, 6))
        clips = np.stack([O.synth_clip(1000 * seed + 10 * idx + i, n, sr, ("am", "noise", "quiet_tail")[i % 3]) for i in range(B)])
        d = torch.from_numpy(clips).cuda()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wants = [O.mfcc(clips[i], O.OracleConfig(**okw)) for i in range(B)]
        # default kernel (DCT fused where the plan can), the same without the fused DCT, and the matrix-pipe variant
        for mode in ("default", "nofuse", "m12"):
            plan.set_variant("m12" if mode == "m12" else None)
            plan.set_fuse_dct(mode != "nofuse")
            if mode == "m12" and plan.kernel_path != "radix16-m12":
                continue
            if mode == "nofuse" and not plan.fused_dct and idx % 4:
                pass
            got = plan.mfcc(d).cpu().numpy()
            key = plan.kernel_path + ("+dct" if plan.fused_dct else "")
            paths[key] = paths.get(key, 0) + 1
            for i in range(B):
                want = wants[i]
                ok, rel = close(got[i], want) if got[i].shape == want.shape else (False, -1)
                if not ok:
                    bad += 1
                    print("MISMATCH", idx, mode, key, kw, "n", n, "clip", i, "rel", rel, flush=True)
        if idx % 25 == 0: print("..", idx, paths, flush=True)
    print("done", n_cfg, "configs, mismatches:", bad, paths)
    sys.exit(1 if bad else 0)
main()
