"""dev helper (GPU box): randomized check of the fused tail -- mfcc_modspec() as one launch vs the separate launches:
MFCC bit for bit (clips that clamp included), modulation spectrum to float32 round-off -- over random n_fft 512-class
configurations, hops, clip lengths (trajectory lengths 257 .. 1024), batch sizes and alignments.
usage: python tools/fuzz_tail.py [n_configs] [seed]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from modulation_mfcc_amd import MfccConfig, MfccPlan

def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    bad = fused_runs = 0
    for idx in range(n_cfg):
        n_fft = int(rng.choice([512, 512, 512, 256, 128]))
        win = int(rng.integers(max(2, n_fft // 2), n_fft + 1))
        hop = int(rng.integers(8, 253))
        if rng.random() < 0.7: hop += hop & 1
        sr = int(rng.choice([8000, 16000, 22050]))
        n_mels = int(rng.integers(8, 49))
        n_mfcc = int(rng.integers(1, min(n_mels, 16) + 1))
        # (fmax above Nyquist: empty mel filters, handled analytically; `wide`: the opt-in one-launch forms -- trajectories of
        # 1025 .. 2048 frames and mfcc() on plans with empty filters, mm_plan_set_fuse_tail(plan, 2))
        kw = dict(sr=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=n_mels, n_mfcc=n_mfcc, fmin=50.0,
                  fmax=float(rng.choice([sr / 2, sr / 2, sr * 0.7])),
                  top_db=float(rng.choice([80.0, 30.0, -1.0])), preemph=float(rng.choice([0.0, 0.0, 0.97])))
        plan = MfccPlan(MfccConfig(**kw))
        wide = rng.random() < 0.4
        T = int(rng.integers(1025, 2049)) if wide and hop <= 120 else int(rng.integers(257, 1025))
        default_mode = 2 if wide else 1
        plan.set_fuse_tail(default_mode)
        n = (T - 1) * hop + int(rng.integers(0, hop))
        if rng.random() < 0.6: n = n // 4 * 4
        B = int(rng.choice([256, 256, 512, 300, 257, 768]))
        audio = 0.1 * torch.randn((B, n), generator=g, device=dev)
        audio[::3, n // 2:] *= 1e-5
        audio[1::5] = 0.0
        audio[1::5, n // 3] = 0.5
        if rng.random() < 0.3:                      # unaligned rows
            big = torch.zeros((B, n + 3), device=dev)
            big[:, 1:n + 1] = audio
            audio = big[:, 1:n + 1]
        fused = plan.fused_tail(B, n)
        fused_runs += int(fused)
        m1, s1 = plan.mfcc_modspec(audio)
        mc = plan.mfcc(audio)
        plan.set_fuse_tail(False)
        m0, s0 = plan.mfcc_modspec(audio)
        plan.set_fuse_tail(default_mode)
        ok_m = torch.equal(m1, m0) and torch.equal(mc, m0)
        err = (torch.view_as_real(s1) - torch.view_as_real(s0)).abs().amax(dim=(2, 3))
        scale = torch.view_as_real(s0).abs().amax(dim=(2, 3))
        ok_s = bool((err <= 4e-7 * scale + 1e-30).all()) and bool(torch.isfinite(torch.view_as_real(s1)).all())
        if not (ok_m and ok_s):
            bad += 1
            print("MISMATCH", idx, kw, "B", B, "n", n, "T", plan.cfg.num_frames(n), "fused", fused, "path", plan.kernel_path,
                  "mfcc", ok_m, "spec rel", float((err / (scale + 1e-30)).max()), flush=True)
    print(f"{n_cfg} configurations, {fused_runs} ran the fused tail, {bad} mismatches")

main()
