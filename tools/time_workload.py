"""dev (GPU box): steady-state time of one bench workload (c2 / c3 / c4 / refdefault [+ -mod]) on the library named by MODMFCC_LIB
-- one line: per-kernel device times (HIP events) and wall ms per step.  Used by tools/ab_libs.sh for same-box A/B of two builds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from modulation_mfcc_amd import MfccConfig, MfccPlan
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
with_mod = name.endswith("-mod") or name == "c3"
wl = name.replace("-mod", "")
_, B, ch, secs, kw, _ = bench.WORKLOADS[wl]
cfg = MfccConfig(**kw)
n = int(secs * cfg.sr)
dev = torch.device("cuda", 0)
plan = MfccPlan(cfg)
if os.environ.get("MM_FUSE_TAIL"):
    plan.set_fuse_tail(int(os.environ["MM_FUSE_TAIL"]))      # 0 separate launches, 1 default, 2 the wider one-launch forms
audio = bench.synth_batch(torch, dev, B * ch, n, cfg.sr, 0)
T = cfg.num_frames(n)
out = torch.empty((B * ch, cfg.n_mfcc, T), device=dev)
mod = torch.empty((B * ch, cfg.n_mfcc, cfg.mod_fft_len(T) // 2 + 1), dtype=torch.complex64, device=dev) if with_mod else None
fn = (lambda: plan.mfcc_modspec(audio, out=out, out_mod=mod)) if with_mod else (lambda: plan.mfcc(audio, out=out))
for _ in range(50):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fn()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
plan.timing_enable(True)
for _ in range(20):
    fn()
torch.cuda.synchronize()
plan.timing_enable(False)
tr = plan.timing_read()
print(os.path.basename(os.environ.get("MODMFCC_LIB", "product")), name, "wall_ms %.4f" % wall, {k: round(a / c, 4) for k, (a, c) in tr.items()}, flush=True)
