"""dev: random shapes / filters through the float64 IIR paths on the device against scipy on the host (GPU box):
  * applyFilter(filt='iir') on [rows, n] curves (mm_sosfiltfilt_f64: segmented rows up to 4 sections, time-major beyond)
  * the MFCC-change tail (mm_mfcc_change_f64: clip-resident and time-major forms)
usage: python tools/fuzz_sos.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import mfcc_oracle as O
from modulation_mfcc_amd import applyFilter, MfccConfig, MfccPlan, tail
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
worst = 0.0
t0 = time.time()
for c in range(cases):
    order = int(rng.integers(1, 9))
    kind = ["low", "high", "band"][int(rng.integers(0, 3))]
    sr = float(rng.choice([100.0, 200.0, 16000.0]))
    lo = float(rng.uniform(0.02, 0.3)) * sr / 2
    cut = [lo] if kind != "band" else [lo, min(lo * float(rng.uniform(1.5, 3.0)), 0.95 * sr / 2)]
    nsec = order if kind == "band" else (order + 1) // 2
    pad = 3 * (2 * nsec + 1)
    n = int(rng.choice([pad + 1, pad + 2, int(rng.integers(pad + 1, 1200)), int(rng.integers(1000, 6000)), int(rng.integers(60000, 200000))],
                       p=[0.1, 0.1, 0.4, 0.3, 0.1]))
    rows = int(rng.integers(1, 8)) if rng.random() > 0.15 or n > 20000 else int(rng.integers(128, 261))      # (>= 128 rows: a workgroup per row)
    x = rng.standard_normal((rows, n)).cumsum(axis=1) * float(rng.choice([1e-3, 1.0, 1e3])) + float(rng.normal())
    try:
        want = np.stack([applyFilter(r, sr, filt="iir", cutOff=cut, filtLen=order, filtType=kind) for r in x])
    except ValueError as e:
        if "padlen" in str(e): continue
        raise
    got = applyFilter(torch.from_numpy(x).to(dev), sr, filt="iir", cutOff=cut, filtLen=order, filtType=kind).cpu().numpy()
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    worst = max(worst, err)
    if not (err <= 1e-8):
        bad += 1
        print(f"FILTER MISMATCH case {c}: order {order} {kind} cut {cut} sr {sr} rows {rows} n {n}: rel err {err:.2e}", flush=True)
print(f"applyFilter iir: {cases} cases, {bad} mismatches, worst rel err {worst:.2e}, {time.time()-t0:.0f} s", flush=True)
bad2 = 0; worst2 = 0.0
plans = {}
for c in range(cases):
    n_mfcc = int(rng.integers(2, 41))
    T = int(rng.choice([int(rng.integers(30, 400)), int(rng.integers(400, 1500)), int(rng.integers(1500, 5000))]))
    B = int(rng.integers(1, 6))
    kw = dict(filtOrd=int(rng.integers(1, 9)), filtCutoff=float(rng.uniform(3, 30)), outFiltCutOff=[float(rng.uniform(3, 30))],
              outFiltLen=int(rng.integers(1, 9)), removeFirst=int(rng.integers(0, 2)), diffMethod=str(rng.choice(["grad", "sg"])))
    if rng.random() < 0.2: kw["outFilter"] = None
    tstep = float(rng.choice([0.005, 0.01]))
    if n_mfcc - kw["removeFirst"] < 1: continue
    if n_mfcc not in plans:
        plans[n_mfcc] = MfccPlan(MfccConfig(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=max(40, n_mfcc), n_mfcc=n_mfcc, fmin=100.0, fmax=8000.0))
    plan = plans[n_mfcc]
    m = rng.standard_normal((B, n_mfcc, T)).cumsum(axis=2).astype(np.float32)
    try:
        want = [O.mfcc_change_tail(m[i], tStep=tstep, **kw) for i in range(B)]
    except ValueError as e:
        if "padlen" in str(e): continue
        raise
    for form in (True, False):
        prev = plan.set_fuse_tail(form)
        got = tail.mfcc_change_device(plan, torch.from_numpy(m).to(dev), tStep=tstep, **kw).cpu().numpy()
        plan.set_fuse_tail(prev)
        for i in range(B):
            err = np.abs(got[i] - want[i]).max() / max(np.abs(want[i]).max(), 1e-300)
            worst2 = max(worst2, err)
            if not (err <= 1e-9):
                bad2 += 1
                print(f"TAIL MISMATCH case {c} form {'clip' if form else 'time-major'}: n_mfcc {n_mfcc} T {T} B {B} {kw}: {err:.2e}", flush=True)
print(f"change tail: {cases} cases x 2 forms, {bad2} mismatches, worst rel err {worst2:.2e}, {time.time()-t0:.0f} s", flush=True)
sys.exit(1 if bad or bad2 else 0)
