import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Fixture -> (oracle cfg kwargs, regenerated input clip, dict of expected arrays)."""
    import mfcc_oracle as O
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("sr", "n_fft", "win_length", "hop_length", "n_mels", "n_mfcc"):
        kw[k] = int(kw[k])
    seed, n = (int(v) for v in z["recipe"])
    y = O.synth_clip(seed, n, kw["sr"], str(z["kind"]))
    assert int(np.abs(y).sum() * 1e6) == int(z["audio_crc"][0]), "input recipe drifted"
    return kw, y, {k: z[k] for k in z.files}


XCHECK = os.path.join(ROOT, "tests", "golden_xcheck")
XCHECK_NAMES = sorted(f[:-4] for f in os.listdir(XCHECK) if f.endswith(".npz")) if os.path.isdir(XCHECK) else []


def load_xcheck(name):
    """Fixture written by oracle/crosscheck_transformers.py --write: MFCCs of an INDEPENDENT librosa-compatible
    implementation (transformers.audio_utils + scipy.fftpack.dct) -> (cfg kwargs, regenerated input, expected)."""
    import mfcc_oracle as O
    z = np.load(os.path.join(XCHECK, name + ".npz"))
    kw = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("sr", "n_fft", "win_length", "hop_length", "n_mels", "n_mfcc"):
        kw[k] = int(kw[k])
    seed, n = (int(v) for v in z["recipe"])
    return kw, O.synth_clip(seed, n, kw["sr"], str(z["kind"])), z["mfcc"]


GOLDEN_NAMES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")) \
    if os.path.isdir(GOLDEN) else []


def mfcc_close(a, b, what=""):
    """North-star tolerance: 1e-4 relative to the clip's max |MFCC| AND elementwise
    |a-b| <= 1e-4*|b| + 1e-3 (MFCCs are dB-derived and cross zero)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = np.abs(a - b)
    assert err.max() <= 1e-4 * scale, f"{what}: max err {err.max():.3e} vs 1e-4*{scale:.3e}"
    bad = err > 1e-4 * np.abs(b) + 1e-3
    assert not bad.any(), f"{what}: {bad.sum()} elements beyond 1e-4*|b|+1e-3 (worst {err.max():.3e})"


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch.device("cuda", 0)


# Published finite-difference coefficient tables (B. Fornberg, "Generation of finite difference formulas on arbitrarily
# spaced grids", Math. Comp. 51 (1988), tables 1 and 3; unit spacing), keyed (derivative, accuracy): what
# findiff.FinDiff(0, h, derivative, acc=accuracy) -- script/calc.py:636 -- applies: `central` inside, `forward` on the
# first samples (the last ones take its mirror image, sign (-1)^derivative).  Numbers typed from the tables, NOT
# computed by this repository: they pin calc._fd_weights / _findiff_first_axis and the device stencil.
from fractions import Fraction as _F

FORNBERG = {
    (1, 2): dict(central=[_F(-1, 2), 0, _F(1, 2)], forward=[_F(-3, 2), 2, _F(-1, 2)]),
    (1, 4): dict(central=[_F(1, 12), _F(-2, 3), 0, _F(2, 3), _F(-1, 12)],
                 forward=[_F(-25, 12), 4, -3, _F(4, 3), _F(-1, 4)]),
    (1, 6): dict(central=[_F(-1, 60), _F(3, 20), _F(-3, 4), 0, _F(3, 4), _F(-3, 20), _F(1, 60)],
                 forward=[_F(-49, 20), 6, _F(-15, 2), _F(20, 3), _F(-15, 4), _F(6, 5), _F(-1, 6)]),
    (2, 2): dict(central=[1, -2, 1], forward=[2, -5, 4, -1]),
    (2, 4): dict(central=[_F(-1, 12), _F(4, 3), _F(-5, 2), _F(4, 3), _F(-1, 12)],
                 forward=[_F(15, 4), _F(-77, 6), _F(107, 6), -13, _F(61, 12), _F(-5, 6)]),
}


def fornberg_apply(x, h, deriv, acc):
    """The derivative findiff computes, from the hard-coded tables alone: central stencil inside, the forward stencil on
    the first `half` samples, its mirror image on the last ones."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    c = np.array([float(v) for v in FORNBERG[(deriv, acc)]["central"]])
    f = np.array([float(v) for v in FORNBERG[(deriv, acc)]["forward"]])
    half = len(c) // 2
    out = np.empty_like(x)
    for i in range(n):
        if i < half:
            out[i] = f @ x[i:i + len(f)]
        elif i >= n - half:
            out[i] = ((-1) ** deriv) * (f @ x[i - len(f) + 1:i + 1][::-1])
        else:
            out[i] = c @ x[i - half:i + half + 1]
    return out / h ** deriv
