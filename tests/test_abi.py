"""CPU: libmodmfcc.so loads, exports every symbol include/modmfcc.h declares, and its host-only
table builders reproduce the oracle's tables.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.fftpack
import scipy.signal

import mfcc_oracle as O
from conftest import ROOT
from modulation_mfcc_amd import _lib
from modulation_mfcc_amd.plan import MfccConfig, butter_sos


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "modmfcc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/modmfcc.h but not exported"
    assert set(syms) == set(_lib.PROTOTYPES), set(syms) ^ set(_lib.PROTOTYPES)
    assert lib.mm_version() == 123
    assert lib.mm_strerror(-2) == b"unsupported configuration"


def test_config_struct_layout_and_defaults():
    lib = _lib.load()
    c = _lib.mm_config()
    assert lib.mm_config_default(C.byref(c)) == 0
    # script/main.py:732-748 at 10 kHz: 25 ms / 5 ms, 13 MFCC, n_fft 512, 100..10000 Hz
    assert (c.sr, c.n_fft, c.win_length, c.hop_length, c.n_mels, c.n_mfcc) == (10000.0, 512, 250, 50, 128, 13)
    assert (c.fmin, c.fmax, c.center, c.n_mod_fft) == (100.0, 10000.0, 1, 0)
    assert c.top_db == 80.0 and abs(c.amin - 1e-10) < 1e-17
    assert lib.mm_config_validate(C.byref(c)) == 0
    assert lib.mm_num_frames(C.byref(c), 10000) == 201
    assert lib.mm_num_bins(C.byref(c)) == 257
    assert lib.mm_mod_fft_len(C.byref(c), 1001) == 1024


@pytest.mark.parametrize("bad,exc", [
    (dict(n_fft=16384), NotImplementedError), (dict(n_fft=1, win_length=1), ValueError),
    (dict(win_length=600), ValueError), (dict(hop_length=0), ValueError),
    (dict(n_mfcc=200), ValueError), (dict(fmax=50.0), ValueError), (dict(center=False), NotImplementedError),
    (dict(amin=0.0), ValueError), (dict(n_mod_fft=1000), NotImplementedError),
    (dict(n_mod_fft=1 << 25), NotImplementedError),
])
def test_validate_rejects(bad, exc):
    with pytest.raises(exc):
        MfccConfig(**bad).validate()


@pytest.mark.parametrize("n_fft", [400, 500, 600, 1000, 1536, 441, 251, 8192, 16, 2, 8191])
def test_any_integer_n_fft_is_a_valid_configuration(n_fft):
    """librosa takes any n_fft >= win_length and the reference's dialog lets the user type one
    (script/config_dialog.py:141,610): every integer in [2, 8192] validates (the non-power-of-two lengths run on
    the mixed-radix / Bluestein STFT kernel, csrc/mm_anyfft.hip.inc)."""
    c = MfccConfig(n_fft=n_fft, win_length=min(250, n_fft), n_mels=min(128, max(1, n_fft // 4)), n_mfcc=1).validate()
    assert c.n_bins == n_fft // 2 + 1
    # librosa: pad n_fft // 2 on both sides, 1 + (padded - n_fft) // hop frames -- one frame less at a hop boundary
    # when n_fft is odd (the oracle's frame_signal does the same)
    for n in (50, 99, 100, 101, 1000):
        assert c.num_frames(n) == 1 + (n + 2 * (n_fft // 2) - n_fft) // c.hop_length == O.frame_signal(np.zeros(n, np.float32), n_fft, c.hop_length).shape[0]


def test_reference_call_truncation():
    # script/mfcc.py:382-384: int(0.025*22050)=551, int(0.01*22050)=220
    c = MfccConfig.from_reference_call(22050, tStep=0.01, winLen=0.025, n_fft=1024)
    assert (c.win_length, c.hop_length, c.n_mels) == (551, 220, 128)
    c = MfccConfig.from_reference_call(10000, tStep=0.005, winLen=0.025)
    assert (c.win_length, c.hop_length, c.fmax) == (250, 50, 10000.0)
    assert c.num_frames(10000) == 201


CFGS = [
    dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
    dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0),
    dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0),
    dict(sr=22050, n_fft=1024, win_length=551, hop_length=220, n_mels=64, n_mfcc=20, fmin=0.0, fmax=11025.0),
    dict(sr=8000, n_fft=256, win_length=255, hop_length=3, n_mels=23, n_mfcc=23, fmin=20.0, fmax=3900.0),
    dict(sr=16000, n_fft=400, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
    dict(sr=10000, n_fft=1000, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0),
    dict(sr=44100, n_fft=441, win_length=441, hop_length=147, n_mels=30, n_mfcc=12, fmin=50.0, fmax=20000.0),
]


@pytest.mark.parametrize("kw", CFGS)
def test_host_tables_match_oracle(kw):
    c = MfccConfig(**kw)
    W = O.mel_filterbank(kw["sr"], kw["n_fft"], kw["n_mels"], kw["fmin"], kw["fmax"])
    got = c.mel_filterbank()
    assert got.shape == W.shape
    np.testing.assert_allclose(got, W, rtol=3e-7, atol=1e-12)
    assert ((got == 0) == (W == 0)).all()
    w = O.hann_window_padded(kw["win_length"], kw["n_fft"]).astype(np.float32)
    np.testing.assert_allclose(c.window(), w, rtol=0, atol=6e-8)
    D = scipy.fftpack.dct(np.eye(kw["n_mels"]), axis=0, type=2, norm="ortho")[:kw["n_mfcc"]]
    np.testing.assert_allclose(c.dct_matrix(), D, rtol=0, atol=3e-8)


def test_butter_sos_matches_scipy():
    for order in range(1, 11):
        for wn in (0.003, 0.048, 0.12, 0.3, 0.5, 0.66, 0.9):
            np.testing.assert_allclose(butter_sos(order, wn), scipy.signal.butter(order, wn, output="sos"),
                                       rtol=1e-12, atol=1e-14)
    with pytest.raises(ValueError):
        butter_sos(6, 1.5)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from modulation_mfcc_amd import get_MFCCS_change, MfccPlan
    with pytest.raises(RuntimeError):
        MfccPlan(MfccConfig())
    with pytest.raises(RuntimeError):
        get_MFCCS_change(np.zeros(1000, dtype=np.float32), 10000)


def test_package_does_not_import_oracle():
    import subprocess
    import sys
    code = ("import sys; import modulation_mfcc_amd, modulation_mfcc_amd.mfcc, modulation_mfcc_amd.calc;"
            "assert not any('oracle' in m for m in sys.modules), [m for m in sys.modules if 'oracle' in m]")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "modulation_mfcc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".inc")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower().replace("# oracle", ""), f


def _sweep_emulate(cfg, P, n_waves=8):
    """Python emulation of phase B of logmel512_kernel (mm_fft16.hip.inc) on one power row."""
    wlo, whi, d, part = cfg.mel_sweep(n_waves)
    out = np.full(cfg.n_mels, np.nan)
    for (kb, ke, m0, m1) in part:
        A = B = 0.0
        dd = m0 - 1
        for k in range(kb, ke):
            while dd < d[k]:
                if m0 <= dd < m1:
                    assert np.isnan(out[dd]); out[dd] = A
                A, B, dd = B, 0.0, dd + 1
            A += float(wlo[k]) * P[k]
            B += float(whi[k]) * P[k]
        while dd < m1:
            if dd >= m0:
                assert np.isnan(out[dd]); out[dd] = A
            A, B, dd = B, 0.0, dd + 1
    return out


@pytest.mark.parametrize("kw", CFGS + [
    dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=128, n_mfcc=13, fmin=0.0, fmax=8000.0),
    dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=3, n_mfcc=2, fmin=300.0, fmax=3000.0),
    dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=256, n_mfcc=13, fmin=0.0, fmax=8000.0),
])
def test_mel_sweep_equals_dense_matrix(kw):
    c = MfccConfig(**kw)
    W = c.mel_filterbank().astype(np.float64)
    wlo, whi, d, part = c.mel_sweep(8)
    assert (np.diff(d) >= 0).all() and d.min() >= -1 and d.max() <= c.n_mels - 1
    # the sweep tables hold exactly the matrix entries
    R = np.zeros_like(W)
    for k in range(c.n_bins):
        if wlo[k]:
            R[d[k], k] = wlo[k]
        if whi[k]:
            R[d[k] + 1, k] = whi[k]
    np.testing.assert_array_equal(R, W)
    # the per-wave partition covers every filter exactly once, in order
    assert part[0, 2] == 0 and part[-1, 3] == c.n_mels and (part[1:, 2] == part[:-1, 3]).all()
    P = np.random.default_rng(3).uniform(0.0, 2.0, c.n_bins)
    got = _sweep_emulate(c, P)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, W @ P, rtol=1e-12, atol=1e-15)
    # run form (what phase B of logmel512_kernel executes): emulate it on one padded power row
    hdr, grp, rpart = c.mel_runs(8)
    Pp = np.zeros(4 * ((c.n_bins + 3) // 4))
    Pp[:c.n_bins] = P
    out = np.full(c.n_mels, np.nan)
    for (r0, r1, m0, m1) in rpart:
        carry = 0.0
        for (k4, ng, go, dd) in hdr[r0:r1]:
            assert k4 % 4 == 0 and k4 + 4 * ng <= len(Pp)
            sa = sb = 0.0
            for g in range(ng):
                pv = Pp[k4 + 4 * g:k4 + 4 * g + 4]
                sa += float(grp[go + g, :4].astype(np.float64) @ pv)
                sb += float(grp[go + g, 4:].astype(np.float64) @ pv)
            if dd >= m0:
                assert np.isnan(out[dd]) and dd < m1
                out[dd] = carry + sa
            carry = sb
    assert not np.isnan(out).any()
    np.testing.assert_allclose(out, W @ P, rtol=1e-12, atol=1e-15)


def test_sample_count_bound():
    """The radix-16 kernels index samples with 32-bit offsets (4 * index in the wave-per-frame kernel): the C ABI
    refuses clips beyond 2^29 - 8192 samples (9 hours at 16 kHz) instead of wrapping.  No GPU: a NULL plan is
    rejected first, so the bound is read from the header comment and the source."""
    csrc = os.path.join(ROOT, "modulation_mfcc_amd", "csrc")
    src = open(os.path.join(csrc, "mm_api.hip")).read() + open(os.path.join(csrc, "mm_plan.h")).read()
    assert "#define MM_MAX_SAMPLES (((int64_t)1 << 29) - 8192)" in src
    assert "if (n_samples > MM_MAX_SAMPLES) return MM_ERR_INVALID_ARG;" in src


def test_iir_entry_points_validate_on_the_host():
    """mm_sosfiltfilt_f64 / _f32_f64 reject bad arguments before they touch the device; the workspace query is pure host
    arithmetic and covers both device forms (segmented rows up to 4 sections, time-major beyond)."""
    import numpy as np
    lib = _lib.load()
    sos = np.ascontiguousarray(np.array([[0.1, 0.2, 0.1, 1.0, -0.5, 0.2]]))
    for fn in (lib.mm_sosfiltfilt_f64, lib.mm_sosfiltfilt_f32_f64):
        assert fn(None, 4, 100, 100, sos.ctypes.data, 1, None, None, 0, None) == -1          # null pointers
    for rows, n in ((1, 22), (256, 160000), (1024, 1001), (3, 4800000)):
        ws = lib.mm_sosfiltfilt_workspace_bytes(rows, n)
        n_ext = n + 2 * 27
        seg = 8 * (1024 + 2 * rows * (-(-n_ext // 1088)) * 8 + rows * ((n_ext + 1) // 2 * 2))     # table, segment states, padded rows
        tm = 8 * (n + 2 * 99) * (-(-rows // 64) * 64)
        assert ws >= seg and ws >= tm and ws <= 2 * max(seg, tm)
    assert lib.mm_sosfiltfilt_workspace_bytes(0, 100) == 0
