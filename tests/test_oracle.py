"""CPU: the oracle against the committed golden vectors and against first principles."""
import numpy as np
import pytest
import scipy.fftpack
import scipy.signal

import mfcc_oracle as O
from conftest import GOLDEN_NAMES, load_golden, mfcc_close


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_reproduces_golden(name):
    kw, y, exp = load_golden(name)
    cfg = O.OracleConfig(**kw)
    m = O.mfcc(y, cfg)
    assert m.dtype == np.float32 and m.shape == exp["mfcc"].shape
    assert m.shape == (cfg.n_mfcc, O.num_frames(len(y), cfg.hop_length))
    np.testing.assert_allclose(m, exp["mfcc"], rtol=0, atol=2e-4 * max(1.0, np.abs(exp["mfcc"]).max() * 1e-2))
    if "modspec" in exp:
        np.testing.assert_allclose(O.modspec(m), exp["modspec"], rtol=1e-5, atol=1e-2)


def test_mel_matches_definition():
    # Slaney scale anchors: 1000 Hz <-> 15 mel, linear below, log above
    assert abs(O.hz_to_mel(1000.0) - 15.0) < 1e-12
    assert abs(O.mel_to_hz(15.0) - 1000.0) < 1e-9
    assert abs(O.hz_to_mel(200.0 / 3 * 4) - 4.0) < 1e-12
    f = np.array([50.0, 999.0, 1000.0, 4000.0, 7999.0])
    np.testing.assert_allclose(O.mel_to_hz(O.hz_to_mel(f)), f, rtol=1e-12)
    W = O.mel_filterbank(16000, 512, 40, 100.0, 8000.0)
    assert W.shape == (40, 257) and W.dtype == np.float32 and (W >= 0).all()
    assert ((W > 0).sum(0) <= 2).all()            # <= 2 filters per bin: the sparse path's premise
    assert int((W > 0).sum()) == 484
    # reference defaults at 10 kHz put fmax above Nyquist: 26 of 128 filters are empty
    Wd = O.mel_filterbank(10000, 512, 128, 100.0, 10000.0)
    assert int((Wd.max(1) == 0).sum()) == 26


def test_stft_power_against_direct_dft():
    rng = np.random.default_rng(0)
    y = rng.standard_normal(1000).astype(np.float32)
    P = O.stft_power(y, 64, 16, 48)
    assert P.shape == (1 + 1000 // 16, 33)
    win = O.hann_window_padded(48, 64)
    yp = np.pad(y.astype(np.float64), (32, 32))
    t = 7
    fr = yp[t * 16:t * 16 + 64] * win
    k = np.arange(33)[:, None]
    X = (fr[None, :] * np.exp(-2j * np.pi * k * np.arange(64)[None, :] / 64)).sum(1)
    np.testing.assert_allclose(P[t], np.abs(X) ** 2, rtol=2e-5, atol=1e-6)


def test_power_to_db_clamp_is_per_clip():
    S = np.array([[1.0, 1e-12], [1e-3, 10.0]], dtype=np.float32)
    d = O.power_to_db(S)
    assert d.max() == pytest.approx(10.0)
    assert d.min() == pytest.approx(10.0 - 80.0)          # 1e-12 -> amin -> -100 dB -> clamped to -70
    assert O.power_to_db(S, top_db=None).min() == pytest.approx(-100.0)


def test_dct_is_scipy_ortho():
    x = np.random.default_rng(1).standard_normal((40, 7)).astype(np.float32)
    m = O.mfcc_from_logmel(x.T, 13)
    k = np.arange(13)[:, None]
    n = np.arange(40)[None, :]
    D = 2 * np.cos(np.pi * k * (2 * n + 1) / 80) * np.where(k == 0, np.sqrt(1 / 160), np.sqrt(1 / 80))
    np.testing.assert_allclose(m, D @ x, rtol=1e-4, atol=1e-4)


def test_change_tail_matches_reference_formula():
    kw, y, exp = load_golden("refdefault_am")
    tot, T = O.get_MFCCS_change(y, kw["sr"], tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=512,
                                minFreq=100, maxFreq=10000, outFiltCutOff=[12])
    np.testing.assert_allclose(tot, exp["totChange"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(T, exp["T"])
    assert T[0] == pytest.approx(0.005 + 0.0125) and len(T) == exp["mfcc"].shape[1]


def test_float32_fft_variant_within_tolerance():
    kw, y, exp = load_golden("c1_am")
    cfg = O.OracleConfig(**kw)
    a = O.mfcc(y, cfg, fft_dtype=np.float32)
    assert np.abs(a - exp["mfcc"]).max() <= 1e-5 * np.abs(exp["mfcc"]).max()


def test_rms_envelope():
    x = np.ones(1000, dtype=np.float32)
    r = O.rms_envelope(x, 100, 10)
    assert r.shape == (101,) and r[50] == pytest.approx(1.0) and r[0] == pytest.approx(np.sqrt(0.5))


def test_oracle_matches_the_independent_implementation():
    """tests/golden_xcheck/*.npz hold MFCCs computed by transformers.audio_utils (documented librosa
    compatibility) + scipy.fftpack.dct -- numbers this repository did not produce.  The oracle agrees to
    float32 round-off.  Evidence, not a pin (the reference ships no vectors)."""
    from conftest import XCHECK_NAMES, load_xcheck
    assert len(XCHECK_NAMES) >= 5
    for name in XCHECK_NAMES:
        kw, y, want = load_xcheck(name)
        got = O.mfcc(y, O.OracleConfig(**kw))
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max(), name


def test_float64_branch_of_the_oracle():
    """librosa keeps a float64 signal in double precision (complex128 STFT); librosa.load never produces one, an
    ndarray caller can.  mfcc_f64 restates that branch; it agrees with the float32 branch to float32 round-off, which
    is what lets the build compute every input in float32 (GPU test test_float64_input_is_computed_in_float32)."""
    kw, y, exp = load_golden("c1_am")
    cfg = O.OracleConfig(**kw)
    m64 = O.mfcc_f64(y.astype(np.float64), cfg)
    assert m64.dtype == np.float64 and m64.shape == exp["mfcc"].shape
    mfcc_close(exp["mfcc"], m64, "float32 branch vs float64 branch")
    kw2, y2, exp2 = load_golden("c1_quiet_tail")
    mfcc_close(exp2["mfcc"], O.mfcc_f64(y2.astype(np.float64), O.OracleConfig(**kw2)), "clamp case")
