"""CPU: host logic of the drop-in surface (filters, tail, velocity, envelope) against the oracle's
restatement of the reference and against scipy."""
import numpy as np
import pytest
import scipy.signal

import mfcc_oracle as O
from conftest import FORNBERG, fornberg_apply, load_golden
from modulation_mfcc_amd import applyFilter, calculate_amplitude_envelope, get_velocity
from modulation_mfcc_amd import tail
from modulation_mfcc_amd.filters import applyFilter as af2


def test_tail_on_golden_mfcc_matches_oracle_change():
    kw, y, exp = load_golden("refdefault_am")
    got = tail.mfcc_change(exp["mfcc"], tStep=0.005, outFiltCutOff=[12])
    np.testing.assert_allclose(got, exp["totChange"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(tail.time_anchors(exp["mfcc"].shape[1], 0.005, 0.025), exp["T"])


@pytest.mark.parametrize("kwargs", [
    dict(outFilter=None), dict(diffMethod="sg", outFiltCutOff=[10]),
    dict(outFilter="fir", outFiltCutOff=[20], outFiltLen=11), dict(outFilter="sg", outFiltCutOff=[1], outFiltLen=7),
    dict(removeFirst=0, outFiltCutOff=[12]), dict(outFilter="iir", outFiltType="band", outFiltCutOff=[2, 20]),
])
def test_tail_variants_match_oracle(kwargs):
    kw, y, exp = load_golden("refdefault_am")
    want = O.mfcc_change_tail(exp["mfcc"], tStep=0.005, **kwargs)
    got = tail.mfcc_change(exp["mfcc"], tStep=0.005, **kwargs)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13)


def test_apply_filter_errors_are_the_references():
    x = np.random.default_rng(0).standard_normal(300)
    assert applyFilter is af2
    with pytest.raises(Exception, match="CutOff is None"):
        applyFilter(x, 200.0, cutOff=None)
    with pytest.raises(Exception, match="filt is None"):
        applyFilter(x, 200.0, filt=None, cutOff=[12])
    with pytest.raises(Exception, match="filtType must be one among"):
        applyFilter(x, 200.0, cutOff=[12], filtType="notch")
    with pytest.raises(Exception, match="smaller than the half"):
        applyFilter(x, 200.0, cutOff=[100])
    with pytest.raises(Exception, match=r"cutOff\[0\]<cutOff\[1\]"):
        applyFilter(x, 200.0, cutOff=[30, 20], filtType="band")
    with pytest.raises(Exception, match="only one or two cut off"):
        applyFilter(x, 200.0, cutOff=[10, 20], filtType="low")
    with pytest.raises(Exception, match="sg .* can only be lowpass"):
        applyFilter(x, 200.0, filt="sg", cutOff=[10, 20], filtType="band")
    with pytest.raises(TypeError):
        applyFilter(x, sr=200.0, cutOff=[12])          # sr is positional-only, as in the reference


def test_apply_filter_values():
    x = np.random.default_rng(1).standard_normal(400)
    for kw in (dict(cutOff=[12]), dict(cutOff=[12], filtType="high"), dict(cutOff=[5, 30], filtType="band"),
               dict(filt="fir", cutOff=[12], filtLen=9), dict(filt="sg", cutOff=[1], filtLen=9)):
        np.testing.assert_allclose(applyFilter(x, 200.0, **kw), O.apply_filter(x, 200.0, **kw), rtol=1e-12, atol=1e-14)
    sos = scipy.signal.butter(4, 0.2, output="sos")
    np.testing.assert_allclose(applyFilter(x, 200.0, cutOff=[12], coeffs=sos), scipy.signal.sosfiltfilt(sos, x))


def test_get_velocity():
    t = np.arange(500) / 100.0
    x = np.sin(2 * np.pi * 1.5 * t)
    np.testing.assert_allclose(get_velocity(x, 100.0), np.gradient(x, 0.01))
    np.testing.assert_allclose(get_velocity(x, 100.0, difference=2), np.gradient(np.gradient(x, 0.01), 0.01))
    v = get_velocity(x, 100.0, method="sg", width=5, polyOrder=2)
    np.testing.assert_allclose(v, scipy.signal.savgol_filter(x, 5, 2, deriv=1, axis=0, mode="interp"))
    fd = get_velocity(x, 100.0, method="finDiff", accOrder=2)
    np.testing.assert_allclose(fd[1:-1], (x[2:] - x[:-2]) / 0.02, rtol=1e-10, atol=1e-10)
    assert fd[0] == pytest.approx((-3 * x[0] + 4 * x[1] - x[2]) / 0.02)
    fd2 = get_velocity(x, 100.0, difference=2, method="finDiff", accOrder=2)
    np.testing.assert_allclose(fd2[1:-1], (x[2:] - 2 * x[1:-1] + x[:-2]) / 1e-4, rtol=1e-8, atol=1e-6)
    exact = 2 * np.pi * 1.5 * np.cos(2 * np.pi * 1.5 * t)
    assert np.abs(get_velocity(x, 100.0, method="finDiff", accOrder=4) - exact)[5:-5].max() < 2e-3
    with pytest.raises(ValueError, match="Méthode inconnue"):
        get_velocity(x, 100.0, method="spline")


def test_amplitude_envelope_needs_the_gpu():
    """The envelope is computed by the device kernels (row N3); without a GPU the call fails loudly instead
    of falling back to a host path.  (GPU parity: tests/test_gpu_parity.py::test_amplitude_envelope_on_device.)"""
    import torch
    rng = np.random.default_rng(2)
    x = rng.standard_normal(4000).astype(np.float32)
    with pytest.raises(NotImplementedError):
        calculate_amplitude_envelope(x, 8000.0, method="RMSpraat")
    with pytest.raises(UnboundLocalError):
        calculate_amplitude_envelope(x, 8000.0, method="rms")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            calculate_amplitude_envelope(x, 8000.0, method="RMS", winLen=0.05, hopLen=0.01)


def _write_wav(path, data, sr, kind="int", bits=16, extensible=False):
    """Minimal RIFF/WAVE writer for the tests: data [n, ch] float in [-1, 1)."""
    import struct
    data = np.atleast_2d(np.asarray(data, dtype=np.float64).T).T
    n, ch = data.shape
    if kind == "float":
        raw = data.astype("<f4" if bits == 32 else "<f8").tobytes()
        tag = 3
    elif bits == 8:
        raw = np.clip(np.round(data * 128 + 128), 0, 255).astype(np.uint8).tobytes()
        tag = 1
    elif bits == 24:
        q = np.clip(np.round(data * (1 << 23)), -(1 << 23), (1 << 23) - 1).astype("<i4")
        raw = q.reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
        tag = 1
    else:
        q = np.clip(np.round(data * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
        raw = q.astype("<i2" if bits == 16 else "<i4").tobytes()
        tag = 1
    align = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, sr, sr * align, align, bits, 22, bits, 0) + guid
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, sr, sr * align, align, bits)
    junk = b"LIST" + struct.pack("<I", 5) + b"abcde\0"          # an odd-sized chunk before the data
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + junk + b"data" + struct.pack("<I", len(raw)) + raw
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return raw


def test_wav_header_parser(tmp_path):
    """Row N4, host part: the RIFF/WAVE parser (modulation_mfcc_amd/audio_io.py) on PCM 8 / 16 / 24 / 32, float
    32 / 64, WAVE_FORMAT_EXTENSIBLE, an odd-sized chunk in front of the data, mono and stereo."""
    from modulation_mfcc_amd.audio_io import read_wav_header
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.9, 0.9, (1000, 2))
    for kind, bits, ext, fmt in (("int", 8, False, 1), ("int", 16, False, 2), ("int", 24, False, 3), ("int", 32, False, 4),
                                 ("float", 32, False, 5), ("float", 64, False, 6), ("int", 24, True, 3), ("float", 32, True, 5)):
        for ch in (1, 2):
            p = str(tmp_path / f"t_{kind}{bits}_{ext}_{ch}.wav")
            raw = _write_wav(p, x[:, :ch], 22050, kind, bits, ext)
            h = read_wav_header(p)
            assert (h["sr"], h["channels"], h["bits"], h["kind"], h["fmt"]) == (22050.0, ch, bits, kind, fmt)
            assert h["n_frames"] == 1000 and h["data_bytes"] == len(raw)
            assert open(p, "rb").read()[h["data_offset"]:h["data_offset"] + 16] == raw[:16]
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFF\x04\x00\x00\x00WAVX")
    with pytest.raises(ValueError, match="not a RIFF/WAVE"):
        read_wav_header(str(bad))
    import struct
    short = tmp_path / "short_fmt.wav"                       # a fmt chunk of 8 bytes: ValueError, not struct.error
    body = b"WAVE" + b"fmt " + struct.pack("<I", 8) + b"\x01\x00\x01\x00\x22\x56\x00\x00" + b"data" + struct.pack("<I", 4) + b"\0\0\0\0"
    short.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    with pytest.raises(ValueError, match="truncated fmt chunk"):
        read_wav_header(str(short))


@pytest.mark.parametrize("sr_in,sr_out", [(48000, 16000), (44100, 16000), (44100, 10000), (16000, 10000), (8000, 16000)])
def test_resampler_quality(sr_in, sr_out):
    """Row N4: the low-pass the device resampler applies meets the soxr-HQ-class specification it is designed
    to (pass band to 0.913 of the lower Nyquist within 1e-4 dB, >= 120 dB from the Nyquist up), and polyphase
    conversion with it reproduces an in-band multi-tone signal -- sampled analytically at the new rate -- to
    better than 100 dB, while a tone above the new Nyquist disappears.  (The kernel itself is compared with
    this same arithmetic on the GPU: tests/test_gpu_parity.py::test_load_and_resample_on_device.)"""
    from modulation_mfcc_amd.audio_io import design_taps, resample_ratio
    L, M = resample_ratio(sr_in, sr_out)
    assert L * sr_in == M * sr_out
    h, half = design_taps(L, M)
    assert len(h) == 2 * half + 1 and abs(h.sum() - L) < 1e-9 and np.allclose(h, h[::-1])
    H = np.abs(np.fft.rfft(h, 1 << 20)) / L
    f = np.fft.rfftfreq(1 << 20)
    fN = 0.5 / max(L, M)
    pb, sb = H[f <= 0.913 * fN], H[f >= fN]
    assert 20 * np.log10(pb.max() / pb.min()) < 1e-4 and 20 * np.log10(sb.max()) < -120.0
    n = 6000
    t_in, n_out = np.arange(n) / sr_in, -(-n * L // M)
    t_out = np.arange(n_out) / sr_out
    f_lo = min(sr_in, sr_out) / 2
    tones = [0.05 * f_lo, 0.37 * f_lo, 0.9 * f_lo]
    sig = lambda t: sum(np.sin(2 * np.pi * fr * t + 0.3 * i) for i, fr in enumerate(tones))   # noqa: E731
    y = scipy.signal.resample_poly(sig(t_in), L, M, window=h / L)      # scipy multiplies explicit taps by `up`
    mid = slice(half // M + 50, n_out - half // M - 50)                # away from the zero-padded clip ends
    err = np.abs(y - sig(t_out))[mid].max()
    assert 20 * np.log10(err / 3.0) < -100.0
    if sr_out < sr_in:
        alias = scipy.signal.resample_poly(np.sin(2 * np.pi * 1.2 * f_lo * t_in), L, M, window=h / L)
        assert 20 * np.log10(np.abs(alias[mid]).max()) < -115.0


@pytest.mark.parametrize("sr_in,sr_out", [(44100, 16000), (44100, 10000), (48000, 16000), (16000, 10000), (8000, 16000),
                                          (22050, 16000), (16000, 44100)])
def test_resampler_as_a_banded_gemm_on_the_host(sr_in, sr_out):
    """Row N4, host part of the matrix-pipe resampler: audio_io.banded_tables lays the polyphase FIR out as the banded
    GEMM mm_resample_banded_f32 multiplies (outputs m = q F + 16 b + r against the window x[q S + lo_b + k]); applied with
    numpy (banded_resample_numpy: exactly the kernel's index arithmetic, float64 sums) it reproduces
    scipy.signal.resample_poly with the same taps on clips shorter than the filter, ragged lengths and long clips;
    the MFMA lane order of the A table is the documented one."""
    from modulation_mfcc_amd.audio_io import banded_resample_numpy, banded_tables, design_taps, resample_ratio
    L, M = resample_ratio(sr_in, sr_out)
    h, half = design_taps(L, M)
    h = h.astype(np.float32).astype(np.float64)
    tb = banded_tables(L, M, h, half)
    F, S, NB, ks = tb["F"], tb["S"], tb["NB"], tb["ksteps"]
    assert F >= 16 and F % L == 0 and S * L == F * M and NB == -(-F // 16) and 16 * NB <= 1.13 * F
    assert ks % 8 == 0 and tb["atab"].shape == (NB, ks // 4, 64, 4) and tb["lo_off"].shape == (NB,) and tb["lo_off"][0] == 0
    assert tb["win"] == int(tb["lo_off"].max()) + 4 * ks
    from modulation_mfcc_amd.audio_io import BANDED_KOFF
    for b in (0, NB - 1):       # lane l of k-step 8 g + s holds A[b][r = l % 16][k = 32 g + KOFF[l // 16] + s]
        for s_ in (0, 3, ks // 2 + 1, ks - 1):
            for lane in (0, 17, 35, 63):
                k_ = 32 * (s_ // 8) + BANDED_KOFF[lane // 16] + s_ % 8
                assert tb["atab"][b, s_ // 4, lane, s_ % 4] == tb["A"][b, lane % 16, k_]
    # every row of every block carries all of its phase's taps: row sums = the polyphase branches' DC gains
    u = np.arange(16 * NB)
    ph = (u * M + half) % L
    want_dc = np.array([h[p_::L].sum() for p_ in ph])
    np.testing.assert_allclose(tb["A"].astype(np.float64).sum(axis=2).reshape(-1), want_dc, rtol=0, atol=1e-5)
    rng = np.random.default_rng(1)
    for n in (1, 5, 333, 4001, 20000):
        x = rng.standard_normal(n)
        n_out = -(-n * L // M)
        want = scipy.signal.resample_poly(x, L, M, window=h / L)
        got = banded_resample_numpy(x, tb, n_out)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0) + 1e-13, (n, np.abs(got - want).max())


def _stencil_numpy(st, y):
    n = len(y)
    out = np.empty(n)
    for i in range(n):
        if i < st["n_edge"]:
            out[i] = np.dot(st["el"][i], y[:st["edge_w"]]) / st["den_e"]
        elif i >= n - st["n_edge"]:
            out[i] = np.dot(st["er"][i - (n - st["n_edge"])], y[n - st["edge_w"]:]) / st["den_e"]
        else:
            out[i] = sum(c * y[i + o] for c, o in zip(st["c"], st["off"])) / st["den_c"]
    return out


@pytest.mark.parametrize("n_taps,kind,cut", [(6, "lowpass", [12.0]), (2, "lowpass", [20.0]), (8, "lowpass", [30.0]),
                                             (7, "highpass", [15.0]), (5, "bandpass", [8.0, 25.0])])
def test_fir_filtfilt_stencil_on_the_host(n_taps, kind, cut):
    """applyFilter(filt='fir') = scipy.signal.filtfilt(firwin taps, 1, x) as ONE banded operator
    (filters.fir_filtfilt_stencil): applied with numpy it reproduces scipy's two-pass result to round-off,
    including the odd-extension edges.  The kernel applies exactly this arithmetic (GPU test
    test_fir_filter_on_device)."""
    from modulation_mfcc_amd.filters import fir_filtfilt_stencil
    taps = scipy.signal.firwin(n_taps, np.asarray(cut) / 50.0, window=("kaiser", 7.4), pass_zero=kind)
    st = fir_filtfilt_stencil(taps)
    assert len(st["c"]) <= 16 and st["n_edge"] <= 8 and st["edge_w"] <= 16
    rng = np.random.default_rng(3)
    for n in (3 * n_taps + 1, 60, 333):
        x = rng.standard_normal(n).cumsum() + 5.0
        want = scipy.signal.filtfilt(taps, 1, x)
        got = _stencil_numpy(st, x)
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1.0), (n, np.abs(got - want).max())
    with pytest.raises(NotImplementedError):
        fir_filtfilt_stencil(np.ones(9) / 9)


@pytest.mark.parametrize("kw", [
    dict(method="gradient", difference=1), dict(method="gradient", difference=2),
    dict(method="sg", width=3, polyOrder=2, difference=1), dict(method="sg", width=7, polyOrder=3, difference=2),
    dict(method="sg", width=11, polyOrder=2, difference=1), dict(method="sg", width=9, polyOrder=3, difference=0),
    # even windows (scipy centres them at width // 2): 6 / 3 is the reference's default outFiltLen / outFiltPolyOrd
    dict(method="sg", width=6, polyOrder=3, difference=0), dict(method="sg", width=6, polyOrder=3, difference=1),
    dict(method="sg", width=4, polyOrder=2, difference=1), dict(method="sg", width=8, polyOrder=3, difference=2),
    dict(method="sg", width=16, polyOrder=4, difference=0),
    dict(method="finDiff", difference=1, accOrder=2), dict(method="finDiff", difference=2, accOrder=4),
    dict(method="finDiff", difference=1, accOrder=6),
])
def test_velocity_stencils_on_the_host(kw):
    """Row N2, host part: the banded operator calc.velocity_stencil hands to mm_stencil_f64 -- interior taps, dense
    rows for the first / last samples, denominators -- applied with numpy reproduces get_velocity's host
    arithmetic (np.gradient exactly; scipy's savgol_filter(mode='interp') and the findiff stencils to round-off).
    The kernel applies exactly this arithmetic (tests/test_gpu_parity.py::test_velocity_on_device)."""
    from modulation_mfcc_amd.calc import velocity_stencil
    rng = np.random.default_rng(0)
    x = rng.standard_normal(200).cumsum()
    st, passes = velocity_stencil(200.0, **kw)
    assert len(st["c"]) <= 16 and st["n_edge"] <= 8 and st["edge_w"] <= 16
    assert len(st["off"]) == len(st["c"])
    y = x
    for _ in range(passes):
        y = _stencil_numpy(st, y)
    if kw["method"] == "sg" and kw["difference"] == 0:
        want = scipy.signal.savgol_filter(x, kw["width"], kw["polyOrder"], deriv=0, mode="interp")
    else:
        want = get_velocity(x, 200.0, **kw)
    if kw["method"] == "gradient":
        np.testing.assert_array_equal(y, want)
    else:
        assert np.abs(y - want).max() <= 1e-13 * np.abs(want).max()
    with pytest.raises(NotImplementedError):
        velocity_stencil(200.0, 1, "sg", 21, 2, 3)


@pytest.mark.parametrize("deriv,acc", sorted(FORNBERG))
def test_findiff_branch_is_pinned_by_the_published_tables(deriv, acc):
    """get_velocity(method='finDiff') restates findiff.FinDiff(0, 1/sr, difference, acc=accOrder) (script/calc.py:636;
    findiff is not installed).  Pin: (a) the weights calc._fd_weights solves for equal Fornberg's published tables
    (conftest.FORNBERG, typed in as fractions); (b) the whole curve, edges included, equals the tables applied by
    hand; (c) a polynomial of degree <= the stencil's exactness comes out with its exact derivative at EVERY sample
    (interior and one-sided ends); (d) the banded operator handed to the device carries the same numbers."""
    from modulation_mfcc_amd.calc import _fd_weights, velocity_stencil
    tab = FORNBERG[(deriv, acc)]
    half = len(tab["central"]) // 2
    np.testing.assert_allclose(_fd_weights(np.arange(-half, half + 1), deriv), [float(v) for v in tab["central"]],
                               rtol=0, atol=2e-13)
    nf = len(tab["forward"])
    np.testing.assert_allclose(_fd_weights(np.arange(0, nf), deriv), [float(v) for v in tab["forward"]], rtol=0, atol=5e-12)
    rng = np.random.default_rng(deriv * 10 + acc)
    sr = 200.0
    for n in (nf + half, 40, 201):
        x = rng.standard_normal(n).cumsum()
        want = fornberg_apply(x, 1 / sr, deriv, acc)
        got = get_velocity(x, sr, difference=deriv, method="finDiff", accOrder=acc)
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    # exactness: central weights differentiate degree <= 2*half exactly, the one-sided ones degree <= nf - 1
    t = np.arange(60) / sr
    deg = min(2 * half, nf - 1)
    coef = rng.standard_normal(deg + 1)
    poly = np.polynomial.Polynomial(coef)
    got = get_velocity(poly(t), sr, difference=deriv, method="finDiff", accOrder=acc)
    exact = poly.deriv(deriv)(t)
    assert np.abs(got - exact).max() <= 1e-8 * max(1.0, np.abs(exact).max())
    # the device operator: interior taps and the first / last rows are the table's numbers
    st, passes = velocity_stencil(sr, deriv, "finDiff", accOrder=acc)
    assert passes == 1 and st["off"] == list(range(-half, half + 1)) and st["n_edge"] == half
    np.testing.assert_allclose(st["c"], [float(v) for v in tab["central"]], rtol=0, atol=2e-13)
    for i in range(half):
        row_l = np.zeros(st["edge_w"]); row_l[i:i + nf] = [float(v) for v in tab["forward"]]
        np.testing.assert_allclose(st["el"][i], row_l, rtol=0, atol=5e-12)
        row_r = np.zeros(st["edge_w"])
        pos = st["edge_w"] - half + i
        row_r[pos - nf + 1:pos + 1] = ((-1) ** deriv) * np.array([float(v) for v in tab["forward"]])[::-1]
        np.testing.assert_allclose(st["er"][i], row_r, rtol=0, atol=5e-12)
    assert st["den_c"] == st["den_e"] == (1 / sr) ** deriv


def test_iir_design_cache_does_not_accept_what_scipy_rejects():
    """the cached Butterworth design is keyed on int(filtLen); a non-integral order must reach
    scipy.signal.butter -- the reference's own call (script/mfcc.py:98-101) -- and raise ITS error, not be truncated."""
    from modulation_mfcc_amd import filters
    a = filters.iir_sos(200.0, cutOff=[12], filtLen=6)
    np.testing.assert_array_equal(a, scipy.signal.butter(6, 12 / 100.0, btype="lowpass", output="sos"))
    np.testing.assert_array_equal(filters.iir_sos(200.0, cutOff=[12], filtLen=np.int64(6)), a)
    for bad in (6.5, 5.999, True):
        try:
            want = scipy.signal.butter(bad, 12 / 100.0, btype="lowpass", output="sos")
        except Exception as e:                 # what the reference's call raises for this order
            with pytest.raises(type(e), match="Filter order"):
                filters.iir_sos(200.0, cutOff=[12], filtLen=bad)
        else:                                  # scipy accepts it (True == order 1): the same sections, uncached
            np.testing.assert_array_equal(filters.iir_sos(200.0, cutOff=[12], filtLen=bad), want)


def test_bench_names_the_kernel_instantiation_of_a_workload():
    """bench.py reads a kernel's HBM traffic from the committed counter summaries by kernel NAME: the staged-sample kernel's
    instantiations differ by their staging groups (NR follows the hop: DESIGN.md 4.0), so BASELINE configs[1] / [2] (hop 160: NR 3)
    and the reference's default call (hop 50: NR 1) must not share a key -- they did, and configs[1] was reported with the
    reference default's traffic."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from modulation_mfcc_amd import MfccConfig
    c16 = MfccConfig(**bench.WORKLOADS["c3"][4])
    rd = MfccConfig(**bench.WORKLOADS["refdefault"][4])
    assert bench.s16_kernel_key(c16, 1) == "logmel512s_kernel<1, 3,"
    assert bench.s16_kernel_key(c16, 2) == "logmel512s_kernel<2, 3,"
    assert bench.s16_kernel_key(rd, 1) == "logmel512s_kernel<1, 1,"
    for hop, nr in ((56, 1), (57, 2), (121, 2), (122, 3), (186, 3), (187, 4), (252, 4)):
        assert bench.s16_kernel_key(MfccConfig(**dict(bench.WORKLOADS["c3"][4], hop_length=hop)), 1) == f"logmel512s_kernel<1, {nr},"
    assert bench.s16_kernel_key(MfccConfig(**dict(bench.WORKLOADS["refdefault"][4], preemph=0.97)), 1) == "logmel512s_kernel<1, 3,"
