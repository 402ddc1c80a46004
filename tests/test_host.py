"""CPU: host logic of the drop-in surface (filters, tail, velocity, envelope) against the oracle's
restatement of the reference and against scipy."""
import numpy as np
import pytest
import scipy.signal

import mfcc_oracle as O
from conftest import load_golden
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
