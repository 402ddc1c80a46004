"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden vectors.

Tolerance (north star / SURVEY 8(c)): MFCC within 1e-4 relative -- ``mfcc_close`` checks
max|a-b| <= 1e-4 * max|b| per clip AND |a-b| <= 1e-4*|b| + 1e-3 elementwise.
"""
import contextlib

import numpy as np
import pytest

import mfcc_oracle as O
from conftest import FORNBERG, GOLDEN_NAMES, XCHECK_NAMES, fornberg_apply, load_golden, load_xcheck, mfcc_close

pytestmark = pytest.mark.gpu


def _plan(kw, **extra):
    from modulation_mfcc_amd import MfccConfig, get_plan
    return get_plan(MfccConfig(**{**kw, **extra}))


def _dev(x, gpu):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(gpu)


def test_native_library_is_loaded(gpu):
    from modulation_mfcc_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libmodmfcc.so" in maps


class _variant:
    """Pin the fused-kernel variant for the calls inside (mm_plan_set_variant): 'm12' (default where it
    applies: 12 waves, mel + DCT on the matrix pipe), 'w16s' (16 waves, samples staged through LDS), 'w16'
    (16 waves, direct loads), 'w8' (the 8-wave kernel, otherwise only used when the mel table is too big
    for w16), 'wpf' (wave per frame group) or 'generic'.  Configurations a variant does not cover fall
    through to the next one."""

    def __init__(self, plan, which):
        self.plan, self.which = plan, which

    def __enter__(self):
        self.old = self.plan.set_variant(None if self.which == "generic" else self.which)
        self.plan.force_generic(self.which == "generic")
        return self

    def __exit__(self, *a):
        self.plan.force_generic(False)
        self.plan.set_variant(self.old)


VARIANTS = ["m12", "w16s", "h32", "w16", "w8", "wpf", "generic"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_mfcc_matches_golden(name, variant, gpu):
    kw, y, exp = load_golden(name)
    plan = _plan(kw)
    with _variant(plan, variant):
        got = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
    mfcc_close(got, exp["mfcc"], f"{name} {variant}")


@pytest.mark.parametrize("name", XCHECK_NAMES)
def test_mfcc_matches_the_independent_implementation(name, gpu):
    """The HIP path against MFCCs of transformers.audio_utils + scipy.fftpack.dct (tests/golden_xcheck, written
    by oracle/crosscheck_transformers.py --write): numbers neither the oracle nor the kernels produced."""
    kw, y, want = load_xcheck(name)
    got = _plan(kw).mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
    mfcc_close(got, want, f"xcheck {name}")


@pytest.mark.parametrize("n", [4, 8, 160, 252, 256, 260, 512, 10236, 10240, 10244, 16000, 40964])
def test_staged_kernel_lengths(n, gpu):
    """The staged-sample kernel (w16s) on lengths around its edges (clip shorter than the centre pad,
    exactly one tile of 64 frames, one frame more / fewer), several clips per launch so that tiles of
    different clips follow each other in one workgroup: == direct-load kernel == oracle."""
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-w16s"
    clips = np.stack([O.synth_clip(700 + n + i, n, 16000, k) for i, k in enumerate(["am", "noise", "am", "noise", "impulse"])])
    d = _dev(clips, gpu)
    got = plan.mfcc(d).cpu().numpy()
    P = plan.stft_power(d).cpu().numpy()
    with _variant(plan, "w16"):
        ref = plan.mfcc(d).cpu().numpy()
        Pr = plan.stft_power(d).cpu().numpy()
    np.testing.assert_array_equal(P, Pr)       # same arithmetic as the direct-load kernel: bit-identical
    lm, lmr = plan.logmel(d)[0].cpu().numpy(), None
    with _variant(plan, "w16"):
        lmr = plan.logmel(d)[0].cpu().numpy()
    np.testing.assert_array_equal(lm, lmr)     # ... up to the log-mel rows; the staged kernel then applies the DCT
    for i in range(clips.shape[0]):            # itself (f32 MFMA chain), the direct-load one in dct_clamp_kernel
        mfcc_close(got[i], ref[i], f"w16s vs w16 n={n} clip {i}")
    for i in range(clips.shape[0]):
        mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**kw)), f"w16s n={n} clip {i}")


@pytest.mark.parametrize("n", [5, 6, 7, 161, 1022, 10241, 16001, 16002, 16003, 40963])
def test_staged_kernel_unaligned(n, gpu):
    """Lengths that are not multiples of 4, odd row pitches and rows that start off a 16-byte boundary
    run on the UNAL instantiation of the staged kernel (4-byte aligned 16-byte loads, the group that
    straddles the clip end re-aligned): == oracle, and bit-identical to the direct-load kernel where
    that one applies (even pitch, 8-byte aligned rows)."""
    import torch
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    clips = np.stack([O.synth_clip(900 + n + i, n, 16000, k) for i, k in enumerate(["am", "noise", "am"])])
    want = [O.mfcc(c, O.OracleConfig(**kw)) for c in clips]
    for pad, lead in ((0, 0), (1, 0), (2, 2), (3, 1), (6, 3)):
        big = torch.zeros((3, lead + n + pad), dtype=torch.float32, device=gpu)
        big[:, lead:lead + n] = _dev(clips, gpu)
        view = big[:, lead:lead + n]
        got = plan.mfcc(view).cpu().numpy()
        for i in range(3):
            mfcc_close(got[i], want[i], f"unaligned n={n} pad={pad} lead={lead} clip {i}")
        if (lead + n + pad) % 2 == 0 and lead % 2 == 0 and n >= 2:
            lm = plan.logmel(view)[0].cpu().numpy()
            with _variant(plan, "w16"):
                ref = plan.mfcc(view).cpu().numpy()
                lmr = plan.logmel(view)[0].cpu().numpy()
            np.testing.assert_array_equal(lm, lmr)
            for i in range(3):
                mfcc_close(got[i], ref[i], f"unaligned w16s vs w16 n={n} pad={pad} lead={lead} clip {i}")


def test_staged_kernel_large_hop(gpu):
    """hop 240 needs four 16-byte groups per thread (NR = 4); hop 200 with n_fft 512 / win 512."""
    for hop, win in ((240, 480), (200, 512), (184, 400), (188, 400)):
        kw = dict(sr=16000, n_fft=512, win_length=win, hop_length=hop, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
        plan = _plan(kw)
        assert plan.kernel_path == "radix16-w16s"
        clips = np.stack([O.synth_clip(900 + hop + i, 48000, 16000, "am") for i in range(3)])
        got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
        for i in range(3):
            mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**kw)), f"hop {hop} clip {i}")


@pytest.mark.parametrize("n", [4, 8, 160, 252, 7680, 7684, 10236, 16000, 40964, 5, 161, 10241, 16003])
def test_matrix_pipe_kernel_lengths(n, gpu):
    """The 12 + 4 wave kernel with the mel contraction and the DCT on the matrix pipe ('m12', opt-in):
    lengths around its edges (clip shorter than the centre pad, exactly one 48-frame tile, one frame more),
    lengths that are not multiples of 4 (register-staged instantiation instead of LDS-DMA), several clips
    per launch incl. one that forces the top_db clamp (the fix-up kernel) and silence: == oracle, and
    == the run-table kernel within float32 round-off."""
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    kinds = ["am", "noise", "quiet_tail", "silence", "impulse"]
    clips = np.stack([O.synth_clip(1700 + n + i, n, 16000, k) for i, k in enumerate(kinds)])
    d = _dev(clips, gpu)
    with _variant(plan, "m12"):
        assert plan.kernel_path == "radix16-m12"
        got = plan.mfcc(d).cpu().numpy()
        lm, mx = plan.logmel(d)
    ref = plan.mfcc(d).cpu().numpy()
    lmr, mxr = plan.logmel(d)
    np.testing.assert_allclose(lm.cpu().numpy(), lmr.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(mx.cpu().numpy(), mxr.cpu().numpy(), rtol=0, atol=2e-4)
    for i in range(clips.shape[0]):
        want = O.mfcc(clips[i], O.OracleConfig(**kw))
        mfcc_close(got[i], want, f"m12 n={n} {kinds[i]}")
        mfcc_close(got[i], ref[i], f"m12 vs w16s n={n} {kinds[i]}")


def test_matrix_pipe_kernel_configs(gpu):
    """'m12' beyond the benchmark shape: the reference's own defaults (128 mel = 8 filter blocks, 26 empty
    filters, hop 50), pre-emphasis, odd hops, unaligned row views, n_fft 256 embedded in 512, n_mfcc > 16
    (DCT in its own kernel), no clamp (top_db < 0: no log-mel rows are written), and a hop whose samples
    do not fit next to the double-buffered power tile (falls through to the run-table kernel)."""
    import torch
    cases = [
        (load_golden("refdefault_am")[0], 10000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "preemph": 0.97}, 16000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "hop_length": 161}, 16000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "hop_length": 77, "preemph": 0.95}, 16000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "n_mels": 64, "n_mfcc": 20, "hop_length": 128}, 16000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "top_db": -1.0}, 16000, "radix16-m12"),
        (dict(sr=8000, n_fft=256, win_length=200, hop_length=80, n_mels=40, n_mfcc=13, fmin=50.0, fmax=4000.0), 8000, "radix16-m12"),
        ({**load_golden("c1_am")[0], "hop_length": 200}, 16000, "radix16-w16s"),
    ]
    for kw, sr, path in cases:
        plan = _plan(kw)
        okw = {**kw, "top_db": None} if kw.get("top_db", 80.0) < 0 else kw     # the C ABI's "< 0 = no clamp"
        with _variant(plan, "m12"):
            assert plan.kernel_path == path, kw
            for n in (4801, 12000, 30000):
                clips = np.stack([O.synth_clip(40 + n + i, n, sr, k) for i, k in enumerate(["am", "noise", "quiet_tail"])])
                for pad, lead in ((0, 0), (3, 1)):
                    big = torch.zeros((3, lead + n + pad), dtype=torch.float32, device=gpu)
                    big[:, lead:lead + n] = _dev(clips, gpu)
                    got = plan.mfcc(big[:, lead:lead + n]).cpu().numpy()
                    for i in range(3):
                        mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**okw)), f"m12 {kw} n={n} pad={pad} clip {i}")


def test_matrix_pipe_kernel_full_size(gpu):
    """'m12' at BASELINE configs[1] size (1024 x 10 s): permutation equivariance (tiles of different clips
    follow each other in one workgroup; the DCT partial slots and the arrival counters are reused every
    tile), run-to-run bit-exactness, spot clips vs the oracle, and the clamp fix-up on half of the clips."""
    import torch
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    g = torch.Generator(device=gpu).manual_seed(12)
    audio = 0.05 * torch.randn((1024, 160000), generator=g, device=gpu)
    audio[::2, 80000:] = 0.0                      # digital silence: these clips clamp
    with _variant(plan, "m12"):
        m = plan.mfcc(audio)
        assert torch.equal(plan.mfcc(audio), m)
        perm = torch.randperm(1024, device=gpu, generator=g)
        assert torch.equal(plan.mfcc(audio[perm].contiguous()), m[perm])
    ref = plan.mfcc(audio)
    assert ((m - ref).abs().max() / ref.abs().max()).item() < 5e-6
    for i in (0, 1, 511, 1023):
        mfcc_close(m[i].cpu().numpy(), O.mfcc(audio[i].cpu().numpy(), O.OracleConfig(**kw)), f"m12 full size clip {i}")


def test_preemphasis_on_the_staged_kernel(gpu):
    """Pre-emphasis (the build's optional y[n] - a y[n-1]) is applied while the staged kernel stages its
    samples: same result as the generic kernel (which filters on load) and as the oracle, incl. clip
    starts inside and at the edge of a tile and several clips per launch."""
    kw, _, _ = load_golden("c1_am")
    kw = {**kw, "preemph": 0.97}
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-w16s"
    for n in (4, 512, 10240, 16000, 40964):
        clips = np.stack([O.synth_clip(80 + n + i, n, 16000, k) for i, k in enumerate(["am", "noise", "impulse"])])
        d = _dev(clips, gpu)
        got = plan.mfcc(d).cpu().numpy()
        P = plan.stft_power(d).cpu().numpy()
        with _variant(plan, "generic"):
            gen = plan.mfcc(d).cpu().numpy()
            Pg = plan.stft_power(d).cpu().numpy()
        np.testing.assert_allclose(P, Pg, rtol=2e-4, atol=1e-5 * max(Pg.max(), 1e-30))
        for i in range(clips.shape[0]):
            want = O.mfcc(clips[i], O.OracleConfig(**kw))
            mfcc_close(got[i], want, f"staged preemph n={n} clip {i}")
            mfcc_close(gen[i], want, f"generic preemph n={n} clip {i}")
    # unaligned input: no staged kernel -> generic
    odd = _dev(np.stack([O.synth_clip(3, 16002, 16000, "am")]), gpu)
    mfcc_close(plan.mfcc(odd)[0].cpu().numpy(), O.mfcc(odd[0].cpu().numpy(), O.OracleConfig(**kw)), "preemph n%4 != 0")
    # n_fft 2048 / 1024 and n_fft 512 on the wave-per-frame kernel: it filters on its frame loads
    for name, variant in (("c4_am", "wpf"), ("odd_22k", "wpf"), ("c1_am", "wpf")):
        kw2 = {**load_golden(name)[0], "preemph": 0.95}
        plan2 = _plan(kw2)
        with _variant(plan2, variant):
            assert plan2.kernel_path == "radix16-wpf"
            for n in (2, 1023, 4801, 24001, 30000):
                clips = np.stack([O.synth_clip(500 + n + i, n, kw2["sr"], k) for i, k in enumerate(["am", "noise"])])
                got = plan2.mfcc(_dev(clips, gpu)).cpu().numpy()
                for i in range(2):
                    mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**kw2)), f"wpf preemph {name} n={n} clip {i}")


def test_odd_hop_on_the_staged_kernel(gpu):
    """An odd hop puts odd frames at 4-byte aligned samples only: the staged kernel then reads its
    samples from LDS as two b32 instead of one b64; the other n_fft 512 kernels need an even hop."""
    for hop, pre in ((161, 0.0), (77, 0.0), (11, 0.97), (251, 0.0)):
        kw = dict(sr=16000, n_fft=512, win_length=400, hop_length=hop, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0,
                  preemph=pre)
        plan = _plan(kw)
        assert plan.kernel_path == "radix16-w16s"
        clips = np.stack([O.synth_clip(600 + hop + i, 20000, 16000, k) for i, k in enumerate(["am", "noise", "am"])])
        got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
        for i in range(3):
            mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**kw)), f"odd hop {hop} clip {i}")
        with _variant(plan, "w16"):
            assert plan.kernel_path == "radix16-wpf"      # the direct-load kernel needs an even hop: wave-per-frame kernel instead


def test_kernel_variants_selected(gpu):
    kw, _, _ = load_golden("c1_am")
    assert _plan(kw).kernel_path == "radix16-w16s"
    assert _plan(load_golden("refdefault_am")[0]).kernel_path == "radix16-w16s"
    assert _plan({**kw, "hop_length": 256}).kernel_path == "radix16-w16"   # 63*hop + 512 samples > 64 KB of LDS
    with _variant(_plan(kw), "w16"):
        assert _plan(kw).kernel_path == "radix16-w16"
    with _variant(_plan(kw), "h32"):
        assert _plan(kw).kernel_path == "radix16-h32" and not _plan(kw).fused_dct
    assert _plan({**kw, "n_mels": 256}).kernel_path == "radix16-w8"      # mel table too big for w16
    assert _plan(load_golden("c4_am")[0]).kernel_path == "radix16-wpf"     # n_fft 2048: wave-per-frame kernel
    assert _plan(load_golden("odd_22k")[0]).kernel_path == "radix16-wpf"   # n_fft 1024, even hop
    assert _plan(load_golden("ragged_preemph")[0]).kernel_path == "radix16-w16s"  # odd hop + pre-emphasis: staged kernel only
    assert _plan({**load_golden("c4_am")[0], "hop_length": 481}).kernel_path == "radix16-wpf"   # any hop parity


@pytest.mark.parametrize("variant", VARIANTS)
def test_stage_outputs_match_oracle(variant, gpu):
    kw, y, exp = load_golden("c1_am")
    plan = _plan(kw)
    with _variant(plan, variant):
        d = _dev(y, gpu)[None, :]
        P = plan.stft_power(d)[0].cpu().numpy()
        lm, mx = plan.logmel(d)
        lm, mx = lm[0].cpu().numpy(), float(mx[0])
    Pw = O.stft_power(y, kw["n_fft"], kw["hop_length"], kw["win_length"])
    assert P.shape == Pw.shape
    np.testing.assert_allclose(P, Pw, rtol=2e-4, atol=1e-5 * Pw.max())
    np.testing.assert_allclose(P[:8], exp["power_first8"], rtol=2e-4, atol=1e-5 * Pw.max())
    want = exp["logmel_unclamped"].T           # [n_mels, T]
    np.testing.assert_allclose(lm, want, rtol=0, atol=2e-3)
    assert mx == pytest.approx(float(want.max()), abs=2e-3)


@pytest.mark.parametrize("n", [2, 479, 480, 1023, 1025, 4801, 24000, 30001])
def test_nfft2048_kernel_vs_generic_and_oracle(n, gpu):
    """C4-shaped config (48 kHz, n_fft 2048, 80 mel, 40 MFCC): wave-per-frame kernel == generic == oracle
    on ragged lengths, incl. stage outputs."""
    kw, _, _ = load_golden("c4_am")
    y = O.synth_clip(2000 + n, n, kw["sr"], "am" if n > 5000 else "noise")
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-wpf"
    d = _dev(np.stack([y, y[::-1].copy()]), gpu)
    want = [O.mfcc(c, O.OracleConfig(**kw)) for c in (y, y[::-1])]
    fast = plan.mfcc(d).cpu().numpy()
    P = plan.stft_power(d).cpu().numpy()
    lm, mx = plan.logmel(d)
    with _variant(plan, "generic"):
        gen = plan.mfcc(d).cpu().numpy()
        Pg = plan.stft_power(d).cpu().numpy()
        lmg, mxg = plan.logmel(d)
    for i in range(2):
        mfcc_close(gen[i], want[i], f"generic n={n}")
        mfcc_close(fast[i], want[i], f"2048 n={n}")
    np.testing.assert_allclose(P, Pg, rtol=2e-4, atol=1e-5 * max(Pg.max(), 1e-30))
    np.testing.assert_allclose(lm.cpu().numpy(), lmg.cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(mx.cpu().numpy(), mxg.cpu().numpy(), rtol=0, atol=2e-3)


def test_wpf_unaligned_rows(gpu):
    """n_fft 2048 / 1024 on odd row pitches and rows starting off an 8-byte boundary (the frame loads are
    8-byte loads at 4-byte aligned addresses)."""
    import torch
    for name in ("c4_am", "odd_22k"):
        kw = load_golden(name)[0]
        plan = _plan(kw)
        assert plan.kernel_path == "radix16-wpf"
        for n in (4801, 24000, 24003):
            clips = np.stack([O.synth_clip(70 + n + i, n, kw["sr"], k) for i, k in enumerate(["am", "noise"])])
            want = [O.mfcc(c, O.OracleConfig(**kw)) for c in clips]
            for pad, lead in ((0, 0), (1, 0), (3, 1), (2, 3)):
                big = torch.zeros((2, lead + n + pad), dtype=torch.float32, device=gpu)
                big[:, lead:lead + n] = _dev(clips, gpu)
                got = plan.mfcc(big[:, lead:lead + n]).cpu().numpy()
                for i in range(2):
                    mfcc_close(got[i], want[i], f"{name} n={n} pad={pad} lead={lead} clip {i}")


@pytest.mark.parametrize("n_fft,win,hop,sr", [(256, 200, 80, 8000), (256, 256, 64, 8000), (128, 100, 25, 8000),
                                                (64, 64, 16, 8000), (256, 250, 50, 10000), (256, 201, 77, 16000)])
def test_small_nfft_on_the_512_kernels(n_fft, win, hop, sr, gpu):
    """n_fft 64 / 128 / 256 run on the n_fft 512 tile kernels (frame zero-padded to 512 points around its
    centre: same power at every (512/n_fft)-th bin): == oracle == generic kernel, clamp and all."""
    kw = dict(sr=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=min(40, n_fft // 4), n_mfcc=13 if n_fft > 64 else 8,
              fmin=50.0, fmax=sr / 2)
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-w16s"
    for n in (37, 4000, 16001):
        clips = np.stack([O.synth_clip(11 + n + i, n, sr, k) for i, k in enumerate(["am", "noise", "quiet_tail"])])
        d = _dev(clips, gpu)
        got = plan.mfcc(d).cpu().numpy()
        P = plan.stft_power(d).cpu().numpy()
        assert P.shape[-1] == n_fft // 2 + 1
        with _variant(plan, "generic"):
            gen = plan.mfcc(d).cpu().numpy()
        for i in range(3):
            want = O.mfcc(clips[i], O.OracleConfig(**kw))
            assert got[i].shape == want.shape
            mfcc_close(got[i], want, f"n_fft {n_fft} n={n} clip {i}")
            mfcc_close(gen[i], want, f"generic n_fft {n_fft} n={n} clip {i}")


def test_nfft512_calls_beyond_the_tile_kernels(gpu):
    """n_fft 512 with a hop too large for the staged kernel's LDS sample buffer (> 252): even hop and
    aligned rows run on the direct-load kernel, everything else on the wave-per-frame kernel (not the
    generic one)."""
    for hop, n in ((300, 30000), (301, 30000), (300, 30001), (257, 12345)):
        kw = dict(sr=16000, n_fft=512, win_length=400, hop_length=hop, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)
        plan = _plan(kw)
        clips = np.stack([O.synth_clip(33 + hop + i, n, 16000, k) for i, k in enumerate(["am", "noise"])])
        d = _dev(clips, gpu)
        got = plan.mfcc(d).cpu().numpy()
        with _variant(plan, "generic"):
            gen = plan.mfcc(d).cpu().numpy()
        for i in range(2):
            want = O.mfcc(clips[i], O.OracleConfig(**kw))
            mfcc_close(got[i], want, f"hop {hop} n={n} clip {i}")
            mfcc_close(gen[i], want, f"generic hop {hop} n={n} clip {i}")
        plan.timing_enable(True)
        plan.mfcc(d)
        plan.timing_enable(False)
        assert "logmel" in plan.timing_read()


def test_wpf_odd_hop(gpu):
    for name, hop in (("c4_am", 481), ("odd_22k", 221), ("c1_am", 161)):
        kw = {**load_golden(name)[0], "hop_length": hop}
        plan = _plan(kw)
        with _variant(plan, "wpf"):
            assert plan.kernel_path == "radix16-wpf"
            for n in (2, 3, 4800, 24001, 30000):
                clips = np.stack([O.synth_clip(170 + n + i, n, kw["sr"], k) for i, k in enumerate(["am", "noise"])])
                got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
                for i in range(2):
                    mfcc_close(got[i], O.mfcc(clips[i], O.OracleConfig(**kw)), f"wpf odd hop {name} n={n} clip {i}")


def test_nfft2048_batch_determinism(gpu):
    import torch
    kw, _, _ = load_golden("c4_am")
    plan = _plan(kw)
    g = torch.Generator(device=gpu).manual_seed(3)
    audio = 0.1 * torch.randn((37, 48000), generator=g, device=gpu)
    m = plan.mfcc(audio)
    perm = torch.randperm(37, device=gpu, generator=g)
    assert torch.equal(plan.mfcc(audio[perm].contiguous()), m[perm])
    for i in (0, 36):
        mfcc_close(m[i].cpu().numpy(), O.mfcc(audio[i].cpu().numpy(), O.OracleConfig(**kw)), f"clip {i}")


def test_many_mels_w8_kernel(gpu):
    """n_mels = 256: 1-2 bin filters, empty filters, and the 8-wave kernel by default."""
    kw, y, _ = load_golden("c1_am")
    kw = {**kw, "n_mels": 256, "n_mfcc": 20, "fmin": 0.0}
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-w8"
    got = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
    mfcc_close(got, O.mfcc(y, O.OracleConfig(**kw)), "n_mels 256")


def test_clamp_path_is_exercised(gpu):
    kw, y, exp = load_golden("c1_quiet_tail")
    lm = exp["logmel_unclamped"]
    assert lm.min() < lm.max() - 80.0           # the fixture really clamps
    plan = _plan(kw)
    got = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
    mfcc_close(got, exp["mfcc"], "quiet_tail")
    noclamp = _plan(kw, top_db=-1.0).mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
    assert np.abs(noclamp - exp["mfcc"]).max() > 1.0


def test_batch_rows_are_independent_and_strided(gpu):
    import torch
    kw, _, _ = load_golden("c1_am")
    cfg = O.OracleConfig(**kw)
    n = 4000
    clips = np.stack([O.synth_clip(100 + i, n, kw["sr"], k) for i, k in
                      enumerate(["am", "noise", "silence", "quiet_tail", "chirp", "impulse", "am"])])
    plan = _plan(kw)
    for pad in (38, 37):                               # even row pitch: radix-16 path; odd: generic
        big = torch.zeros((clips.shape[0], n + pad), dtype=torch.float32, device=gpu)
        big[:, :n] = _dev(clips, gpu)
        got = plan.mfcc(big[:, :n]).cpu().numpy()      # non-contiguous rows
        for i in range(clips.shape[0]):
            mfcc_close(got[i], O.mfcc(clips[i], cfg), f"clip {i} pad {pad}")
        if pad == 38:
            again = plan.mfcc(_dev(clips[3:4], gpu)).cpu().numpy()
            np.testing.assert_array_equal(again[0], got[3])   # bit-identical at any batch position


@pytest.mark.parametrize("n,L", [(512, 512), (1024, 1001), (2048, 2048), (2048, 1500), (2048, 1), (64, 40),
                                 (4096, 4096), (256, 1), (8192, 6001)])
def test_rfft_rows(n, L, gpu):
    rng = np.random.default_rng(n + L)
    x = rng.standard_normal((37, L)).astype(np.float32)
    kw, _, _ = load_golden("c1_am")
    got = _plan(kw).rfft(_dev(x, gpu), n).cpu().numpy()
    want = O.rfft_rows(x, n)
    assert got.shape == want.shape == (37, n // 2 + 1)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale * np.log2(n)


def test_rfft_2048_many_rows(gpu):
    """n = 2048 runs on the wave-per-row kernel: more rows than one pass of the persistent grid
    (512 workgroups x 8 rows), aligned / odd-pitch / offset views, and the generic kernel beside it."""
    import torch
    rng = np.random.default_rng(2048)
    rows = 4096 + 777
    x = rng.standard_normal((rows, 2048)).astype(np.float32)
    want = O.rfft_rows(x, 2048)
    tol = 2e-6 * np.abs(want).max() * 11
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    xd = _dev(x, gpu)
    got = plan.rfft(xd, 2048).cpu().numpy()
    assert np.abs(got - want).max() <= tol
    big = torch.zeros((rows, 2048 + 3), dtype=torch.float32, device=gpu)
    big[:, 1:2049] = xd
    got_odd = plan.rfft(big[:, 1:2049], 2048).cpu().numpy()      # odd pitch, 4-byte aligned rows
    assert np.abs(got_odd - want).max() <= tol
    plan.force_generic(True)
    try:
        got_gen = plan.rfft(xd, 2048).cpu().numpy()
    finally:
        plan.force_generic(False)
    assert np.abs(got_gen - want).max() <= tol


def test_modspec_matches_oracle(gpu):
    kw, y, exp = load_golden("c1_am")
    plan = _plan(kw)
    m = plan.mfcc(_dev(y, gpu)[None, :])
    ms = plan.modspec(m)[0].cpu().numpy()
    want = exp["modspec"]
    assert ms.shape == want.shape == (13, 65)
    assert np.abs(ms - want).max() <= 1e-4 * np.abs(want).max()
    # 10 s clip: T = 1001 -> n_mod 1024
    y10 = O.synth_clip(11, 160000, 16000, "am")
    m10 = plan.mfcc(_dev(y10, gpu)[None, :])
    ms10 = plan.modspec(m10)[0].cpu().numpy()
    want10 = O.modspec(O.mfcc(y10, O.OracleConfig(**kw)))
    assert ms10.shape == (13, 513)
    assert np.abs(ms10 - want10).max() <= 1e-4 * np.abs(want10).max()


def test_full_size_properties(gpu):
    """BASELINE configs[1] shape (1024 x 10 s): properties that need no oracle at full size."""
    import torch
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    g = torch.Generator(device=gpu).manual_seed(0)
    B, n = 1024, 160000
    t = torch.arange(n, device=gpu, dtype=torch.float32) / 16000.0
    base = 0.3 * torch.sin(2 * np.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 4 * t))
    audio = base[None, :] + 0.05 * torch.randn((B, n), generator=g, device=gpu)
    m = plan.mfcc(audio)
    assert m.shape == (B, 13, 1001) and bool(torch.isfinite(m).all())
    # (1) spot clips agree with the oracle
    for i in (0, 511, 1023):
        mfcc_close(m[i].cpu().numpy(), O.mfcc(audio[i].cpu().numpy(), O.OracleConfig(**kw)), f"clip {i}")
    # (2) permutation equivariance over the batch, bit-exact
    perm = torch.randperm(B, device=gpu, generator=g)
    assert torch.equal(plan.mfcc(audio[perm].contiguous()), m[perm])
    # (3) gain invariance: scaling the input by g shifts c0 by sqrt(n_mels)*20*log10(g), leaves c1.. alone
    m2 = plan.mfcc(audio[:8] * 4.0)
    shift = np.sqrt(kw["n_mels"]) * 20 * np.log10(4.0)
    assert torch.allclose(m2[:, 0], m[:8, 0] + shift, atol=2e-2)
    assert torch.allclose(m2[:, 1:], m[:8, 1:], atol=2e-2)
    # (4) Parseval on the trajectory rFFT
    ms = plan.modspec(m)
    assert ms.shape == (B, 13, 513)
    e_time = (m.double() ** 2).sum(-1)
    w = torch.full((513,), 2.0, device=gpu, dtype=torch.float64)
    w[0] = w[-1] = 1.0
    e_freq = ((ms.real.double() ** 2 + ms.imag.double() ** 2) * w).sum(-1) / 1024
    assert torch.allclose(e_time, e_freq, rtol=1e-4)


@pytest.mark.parametrize("extra,n,batches", [
    (dict(), 160000, (256, 512, 300)),                        # T 1001 -> n_mod 1024; 300 clips: uneven -> separate launches
    (dict(), 48000, (256, 1024, 2560)),                       # T 301 -> n_mod 512; ten clips per workgroup
    (dict(top_db=-1.0), 160000, (256,)),                      # no clamp
    (dict(preemph=0.97, hop_length=161), 48003, (256,)),      # odd hop, unaligned length, pre-emphasis
    (dict(n_mfcc=16), 160000, (256,)),                        # more rows than one pass of the rFFT waves takes
    (dict(n_mfcc=20, n_mels=64), 160000, (256,)),             # no fused DCT for this shape -> separate launches
])
def test_fused_tail_matches_separate_launches(extra, n, batches, gpu):
    """mm_mfcc_modspec_f32: where the plan runs it as ONE launch (whole clips per workgroup, clip extremes kept in
    the workgroup, clamp fix-up and trajectory rFFT inside the tile kernel) the results equal the separate
    launches' -- MFCC bit for bit, including clips that clamp; modulation spectrum to float32 round-off -- spot
    clips against the oracle; the uneven batch takes the separate launches and says so."""
    import torch
    kw, _, _ = load_golden("c1_am")
    kw = dict(kw, **extra)
    plan = _plan(kw)
    g = torch.Generator(device=gpu).manual_seed(3)
    for B in batches:
        audio = 0.1 * torch.randn((B, n), generator=g, device=gpu)
        audio[::3, n // 2:] *= 1e-6                                          # quiet tails: these clips clamp
        audio[1::7] = 0.0
        audio[1::7, 1000] = 1.0                                              # an impulse in silence
        even = B % 256 == 0
        assert plan.fused_tail(B, n) == (even and plan.kernel_path == "radix16-w16s" and plan.fused_dct)
        m1, s1 = plan.mfcc_modspec(audio)
        prev = plan.set_fuse_tail(False)
        try:
            assert not plan.fused_tail(B, n)
            m0, s0 = plan.mfcc_modspec(audio)
        finally:
            plan.set_fuse_tail(prev)
        m2 = plan.mfcc(audio)
        s2 = plan.modspec(m2)
        assert torch.equal(m0, m2) and torch.equal(torch.view_as_real(s0), torch.view_as_real(s2))
        assert torch.equal(m1, m0), float((m1 - m0).abs().max())
        # the in-kernel trajectory rFFT is a second instantiation of the same source: the compiler contracts its
        # multiply-adds differently, so the spectra agree to float32 round-off (1 ulp of the row maximum), not bitwise
        err = (torch.view_as_real(s1) - torch.view_as_real(s0)).abs().amax(dim=(2, 3))
        scale = torch.view_as_real(s0).abs().amax(dim=(2, 3))
        assert bool((err <= 4e-7 * scale + 1e-30).all()), float((err / (scale + 1e-30)).max())
        ocfg = O.OracleConfig(**dict(kw, top_db=None if kw.get("top_db", 80.0) < 0 else kw.get("top_db", 80.0)))
        for i in (0, 1, B - 1):
            mfcc_close(m1[i].cpu().numpy(), O.mfcc(audio[i].cpu().numpy(), ocfg), f"B={B} clip {i}")
    if kw.get("top_db", 80.0) >= 0:
        # the clamp did bite in the quiet-tail clips (otherwise this test would not cover the fix-up)
        lm, mx = plan.logmel(audio[:1])
        assert float(lm.min()) < float(mx[0]) - 80.0


_ANY_NFFT = [
    # (cfg, what)
    (dict(sr=16000, n_fft=400, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "400 = 2^4 5^2 (window == n_fft)"),
    (dict(sr=10000, n_fft=400, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), "reference defaults, n_fft 400"),
    (dict(sr=10000, n_fft=1000, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), "reference defaults, n_fft 1000"),
    (dict(sr=32000, n_fft=800, win_length=800, hop_length=320, n_mels=64, n_mfcc=20, fmin=50.0, fmax=16000.0), "800 = 16 x 25 complex points: two register stages"),
    (dict(sr=16000, n_fft=800, win_length=400, hop_length=37, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0, preemph=0.97), "800, odd hop, pre-emphasis"),
    (dict(sr=16000, n_fft=400, win_length=400, hop_length=99, n_mels=26, n_mfcc=13, fmin=0.0, fmax=8000.0, preemph=0.95, top_db=30.0), "400, odd hop, pre-emphasis"),
    (dict(sr=8000, n_fft=200, win_length=200, hop_length=80, n_mels=24, n_mfcc=12, fmin=50.0, fmax=4000.0), "200 = 2 x (4 x 25): two register stages"),
    (dict(sr=8000, n_fft=240, win_length=240, hop_length=80, n_mels=24, n_mfcc=12, fmin=50.0, fmax=4000.0), "240 = 2 x (8 x 15)"),
    (dict(sr=16000, n_fft=320, win_length=320, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "320 = 2 x (8 x 20)"),
    (dict(sr=16000, n_fft=480, win_length=480, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "480 = 2 x (16 x 15)"),
    (dict(sr=16000, n_fft=640, win_length=640, hop_length=161, n_mels=64, n_mfcc=13, fmin=100.0, fmax=8000.0, preemph=0.97), "640 = 2 x (16 x 20), odd hop, pre-emphasis"),
    (dict(sr=48000, n_fft=960, win_length=960, hop_length=480, n_mels=80, n_mfcc=20, fmin=100.0, fmax=20000.0), "960 = 2 x (24 x 20)"),
    (dict(sr=48000, n_fft=1200, win_length=1200, hop_length=480, n_mels=80, n_mfcc=20, fmin=100.0, fmax=20000.0), "1200 = 2 x (24 x 25)"),
    (dict(sr=44100, n_fft=1600, win_length=1102, hop_length=441, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), "1600 = 2 x (32 x 25)"),
    (dict(sr=48000, n_fft=2000, win_length=1200, hop_length=480, n_mels=80, n_mfcc=20, fmin=100.0, fmax=20000.0, preemph=0.97), "2000 = 2 x (40 x 25), pre-emphasis"),
    (dict(sr=16000, n_fft=600, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "600 = 2^3 3 5^2"),
    (dict(sr=22050, n_fft=1536, win_length=551, hop_length=220, n_mels=64, n_mfcc=20, fmin=0.0, fmax=11025.0), "1536 = 2^9 3"),
    (dict(sr=44100, n_fft=441, win_length=441, hop_length=147, n_mels=30, n_mfcc=12, fmin=50.0, fmax=20000.0), "441 = 3^2 7^2, odd"),
    (dict(sr=8000, n_fft=875, win_length=800, hop_length=100, n_mels=40, n_mfcc=13, fmin=20.0, fmax=4000.0, preemph=0.97), "875 = 5^3 7, odd, pre-emphasis"),
    (dict(sr=16000, n_fft=502, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "502 = 2 x 251: Bluestein on 251 points"),
    (dict(sr=16000, n_fft=499, win_length=400, hop_length=161, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "499 prime: Bluestein, odd hop"),
    (dict(sr=16000, n_fft=8192, win_length=4000, hop_length=1000, n_mels=128, n_mfcc=40, fmin=0.0, fmax=8000.0), "8192: workgroup per frame"),
    (dict(sr=48000, n_fft=6000, win_length=4800, hop_length=960, n_mels=80, n_mfcc=20, fmin=100.0, fmax=20000.0), "6000 = 2^4 3 5^3: workgroup per frame"),
    (dict(sr=16000, n_fft=8190, win_length=800, hop_length=400, n_mels=64, n_mfcc=13, fmin=100.0, fmax=8000.0), "8190 = 2 x 4095 (13 | 4095): Bluestein, M 8192"),
    (dict(sr=48000, n_fft=4099, win_length=2400, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0, top_db=40.0), "4099 prime, odd: Bluestein, M 16384"),
    (dict(sr=8000, n_fft=16, win_length=16, hop_length=4, n_mels=4, n_mfcc=3, fmin=0.0, fmax=4000.0), "16: below the radix kernels' range"),
    (dict(sr=8000, n_fft=6, win_length=5, hop_length=2, n_mels=2, n_mfcc=2, fmin=0.0, fmax=4000.0), "6"),
    (dict(sr=8000, n_fft=3, win_length=3, hop_length=1, n_mels=2, n_mfcc=1, fmin=0.0, fmax=4000.0), "3"),
    (dict(sr=8000, n_fft=2, win_length=2, hop_length=1, n_mels=1, n_mfcc=1, fmin=0.0, fmax=4000.0), "2"),
]


@pytest.mark.parametrize("kw,what", _ANY_NFFT, ids=[f"nfft{c[0]['n_fft']}_{i}" for i, c in enumerate(_ANY_NFFT)])
def test_any_integer_n_fft_matches_oracle(kw, what, gpu):
    """librosa.feature.mfcc takes any n_fft >= win_length and the reference's dialog passes what the user types
    (script/config_dialog.py:141,610 -> script/main.py:1049-1066 -> script/mfcc.py:387): every length that is not a
    power of two in [32, 4096] runs on stft_any_kernel -- mixed-radix Stockham FFT in LDS for 2/3/5/7-smooth lengths
    (even: packed into n/2 complex points; odd: n complex points), Bluestein's chirp-z transform otherwise, one wave or
    one workgroup per frame -- against the oracle: MFCC on am / quiet-tail (the clamp bites) / noise clips of ragged
    lengths, the power and log-mel stage outputs, and MFCC + modulation spectrum in one call."""
    import warnings
    plan = _plan(kw)
    assert plan.kernel_path == "any-length" and not plan.fused_dct
    n_fft, hop = kw["n_fft"], kw["hop_length"]
    okw = dict(kw)
    okw["top_db"] = kw.get("top_db", 80.0)
    ocfg = O.OracleConfig(**okw)
    rng = np.random.default_rng(n_fft)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # empty mel filters of the tiny lengths
        for n in (int(rng.integers(max(2, n_fft // 2), max(3, n_fft))), 3 * n_fft + 5 * hop + 1, 20 * hop + n_fft):
            clips = np.stack([O.synth_clip(7 + n_fft + i, n, kw["sr"], k) for i, k in enumerate(("am", "quiet_tail", "noise"))])
            got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
            for i in range(3):
                want = O.mfcc(clips[i], ocfg)
                assert got[i].shape == want.shape
                mfcc_close(got[i], want, f"n_fft {n_fft} ({what}) n {n} clip {i}")
        y = clips[0]
        P = plan.stft_power(_dev(y, gpu)[None, :])[0].cpu().numpy()
        Pw = O.stft_power(y, n_fft, hop, kw["win_length"], kw.get("preemph", 0.0))
        assert P.shape == Pw.shape == (1 + (n - n_fft % 2) // hop, n_fft // 2 + 1)
        # float32 transform vs the float64 one rounded once: noise relative to the frame's largest bin
        assert np.abs(P - Pw).max() <= 2e-5 * Pw.max(), np.abs(P - Pw).max() / Pw.max()
        lm, mx = plan.logmel(_dev(y, gpu)[None, :])
        lw = O.logmel_unclamped(y, ocfg).T
        assert np.abs(lm[0].cpu().numpy() - lw).max() <= 2e-2 and abs(float(mx[0]) - lw.max()) <= 1e-3
        m2, s2 = plan.mfcc_modspec(_dev(clips, gpu))
        assert not plan.fused_tail(3, n)
        np.testing.assert_array_equal(m2.cpu().numpy(), got)
        wm = O.modspec(got[0])
        assert np.abs(s2[0].cpu().numpy() - wm).max() <= 1e-4 * max(np.abs(wm).max(), 1e-30)


@pytest.mark.parametrize("idx", range(16))
def test_random_any_length_configs_match_oracle(idx, gpu):
    """Random n_fft in [8, 3000] that are NOT powers of two (smooth and Bluestein lengths as they come), random window /
    hop / mel bank / clamp / pre-emphasis, two ragged clips each, against the oracle."""
    import warnings
    rng = np.random.default_rng(31000 + idx)
    while True:
        n_fft = int(rng.integers(8, 3001))
        if n_fft & (n_fft - 1):
            break
    win = int(rng.integers(max(2, n_fft // 4), n_fft + 1))
    hop = int(rng.integers(1, max(2, win)))
    sr = int(rng.choice([8000, 10000, 16000, 22050, 44100, 48000]))
    n_mels = int(rng.integers(2, max(3, min(129, n_fft // 2))))
    n_mfcc = int(rng.integers(1, min(n_mels, 40) + 1))
    kw = dict(sr=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=n_mels, n_mfcc=n_mfcc,
              fmin=float(rng.choice([0.0, 20.0, 100.0])), fmax=float(rng.choice([sr / 2, sr * 0.45, sr * 0.7])),
              top_db=float(rng.choice([80.0, 40.0, -1.0])), preemph=float(rng.choice([0.0, 0.0, 0.97])))
    plan = _plan(kw)
    assert plan.kernel_path == "any-length"
    okw = dict(kw, top_db=None if kw["top_db"] < 0 else kw["top_db"])
    n = int(rng.integers(n_fft // 2, 6 * n_fft + 7 * hop))
    clips = np.stack([O.synth_clip(900 + idx, n, sr, k) for k in ("am", "quiet_tail")])
    got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(2):
            want = O.mfcc(clips[i], O.OracleConfig(**okw))
            assert got[i].shape == want.shape
            mfcc_close(got[i], want, f"any-length cfg {idx} {kw} clip {i}")


def test_drop_in_with_a_typed_n_fft(gpu):
    """get_MFCCS_change as the config dialog calls it (script/main.py:1049-1066) with n_fft values a user types --
    400, 1000 -- at the UI's 10 kHz defaults: no NotImplementedError, results equal the oracle's."""
    from modulation_mfcc_amd import get_MFCCS_change
    kw, y, _ = load_golden("refdefault_am")
    for n_fft in (400, 1000, 600):
        tot, T = get_MFCCS_change(y, 10000, tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=n_fft, removeFirst=1, filtCutoff=12,
                                  filtOrd=6, diffMethod="grad", outFilter="iir", outFiltCutOff=[12])
        wt, wT = O.get_MFCCS_change(y, 10000, tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=n_fft, removeFirst=1, filtCutoff=12,
                                    filtOrd=6, diffMethod="grad", outFilter="iir", outFiltCutOff=[12])
        np.testing.assert_array_equal(T, wT)
        assert np.abs(tot - wt).max() <= 1e-4 * np.abs(wt).max()


def test_headline_config_through_the_headline_entry_point(gpu):
    """BASELINE configs[2] EXACTLY as bench.py times it: 1024 clips x 160 000 samples through mm_mfcc_modspec_f32
    in clip mode (ONE launch, 4 clips per workgroup, T 1001, n_mod 1024) -- the bench's signal model
    (bench.synth_batch) with quiet-tail clips (they clamp: the in-launch fix-up runs) and impulse-in-silence clips
    mixed in.  MFCC bit-equal to the separate launches, spectra within 4e-7 of the row maximum, spot clips (one of
    every kind) against the oracle for MFCC and modulation spectrum, Parseval on every trajectory."""
    import math
    import torch
    kw, _, _ = load_golden("c1_am")
    kw = dict(kw, fmax=8000.0)                            # bench.py C16K
    plan = _plan(kw)
    B, n, sr = 1024, 160000, 16000
    g = torch.Generator(device=gpu).manual_seed(1000)
    t = torch.arange(n, device=gpu, dtype=torch.float64) / sr
    base = (0.3 * torch.sin(2 * math.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * math.pi * 4 * t))).float()
    audio = torch.randn((B, n), generator=g, device=gpu, dtype=torch.float32)
    audio.mul_(0.05).add_(base[None, :])
    audio[5::64, n // 2:] *= 1e-6                         # quiet tails: 16 clips whose minimum lies > 80 dB under the maximum
    audio[9::128] = 0.0
    audio[9::128, 70001] = 1.0                            # impulse in digital silence: 8 clips
    assert plan.kernel_path == "radix16-w16s" and plan.fused_dct and plan.fused_tail(B, n)
    m1, s1 = plan.mfcc_modspec(audio)
    assert m1.shape == (B, 13, 1001) and s1.shape == (B, 13, 513)
    assert bool(torch.isfinite(m1).all()) and bool(torch.isfinite(torch.view_as_real(s1)).all())
    prev = plan.set_fuse_tail(False)
    try:
        assert not plan.fused_tail(B, n)
        m0, s0 = plan.mfcc_modspec(audio)
    finally:
        plan.set_fuse_tail(prev)
    assert torch.equal(m1, m0), float((m1 - m0).abs().max())
    err = (torch.view_as_real(s1) - torch.view_as_real(s0)).abs().amax(dim=(2, 3))
    scale = torch.view_as_real(s0).abs().amax(dim=(2, 3))
    assert bool((err <= 4e-7 * scale + 1e-30).all()), float((err / (scale + 1e-30)).max())
    # the clamp did bite in the quiet-tail clips and in the impulse clips
    lm, mx = plan.logmel(audio[5:10])
    assert float(lm[0].min()) < float(mx[0]) - 80.0 and float(lm[4].min()) < float(mx[4]) - 80.0
    ocfg = O.OracleConfig(**kw)
    for i in (0, 5, 9, 517, 1023):                        # plain, quiet tail, impulse, quiet tail (5 + 64 * 8), last
        a = audio[i].cpu().numpy()
        want = O.mfcc(a, ocfg)
        mfcc_close(m1[i].cpu().numpy(), want, f"headline clip {i}")
        wm = O.modspec(want)
        assert np.abs(s1[i].cpu().numpy() - wm).max() <= 1e-4 * np.abs(wm).max(), f"modspec clip {i}"
    e_time = (m1.double() ** 2).sum(-1)
    w = torch.full((513,), 2.0, device=gpu, dtype=torch.float64)
    w[0] = w[-1] = 1.0
    e_freq = ((s1.real.double() ** 2 + s1.imag.double() ** 2) * w).sum(-1) / 1024
    assert torch.allclose(e_time, e_freq, rtol=1e-4)


@pytest.mark.parametrize("n,B", [(100000, 512), (64321, 256), (51300, 300)])
def test_clip_mode_with_a_2048_point_trajectory(n, B, gpu):
    """The reference's own default call (10 kHz, win 250, hop 50, 128 mel: script/main.py:732-748) as a batch: ten seconds
    are 2001 frames, so the trajectory rFFT is 2048 points -- clip mode (opt-in for this length: set_fuse_tail(2)) runs it
    inside the launch on the 2048-point register transform (s16_fin_modspec_2k).  MFCC bit-equal to the separate launches and the spectrum bit-equal to the
    separate rfft_wpf_kernel<4> (the same arithmetic on the same rows); spot clips of every kind against the oracle;
    Parseval on every trajectory.  Ragged lengths (T 1287, 1027: just above 1024) as well; a batch that does not spread
    evenly over the workgroups (300) takes the separate launches and agrees."""
    import torch
    kw = dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0)
    plan = _plan(kw)
    T = 1 + n // 50
    assert plan.cfg.mod_fft_len(T) == 2048
    rng = np.random.default_rng(n)
    kinds = ("am", "quiet_tail", "noise", "silence", "impulse")
    host = np.stack([O.synth_clip(100 + i, n, 10000, kinds[i % 5]) if i < 10 else
                     (0.05 * rng.standard_normal(n)).astype(np.float32) for i in range(B)])
    audio = _dev(host, gpu)
    assert not plan.fused_tail(B, n)                      # opt-in: set_fuse_tail(2)
    m0, s0 = plan.mfcc_modspec(audio)
    prev = plan.set_fuse_tail(2)
    try:
        assert plan.fused_tail(B, n) == (B != 300)
        m1, s1 = plan.mfcc_modspec(audio)
        mc = plan.mfcc(audio)                             # clip mode without a spectrum (the empty filters' add in the launch)
    finally:
        plan.set_fuse_tail(prev)
    assert m1.shape == (B, 13, T) and s1.shape == (B, 13, 1025)
    assert torch.equal(mc, m0)
    assert torch.equal(m1, m0), float((m1 - m0).abs().max())
    assert torch.equal(torch.view_as_real(s1), torch.view_as_real(s0))
    ocfg = O.OracleConfig(**kw)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # "Empty filters detected"
        for i in (0, 1, 3, 4, B - 1):
            want = O.mfcc(host[i], ocfg)
            mfcc_close(m1[i].cpu().numpy(), want, f"clip {i}")
            wm = O.modspec(want)
            assert np.abs(s1[i].cpu().numpy() - wm).max() <= 1e-4 * max(np.abs(wm).max(), 1e-30), f"modspec clip {i}"
    e_time = (m1.double() ** 2).sum(-1)
    w = torch.full((1025,), 2.0, device=gpu, dtype=torch.float64)
    w[0] = w[-1] = 1.0
    e_freq = ((s1.real.double() ** 2 + s1.imag.double() ** 2) * w).sum(-1) / 2048
    assert torch.allclose(e_time, e_freq, rtol=1e-4)


_HALFBAND_CFGS = [
    # (cfg, what): n_fft 1024 / 2048 plans whose mel bank ends below sr / 4 -- the output-pruned wave-per-frame instantiations
    (dict(sr=48000, n_fft=2048, win_length=1200, hop_length=480, n_mels=80, n_mfcc=40, fmin=100.0, fmax=10000.0), "BASELINE configs[3]: Z 3, 7 pairs"),
    (dict(sr=44100, n_fft=2048, win_length=2048, hop_length=512, n_mels=64, n_mfcc=20, fmin=0.0, fmax=8000.0), "full window, 6 pairs"),
    (dict(sr=48000, n_fft=2048, win_length=1024, hop_length=256, n_mels=40, n_mfcc=13, fmin=50.0, fmax=3000.0), "Z 3, 4 pairs (bank ends at bin 128)"),
    (dict(sr=44100, n_fft=2048, win_length=1280, hop_length=441, n_mels=96, n_mfcc=13, fmin=100.0, fmax=9000.0), "96 mel; the window's edge case for Z 3"),
    (dict(sr=44100, n_fft=1024, win_length=1024, hop_length=256, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "n_fft 1024, 6 pairs"),
    (dict(sr=48000, n_fft=1024, win_length=600, hop_length=240, n_mels=32, n_mfcc=20, fmin=100.0, fmax=5000.0), "n_fft 1024, Z 3, 4 pairs"),
    (dict(sr=22050, n_fft=1024, win_length=640, hop_length=221, n_mels=48, n_mfcc=13, fmin=0.0, fmax=5400.0, top_db=40.0), "n_fft 1024, odd hop, 8 pairs' edge (k_hi 251)"),
    # zero-padded frames: the reference's dialog lets the user type n_fft while winLen stays 25 ms (script/main.py:1049-1066)
    (dict(sr=10000, n_fft=1024, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), "UI defaults, n_fft 1024 typed: Z 6"),
    (dict(sr=10000, n_fft=2048, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), "UI defaults, n_fft 2048 typed: Z 7"),
    (dict(sr=16000, n_fft=2048, win_length=640, hop_length=160, n_mels=80, n_mfcc=13, fmin=100.0, fmax=8000.0), "Z 5, full band"),
    (dict(sr=16000, n_fft=2048, win_length=2048, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "few wide filters: the last run spans nine lanes (used to fall to the generic kernel)"),
    (dict(sr=16000, n_fft=1024, win_length=321, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0), "n_fft 1024, odd window, Z 5"),
    (dict(sr=48000, n_fft=2048, win_length=500, hop_length=240, n_mels=64, n_mfcc=20, fmin=100.0, fmax=10000.0), "Z 6 window with a half-band bank (the Z 3 half-band instantiation)"),
]


@pytest.mark.parametrize("kw,what", _HALFBAND_CFGS, ids=[f"halfband{i}" for i in range(len(_HALFBAND_CFGS))])
def test_half_band_wave_per_frame_kernels(kw, what, gpu):
    """A window that leaves a lane's first / last stage-1 pairs zero and a mel bank that ends below sr / 4 select the pruned
    instantiations of logmel_wpf_kernel (Z = 3; NI = 4 .. 7 with eight-bin sweep slices, up to sixteen waves per CU, stage 3
    through DPP): MFCC, log-mel rows and the power stage (which always runs the full kernel) against the oracle on clips of
    every kind and ragged lengths, and MFCC + modulation spectrum in one call."""
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-wpf"
    ocfg = O.OracleConfig(**dict(kw, top_db=kw.get("top_db", 80.0)))
    hop, n_fft = kw["hop_length"], kw["n_fft"]
    for n in (n_fft // 2 + 3, 7 * hop + n_fft + 1, 40 * hop + 5 * n_fft):
        clips = np.stack([O.synth_clip(500 + i, n, kw["sr"], k) for i, k in enumerate(("am", "quiet_tail", "noise", "impulse", "silence"))])
        got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
        for i in range(len(clips)):
            want = O.mfcc(clips[i], ocfg)
            assert got[i].shape == want.shape
            mfcc_close(got[i], want, f"{what}: n {n} clip {i}")
    y = clips[0]
    lm, mx = plan.logmel(_dev(y, gpu)[None, :])
    lw = O.logmel_unclamped(y, ocfg).T
    assert np.abs(lm[0].cpu().numpy() - lw).max() <= 2e-2 and abs(float(mx[0]) - lw.max()) <= 1e-3
    P = plan.stft_power(_dev(y, gpu)[None, :])[0].cpu().numpy()
    Pw = O.stft_power(y, n_fft, hop, kw["win_length"], 0.0)
    assert np.abs(P - Pw).max() <= 2e-5 * Pw.max()
    m2, s2 = plan.mfcc_modspec(_dev(clips, gpu))
    np.testing.assert_array_equal(m2.cpu().numpy(), got)
    wm = O.modspec(got[0])
    assert np.abs(s2[0].cpu().numpy() - wm).max() <= 1e-4 * max(np.abs(wm).max(), 1e-30)


def test_modulation_spectrum_of_long_trajectories(gpu):
    """Row A8 beyond 8192 frames per clip -- what the reference's own default step makes of a recording (tStep = 0.001,
    script/mfcc.py:296: 10 001 frames per ten seconds; a one-minute file: 60 001): mm_modspec_f32 stops at 8192 points,
    MfccPlan.modspec / mfcc_modspec hand longer trajectories to the Stockham FFT in global memory (mm_hilbert_rfft_f32:
    16 384 = 256 x 64 and 65 536 = 256 x 256 points).  Against np.fft.rfft in float64, Parseval, and the short path on
    the same rows (a trajectory cut to 8000 frames and transformed at n_mod_fft = 16384 through both)."""
    import torch
    from modulation_mfcc_amd import calc
    kw = dict(sr=10000, n_fft=512, win_length=250, hop_length=10, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0)
    plan = _plan(kw)
    for secs, B in ((10.0, 3), (60.0, 1)):
        n = int(secs * 10000)
        T = 1 + n // 10
        nm = plan.cfg.mod_fft_len(T)
        assert nm == (16384 if secs == 10.0 else 65536)
        host = np.stack([O.synth_clip(40 + i, n, 10000, ("am", "noise", "quiet_tail")[i % 3]) for i in range(B)])
        m, s = plan.mfcc_modspec(_dev(host, gpu))
        assert m.shape == (B, 13, T) and s.shape == (B, 13, nm // 2 + 1)
        assert torch.equal(m, plan.mfcc(_dev(host, gpu)))
        mh = m.cpu().numpy()
        want = np.fft.rfft(mh.astype(np.float64), n=nm, axis=-1)
        got = s.cpu().numpy()
        scale = np.abs(want).max(axis=-1, keepdims=True)
        assert (np.abs(got - want) <= 2e-6 * scale).all(), float((np.abs(got - want) / scale).max())
        e_time = (mh.astype(np.float64) ** 2).sum(-1)
        w = np.full(nm // 2 + 1, 2.0); w[0] = w[-1] = 1.0
        e_freq = ((np.abs(got.astype(np.complex128)) ** 2) * w).sum(-1) / nm
        np.testing.assert_allclose(e_freq, e_time, rtol=1e-4)
    # the two transforms agree where both apply: 8000 frames at 8192 points (mm_modspec_f32) and at 16384 (long path)
    rows = torch.randn((5, 8000), device=gpu)
    long16 = calc.rfft_rows_long(rows, 16384).cpu().numpy()
    w16 = np.fft.rfft(rows.cpu().numpy().astype(np.float64), n=16384, axis=-1)
    assert np.abs(long16 - w16).max() <= 2e-6 * np.abs(w16).max()
    # ragged T, many rows (chunked workspace), an output buffer of the caller
    rows = torch.randn((300, 9001), device=gpu)
    out = torch.empty((300, 8193), dtype=torch.complex64, device=gpu)
    calc.rfft_rows_long(rows, 16384, out=out)
    w = np.fft.rfft(rows.cpu().numpy().astype(np.float64), n=16384, axis=-1)
    assert np.abs(out.cpu().numpy() - w).max() <= 2e-6 * np.abs(w).max()


_EMPTY_CFGS = [
    # (cfg, n_samples, what)
    (dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0), 50000,
     "the reference's own default call: 26 of 128 filters above Nyquist; one log-mel tile in LDS (DCT at the end of phase A)"),
    (dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=10000.0), 160000,
     "BASELINE shape with the reference's maxFreq: filter 39 empty; two log-mel tiles"),
    (dict(sr=44100, n_fft=512, win_length=441, hop_length=110, n_mels=96, n_mfcc=20, fmin=60.0, fmax=10000.0), 88200,
     "44.1 kHz: empty filters scattered through the low bands (narrower than a bin); n_mfcc 20: two coefficient blocks"),
    (dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0, top_db=-1.0), 50000,
     "no clamp: the empty filters' share is E[k] L0 alone"),
]


@pytest.mark.parametrize("kw,n,what", _EMPTY_CFGS, ids=[f"empty{i}" for i in range(len(_EMPTY_CFGS))])
def test_empty_filters_are_handled_analytically(kw, n, what, gpu):
    """Filters without a single weight always read 10 log10(amin) = L0 (librosa warns 'Empty filters detected'; the
    reference's default maxFreq = 10000 at 10 kHz gives 26 of 128).  The fused kernel neither computes nor stores them,
    keeps them out of the clip minimum (else EVERY clip of such a plan would take the clamp fix-up) and adds their share
    of the DCT analytically: E[k] L0 in the accumulators + E[k] max(0, thr - L0) once the clip's threshold is known.
    Against the oracle and against the separate clamp + DCT kernel over all rows, for every kind of clip: loud (thr above
    L0: the add), 60 dB down (thr below L0: nothing), quiet tail (live filters clamp: the fix-up's correction on top of the
    stored DCT), digital silence (max = L0), an impulse; tile mode (mm_mfcc_f32) and clip mode
    (mm_mfcc_modspec_f32, one launch) bit-equal."""
    import torch
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-w16s" and plan.fused_dct
    B = 256
    g = torch.Generator(device=gpu).manual_seed(11)
    sr = kw["sr"]
    t = torch.arange(n, device=gpu, dtype=torch.float64) / sr
    base = (0.3 * torch.sin(2 * np.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 4 * t))).float()
    audio = 0.05 * torch.randn((B, n), generator=g, device=gpu) + base[None, :]
    audio[1::8] *= 1e-3                        # 60 dB down: thr < L0, no correction
    audio[2::8, n // 2:] *= 1e-6               # quiet tail: live filters clamp
    audio[3::8] = 0.0                          # digital silence
    audio[4::8] = 0.0
    audio[4::8, n // 3] = 1.0                  # an impulse in silence
    okw = dict(kw)
    okw["top_db"] = None if kw.get("top_db", 80.0) < 0 else kw.get("top_db", 80.0)
    ocfg = O.OracleConfig(**okw)
    m = plan.mfcc(audio)
    assert bool(torch.isfinite(m).all())
    for i in (0, 1, 2, 3, 4, B - 1):
        mfcc_close(m[i].cpu().numpy(), O.mfcc(audio[i].cpu().numpy(), ocfg), f"{what}: clip {i}")
    # opt-in (set_fuse_tail(2)): mm_mfcc_f32 in clip mode -- the add and the fix-up inside the launch -- gives the bits of
    # the tile kernel + dct_fixup_kernel
    prev = plan.set_fuse_tail(2)
    try:
        mt = plan.mfcc(audio)
    finally:
        plan.set_fuse_tail(prev)
    assert torch.equal(m, mt), float((m - mt).abs().max())
    assert torch.equal(plan.mfcc(audio[:77]), m[:77])
    prev = plan.set_fuse_dct(False)
    try:
        assert not plan.fused_dct
        m0 = plan.mfcc(audio)
    finally:
        plan.set_fuse_dct(prev)
    scale = m0.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
    # float32 round-off of a differently ordered sum (the clips that clamp too: the fix-up adds the correction of the values
    # under the threshold to the stored unclamped DCT, mm_clamp_corr)
    assert float(((m - m0).abs() / scale).max()) <= 4e-6
    T = m.shape[2]
    if plan.fused_tail(B, n):
        m2, s2 = plan.mfcc_modspec(audio)
        assert torch.equal(m2, m), float((m2 - m).abs().max())
        wm = O.modspec(O.mfcc(audio[0].cpu().numpy(), ocfg))
        assert np.abs(s2[0].cpu().numpy() - wm).max() <= 1e-4 * np.abs(wm).max()
    else:
        assert plan.cfg.mod_fft_len(T) > 1024


def test_configs4_workload_on_one_gpu(gpu):
    """BASELINE configs[4]'s WORKLOAD -- 8192 clips x 10 s x 16 kHz, MFCC + modulation spectrum -- on one GPU (5.2 GB of
    audio; the 8-GPU run gives each rank one eighth of exactly this batch).  (a) the whole batch through
    mm_mfcc_modspec_f32: 32 clips per workgroup = MM_S16_CPW_MAX, the edge of clip mode, still ONE launch; finite,
    Parseval on every trajectory, five spot clips (plain / quiet tail / impulse / first / last) against the oracle for
    MFCC and spectrum.  (b) the same batch cut with dist.shard_bounds(8192, 8) -- the N = 8 partitioning -- each
    1024-clip shard run separately: the concatenation is BIT-EQUAL to (a), MFCC and spectrum (what sharding must
    guarantee: a clip's result does not depend on which rank or workgroup computed it).  (c) 8448 clips = 33 per
    workgroup: past the clip-mode limit, the separate launches run (fused_tail says so) and the first 8192 clips still
    equal (a) -- MFCC bit for bit, spectra to float32 round-off."""
    import math
    import torch
    from modulation_mfcc_amd.dist import shard_bounds
    kw, _, _ = load_golden("c1_am")
    kw = dict(kw, fmax=8000.0)                            # bench.py C16K
    plan = _plan(kw)
    B, Bx, n, sr = 8192, 8448, 160000, 16000
    g = torch.Generator(device=gpu).manual_seed(4000)
    t = torch.arange(n, device=gpu, dtype=torch.float64) / sr
    base = (0.3 * torch.sin(2 * math.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * math.pi * 4 * t))).float()
    big = torch.randn((Bx, n), generator=g, device=gpu, dtype=torch.float32)
    big.mul_(0.05).add_(base[None, :])
    big[5::64, n // 2:] *= 1e-6                           # quiet tails: these clips clamp (in-launch fix-up)
    big[9::128] = 0.0
    big[9::128, 70001] = 1.0                              # impulse in digital silence
    audio = big[:B]
    assert plan.kernel_path == "radix16-w16s" and plan.fused_dct
    assert plan.fused_tail(B, n)                          # 32 clips per workgroup: the limit, still one launch
    assert not plan.fused_tail(Bx, n)                     # 33: separate launches
    m1, s1 = plan.mfcc_modspec(audio)
    assert m1.shape == (B, 13, 1001) and s1.shape == (B, 13, 513)
    assert bool(torch.isfinite(m1).all()) and bool(torch.isfinite(torch.view_as_real(s1)).all())
    ocfg = O.OracleConfig(**kw)
    for i in (0, 4100, 9 + 128 * 40, 5 + 64 * 100, B - 1):      # first, plain, impulse, quiet tail, last
        want = O.mfcc(audio[i].cpu().numpy(), ocfg)
        mfcc_close(m1[i].cpu().numpy(), want, f"configs[4] clip {i}")
        wm = O.modspec(want)
        assert np.abs(s1[i].cpu().numpy() - wm).max() <= 1e-4 * np.abs(wm).max(), f"modspec clip {i}"
    e_time = (m1.double() ** 2).sum(-1)
    w = torch.full((513,), 2.0, device=gpu, dtype=torch.float64)
    w[0] = w[-1] = 1.0
    e_freq = ((s1.real.double() ** 2 + s1.imag.double() ** 2) * w).sum(-1) / 1024
    assert torch.allclose(e_time, e_freq, rtol=1e-4)
    del e_time, e_freq
    # (b) the N = 8 shards, one at a time
    bounds = shard_bounds(B, 8)
    assert [hi - lo for lo, hi in bounds] == [1024] * 8
    for r, (lo, hi) in enumerate(bounds):
        assert plan.fused_tail(hi - lo, n)
        ms, ss = plan.mfcc_modspec(audio[lo:hi])
        assert torch.equal(ms, m1[lo:hi]), f"shard {r}: MFCC differs from the whole-batch run"
        assert torch.equal(torch.view_as_real(ss), torch.view_as_real(s1[lo:hi])), f"shard {r}: spectrum differs"
    del ms, ss
    # (c) one clip per workgroup more than clip mode takes
    mx, sx = plan.mfcc_modspec(big)
    assert torch.equal(mx[:B], m1)
    err = (torch.view_as_real(sx[:B]) - torch.view_as_real(s1)).abs().amax(dim=(2, 3))
    scale = torch.view_as_real(s1).abs().amax(dim=(2, 3))
    assert bool((err <= 4e-7 * scale + 1e-30).all()), float((err / (scale + 1e-30)).max())
    want = O.mfcc(big[Bx - 1].cpu().numpy(), ocfg)
    mfcc_close(mx[Bx - 1].cpu().numpy(), want, "clip 8447 of the 33-per-workgroup batch")
    wm = O.modspec(want)
    assert np.abs(sx[Bx - 1].cpu().numpy() - wm).max() <= 1e-4 * np.abs(wm).max()


# tools/fuzz.py, seed 7 (gpurun_out/fuzz_r02.log): the two configurations of 300 in which one clip fails the suite's
# elementwise bound |a - b| <= 1e-4 |b| + 1e-3 on every kernel path while staying within 6e-6 of max|MFCC|.
_FUZZ_PINS = [
    # (fuzz idx, cfg, clip seed, n, kind)
    (128, dict(sr=48000, n_fft=512, win_length=280, hop_length=202, n_mels=56, n_mfcc=25, fmin=0.0, fmax=3000.0,
               top_db=-1.0, preemph=0.97), 8280, 17424, "am"),
    (270, dict(sr=48000, n_fft=1024, win_length=344, hop_length=188, n_mels=122, n_mfcc=30, fmin=300.0, fmax=21600.0,
               top_db=-1.0, preemph=0.97), 9702, 17604, "quiet_tail"),
]


@pytest.mark.parametrize("pin", _FUZZ_PINS, ids=lambda p: f"idx{p[0]}")
def test_float32_fft_noise_floor_pins(pin, gpu):
    """The float32-FFT noise floor, pinned.  With pre-emphasis 0.97 at 48 kHz the frame's energy
    sits at the Nyquist end, 40+ dB above the bins the low mel filters collect; a float32 transform carries rounding
    noise relative to the frame's LARGEST bin, librosa under the reference's numpy<2 pin transforms in float64
    (/root/reference requirements.txt:2) and rounds once.  A mel filter 70+ dB under the clip maximum then differs by
    a few 1e-3 dB -- inside the reference's own 80 dB clamp range, so it is reachable -- and one MFCC element of the
    clip exceeds the suite's elementwise bound by < 1e-2 while the north-star bound (1e-4 of max|MFCC|) holds with
    a factor 15 to spare.  Same on every kernel path: asserted for each."""
    import warnings
    idx, kw, seed, n, kind = pin
    y = O.synth_clip(seed, n, kw["sr"], kind)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = O.mfcc(y, O.OracleConfig(**dict(kw, top_db=None)))
        lm = O.logmel_unclamped(y, O.OracleConfig(**dict(kw, top_db=None)))
    assert float(lm.max() - lm.min()) > 60.0              # filters far below the clip maximum exist in this clip
    plan = _plan(kw)
    seen = set()
    for variant in VARIANTS:
        for fuse in (True, False):
            with _variant(plan, variant):
                prev = plan.set_fuse_dct(fuse)
                try:
                    path = plan.kernel_path + ("+dct" if plan.fused_dct else "")
                    if variant != "generic" and path in seen:
                        continue
                    seen.add(path)
                    got = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
                finally:
                    plan.set_fuse_dct(prev)
            err = np.abs(got.astype(np.float64) - want)
            scale = float(np.abs(want).max())
            assert err.max() <= 1e-4 * scale, (path, err.max() / scale)             # the north-star bound
            assert err.max() <= 1.5e-5 * scale, (path, err.max() / scale)           # measured 2.5e-6 .. 5.6e-6
            excess = float((err - (1e-4 * np.abs(want) + 1e-3)).max())              # over the elementwise bound
            assert excess <= 1e-2, (path, excess)                                   # measured < 4e-3
    assert seen


def test_full_size_c4_stereo(gpu):
    """BASELINE configs[3] at full size: a [512, 2, 480000] stereo tensor (48 kHz, 10 s), n_fft 2048 / 80 mel
    / 40 MFCC.  Each channel goes through the STRIDED-row path (audio[:, ch, :], row stride 2 n -- the
    reference picks one channel, script/mfcc.py:377-380) and all 1024 channel-rows through the contiguous
    path: spot clips vs the oracle, strided == contiguous bit for bit, permutation equivariance, finite."""
    import torch
    kw, _, _ = load_golden("c4_am")
    plan = _plan(kw)
    assert plan.kernel_path == "radix16-wpf"
    B, n = 512, 480000
    g = torch.Generator(device=gpu).manual_seed(4)
    t = torch.arange(n, device=gpu, dtype=torch.float32) / 48000.0
    base = 0.3 * torch.sin(2 * np.pi * 220 * t) * (1 + 0.5 * torch.sin(2 * np.pi * 4 * t))
    audio = 0.05 * torch.randn((B, 2, n), generator=g, device=gpu)
    audio += base[None, None, :]
    audio[:, 1, :] *= 0.5                                   # the two channels differ
    allrows = plan.mfcc(audio.view(2 * B, n))               # [1024, 40, 1001]
    assert allrows.shape == (2 * B, 40, 1001) and bool(torch.isfinite(allrows).all())
    for ch in (0, 1):
        view = audio[:, ch, :]
        assert view.stride(0) == 2 * n and not view.is_contiguous()
        m = plan.mfcc(view)
        assert torch.equal(m, allrows[ch::2])
        for i in (0, 255, 511):
            mfcc_close(m[i].cpu().numpy(), O.mfcc(view[i].cpu().numpy(), O.OracleConfig(**kw)), f"c4 ch {ch} clip {i}")
    perm = torch.randperm(B, device=gpu, generator=g)
    assert torch.equal(plan.mfcc(audio[perm][:, 0, :]), allrows[0::2][perm])


def test_drop_in_get_MFCCS_change(gpu):
    from modulation_mfcc_amd import get_MFCCS_change
    kw, y, exp = load_golden("refdefault_am")
    tot, T = get_MFCCS_change(y, 10000, channelN=0, tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=512,
                              minFreq=100, maxFreq=10000, removeFirst=1, filtCutoff=12, filtOrd=6,
                              diffMethod="grad", outFilter="iir", outFiltType="low", outFiltCutOff=[12],
                              outFiltLen=6, outFiltPolyOrd=3)
    np.testing.assert_array_equal(T, exp["T"])
    assert tot.dtype == np.float64 and tot.shape == exp["totChange"].shape
    assert np.abs(tot - exp["totChange"]).max() <= 1e-4 * np.abs(exp["totChange"]).max()
    # the Savitzky-Golay differentiator (any diffMethod other than 'grad', script/mfcc.py:409-412) stays on the device too
    tot_sg, _ = get_MFCCS_change(y, 10000, tStep=0.005, diffMethod="sg", outFiltCutOff=[12])
    want_sg = O.mfcc_change_tail(exp["mfcc"].astype(np.float32), tStep=0.005, diffMethod="sg", outFiltCutOff=[12])
    assert np.abs(tot_sg - want_sg).max() <= 1e-4 * np.abs(want_sg).max()
    # stereo array + channel pick, float64 input (script/mfcc.py:377-380)
    st = np.stack([y.astype(np.float64), np.zeros_like(y, dtype=np.float64)])
    tot2, _ = get_MFCCS_change(st, 10000, channelN=0, tStep=0.005, outFiltCutOff=[12])
    np.testing.assert_allclose(tot2, tot, rtol=1e-12)


@pytest.mark.parametrize("name", ["c1_am", "c1_quiet_tail", "odd_22k", "c4_am"])
def test_float64_input_is_computed_in_float32(name, gpu):
    """An ndarray caller may hand the reference a float64 signal, which librosa then transforms in double precision
    (complex128 STFT; oracle mfcc_f64).  The build computes every input in float32 (librosa.load's dtype): the
    result stays inside the north-star tolerance of that double-precision path."""
    from modulation_mfcc_amd.mfcc import mfcc_array
    from modulation_mfcc_amd import MfccConfig
    kw, y, _ = load_golden(name)
    got = mfcc_array(y.astype(np.float64), MfccConfig(**kw))
    assert got.dtype == np.float32
    mfcc_close(got, O.mfcc_f64(y.astype(np.float64), O.OracleConfig(**kw)), f"{name}: float64 signal")


def test_error_behaviour(gpu):
    import torch
    kw, y, _ = load_golden("c1_am")
    plan = _plan(kw)
    with pytest.raises(TypeError):
        plan.mfcc(torch.zeros(100))                        # host tensor
    with pytest.raises(TypeError):
        plan.mfcc(torch.zeros(100, dtype=torch.float64, device=gpu))
    with pytest.raises(ValueError):
        plan.rfft(torch.zeros((2, 600), device=gpu), 512)  # row longer than n
    with pytest.raises(NotImplementedError):
        plan.rfft(torch.zeros((2, 100), device=gpu), 500)


@pytest.mark.parametrize("n", [1, 2, 3, 159, 160, 161, 511, 513, 10239, 10241, 20481])
def test_ragged_lengths_fast_and_generic(n, gpu):
    """Odd / tiny / tile-boundary clip lengths: edge masking of the radix-16 kernel == generic == oracle."""
    kw, _, _ = load_golden("c1_am")
    y = O.synth_clip(1000 + n, n, kw["sr"], "noise")
    plan = _plan(kw)
    want = O.mfcc(y, O.OracleConfig(**kw))
    for variant in VARIANTS:
        with _variant(plan, variant):
            got = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
        assert got.shape == want.shape == (13, 1 + n // 160)
        mfcc_close(got, want, f"{variant} n={n}")


@pytest.mark.parametrize("kwargs", [
    dict(outFiltCutOff=[12]), dict(outFilter=None), dict(removeFirst=0, outFiltCutOff=[20]),
    dict(filtOrd=5, filtCutoff=8, outFiltCutOff=[10], outFiltLen=3),
    dict(outFiltType="band", outFiltCutOff=[2, 20]), dict(outFiltType="high", outFiltCutOff=[3]),
    dict(diffMethod="sg", outFiltCutOff=[12]), dict(diffMethod="sg", outFilter=None, removeFirst=0),
    dict(outFilter="fir", outFiltCutOff=[12]), dict(outFilter="fir", outFiltType="high", outFiltCutOff=[5], outFiltLen=7),
    dict(outFilter="fir", outFiltType="band", outFiltCutOff=[2, 20], outFiltLen=8, diffMethod="sg"),
    dict(outFilter="sg", outFiltCutOff=[12], outFiltLen=7, outFiltPolyOrd=3),
    dict(outFilter="sg", outFiltCutOff=[12]),        # the reference's defaults: outFiltLen 6 (EVEN window), outFiltPolyOrd 3
    dict(outFilter="sg", outFiltCutOff=[12], outFiltLen=15, outFiltPolyOrd=2, removeFirst=0),
    dict(outFilter="fir", outFiltCutOff=[12], outFiltLen=21),       # beyond the device stencil: host round trip
])
@pytest.mark.parametrize("form", ["clip", "time-major"])
def test_change_tail_on_device(kwargs, form, gpu):
    """Row N1: mm_mfcc_change_f64 against scipy's sosfiltfilt / gradient / savgol_filter (the reference's
    tail, script/mfcc.py:392-427, both differentiators; the 'fir' / 'sg' output filters as a banded operator
    on the device).  Both device forms: the clip-resident launch and the time-major kernels it replaces."""
    from modulation_mfcc_amd import tail
    kw, y, exp = load_golden("refdefault_am")
    plan = _plan(kw)
    m = _dev(np.stack([exp["mfcc"], exp["mfcc"][::-1].copy() * 0.5]), gpu)      # two "clips"
    with _change_form(plan, form):
        got = tail.mfcc_change_device(plan, m, tStep=0.005, **kwargs).cpu().numpy()
    for i, mm in enumerate((exp["mfcc"], exp["mfcc"][::-1] * 0.5)):
        want = O.mfcc_change_tail(mm.astype(np.float32), tStep=0.005, **kwargs)
        assert got[i].shape == want.shape
        # float64 recursion, fused multiply-adds, the odd extension of the float32 MFCC rows formed in float32 exactly as
        # scipy.signal.sosfiltfilt forms it on a float32 array (formed in float64 the curve moves by 1e-7): what is left
        # is rounding, ~1e-12; tolerance 1e-10 of the curve's maximum
        assert np.abs(got[i] - want).max() <= 1e-10 * np.abs(want).max()


@contextlib.contextmanager
def _change_form(plan, form):
    prev = plan.set_fuse_tail(form == "clip")
    try:
        yield
    finally:
        plan.set_fuse_tail(prev)


@pytest.mark.parametrize("form", ["clip", "time-major"])
@pytest.mark.parametrize("n_mfcc,B,T", [(80, 3, 400), (40, 70, 1001), (2, 5, 300), (13, 3, 3001), (13, 2, 9000),
                                        (13, 2, 21000), (129, 2, 130), (13, 300, 100), (13, 1, 60001), (13, 70, 9000),
                                        (5, 3, 4097)])
def test_change_tail_many_rows_and_long_batches(n_mfcc, B, T, form, gpu):
    """Row N1 beyond the reference's 13 coefficients and 10 s: more rows per clip than fit LDS at once (the clip form
    takes them in groups; the time-major derivative kernel's row-walking variant), long clips (13 x 3001 and 5 x 4097:
    groups of rows), longer ones -- one recording of a minute at the reference's default 1 ms step, 2 and 70 x 9000,
    2 x 21000 frames: the segmented-rows form (a wave per 1088 samples) --, clip counts that are not a multiple of 64,
    short clips -- all against scipy's sequential arithmetic, and all again on the time-major kernels."""
    kw, _, _ = load_golden("c1_am")
    plan = _plan(dict(kw, n_mels=max(128, n_mfcc), n_mfcc=n_mfcc))
    from modulation_mfcc_amd import tail
    rng = np.random.default_rng(4)
    m = rng.standard_normal((B, n_mfcc, T)).cumsum(axis=2).astype(np.float32)
    with _change_form(plan, form):
        got = tail.mfcc_change_device(plan, _dev(m, gpu), tStep=0.01, outFiltCutOff=[12]).cpu().numpy()
    for i in (0, B // 2, B - 1):
        want = O.mfcc_change_tail(m[i], tStep=0.01, outFiltCutOff=[12])
        assert np.abs(got[i] - want).max() <= 1e-10 * np.abs(want).max()


@pytest.mark.parametrize("filt,kw", [
    ("fir", dict(cutOff=[12.0], filtLen=6)), ("fir", dict(cutOff=[30.0], filtLen=2)),
    ("fir", dict(cutOff=[20.0], filtLen=7, filtType="high")), ("fir", dict(cutOff=[5.0, 40.0], filtLen=8, filtType="band")),
    ("sg", dict(cutOff=[12.0], filtLen=5, polyOrd=3)), ("sg", dict(cutOff=[12.0], filtLen=13, polyOrd=2)),
    ("sg", dict(cutOff=[12.0], filtLen=6, polyOrd=3)),      # even window: the reference's default outFiltLen / outFiltPolyOrd
    ("sg", dict(cutOff=[12.0], filtLen=16, polyOrd=4)),
    ("iir", dict(cutOff=[12.0], filtLen=6)),
])
def test_apply_filter_on_device(filt, kw, gpu):
    """applyFilter (script/mfcc.py:29-135) on float64 curves that live on the GPU: every method stays there
    ('iir' mm_sosfiltfilt_f64; 'fir' = filtfilt and 'sg' = savgol_filter(mode='interp') as banded operators
    through mm_stencil_f64) and equals the reference's scipy call on the host; ragged lengths down to the
    shortest scipy admits, a single curve, scipy's own error for a curve that is too short."""
    import torch
    from modulation_mfcc_amd import applyFilter
    rng = np.random.default_rng(11)
    sr = 200.0
    n_min = 3 * kw["filtLen"] + 1 if filt == "fir" else (kw["filtLen"] if filt == "sg" else 40)
    for n in (n_min, n_min + 1, 333, 1001):
        x = rng.standard_normal((19, n)).cumsum(axis=1) + 3.0
        want = np.stack([applyFilter(r, sr, filt=filt, **kw) for r in x])
        got = applyFilter(_dev(x, gpu), sr, filt=filt, **kw)
        assert got.is_cuda and got.shape == want.shape
        tol = 1e-10 if filt == "iir" else 1e-12
        assert np.abs(got.cpu().numpy() - want).max() <= tol * np.abs(want).max(), (n, np.abs(got.cpu().numpy() - want).max())
        if filt == "iir":
            # float32 curves (librosa's RMS envelope is one): scipy forms their odd extension in float32 before its
            # recursion upcasts; the device does the same (mm_sosfiltfilt_f32_f64) -- an upcast copy would differ by 3e-9
            x32 = x.astype(np.float32)
            want32 = np.stack([applyFilter(r, sr, filt=filt, **kw) for r in x32])
            got32 = applyFilter(_dev(x32, gpu), sr, filt=filt, **kw)
            assert got32.dtype == torch.float64 and want32.dtype == np.float64
            assert np.abs(got32.cpu().numpy() - want32).max() <= 1e-10 * np.abs(want32).max()
    if filt == "fir":           # a float32 curve: scipy forms filtfilt's odd extension in float32, then filters in float64
        x32 = x.astype(np.float32)
        want32 = np.stack([applyFilter(r, sr, filt=filt, **kw) for r in x32])
        got32 = applyFilter(_dev(x32, gpu), sr, filt=filt, **kw)
        assert got32.dtype == torch.float64 and want32.dtype == np.float64 and got32.shape == want32.shape
        assert np.abs(got32.cpu().numpy() - want32).max() <= 1e-12 * np.abs(want32).max()
        one32 = applyFilter(_dev(x32[2], gpu), sr, filt=filt, **kw).cpu().numpy()
        assert one32.shape == (1001,) and np.abs(one32 - want32[2]).max() <= 1e-12 * np.abs(want32).max()
    if filt == "sg":            # a float32 curve stays float32 (scipy correlates in double and rounds once): same values
        x32 = x.astype(np.float32)
        want32 = np.stack([applyFilter(r, sr, filt=filt, **kw) for r in x32])
        got32 = applyFilter(_dev(x32, gpu), sr, filt=filt, **kw)
        assert got32.dtype == torch.float32 and want32.dtype == np.float32
        assert np.abs(got32.cpu().numpy() - want32).max() <= 2e-7 * np.abs(want32).max()
    one = applyFilter(_dev(x[3], gpu), sr, filt=filt, **kw).cpu().numpy()
    assert one.shape == (1001,) and np.abs(one - want[3]).max() <= 1e-10 * np.abs(want).max()
    if filt == "fir":
        with pytest.raises(ValueError, match="greater than padlen"):
            applyFilter(_dev(x[:, :3 * kw["filtLen"]], gpu), sr, filt=filt, **kw)


@pytest.mark.parametrize("order", [1, 2, 4, 6, 8, 10])
def test_iir_filter_of_long_rows_on_device(order, gpu):
    """applyFilter(filt='iir') on device curves of any length (the Hilbert envelope at the audio rate is 160 000 samples
    per ten-second clip): rows cut into segments of 64 x 17 samples, a wave each -- one segment, segment boundaries
    (n + 2 pad = 1088, 1089, 2 x 1088), more than 64 segments (several rounds of the row-level scan); Butterworth orders
    1 - 8 on the segmented form (1 - 4 sections), order 10 on the time-major kernels; all against scipy's sequential
    sosfiltfilt on the host."""
    from modulation_mfcc_amd import applyFilter
    rng = np.random.default_rng(order)
    pad = 3 * (2 * ((order + 1) // 2) + 1 - (order % 2))
    sr = 16000.0
    for n in (1088 - 2 * pad, 1089 - 2 * pad, 2176 - 2 * pad, 5000, 160000):
        rows = 3 if n > 10000 else 5
        x = np.abs(rng.standard_normal((rows, n)).cumsum(axis=1)) + rng.standard_normal((rows, n))
        want = np.stack([applyFilter(r, sr, filt="iir", cutOff=[12.0 if n > 10000 else 400.0], filtLen=order) for r in x])
        got = applyFilter(_dev(x, gpu), sr, filt="iir", cutOff=[12.0 if n > 10000 else 400.0], filtLen=order)
        assert got.is_cuda and got.shape == want.shape
        err = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        # a 12 Hz low-pass at 16 kHz has its poles within 5e-3 of the unit circle: the states of the cascade are large
        # against its output there and the three levels of Phi products carry ~5e-10 of rounding (1e-12 otherwise)
        assert err <= (1e-8 if n > 10000 else 1e-10), (n, err)
    # batches of 128 rows and more: a workgroup walks a row, 16 segments per round (one partial round; three rounds, the
    # last one partial; float32 rows)
    for rows, n, dt in ((150, 3000, np.float64), (130, 40000, np.float64), (200, 20000, np.float32)):
        x = (np.abs(rng.standard_normal((rows, n)).cumsum(axis=1)) + rng.standard_normal((rows, n))).astype(dt)
        got = applyFilter(_dev(x, gpu), sr, filt="iir", cutOff=[400.0], filtLen=order).cpu().numpy()
        for r in (0, 1, rows // 2, rows - 1):
            want = applyFilter(x[r], sr, filt="iir", cutOff=[400.0], filtLen=order)
            assert np.abs(got[r] - want).max() <= 1e-10 * np.abs(want).max(), (rows, n, r)


@pytest.mark.parametrize("kw", [
    dict(method="gradient", difference=1), dict(method="gradient", difference=2),
    dict(method="sg", width=3, polyOrder=2, difference=1), dict(method="sg", width=7, polyOrder=3, difference=2),
    dict(method="sg", width=15, polyOrder=4, difference=1),
    dict(method="sg", width=6, polyOrder=3, difference=1), dict(method="sg", width=4, polyOrder=2, difference=2),
    dict(method="finDiff", difference=1, accOrder=2), dict(method="finDiff", difference=2, accOrder=4),
    dict(method="finDiff", difference=1, accOrder=6),
])
def test_velocity_on_device(kw, gpu):
    """Row N2: get_velocity (script/calc.py:593-650) on the device -- mm_stencil_f64 on [rows, n] float64
    curves -- against the reference's own arithmetic on the host: np.gradient bit for bit, scipy's
    savgol_filter(mode='interp') and the findiff stencils to float64 round-off; ragged lengths down to the
    shortest the stencil admits, a strided view, a single curve."""
    import torch
    from modulation_mfcc_amd import get_velocity, velocity_batch
    rng = np.random.default_rng(5)
    sr = 200.0
    for n in (16, 17, 200, 1001):
        x = rng.standard_normal((37, n)).cumsum(axis=1)
        want = np.stack([get_velocity(r, sr, **kw) for r in x])
        got = velocity_batch(_dev(x, gpu), sr, **kw).cpu().numpy()
        if kw["method"] == "gradient":
            np.testing.assert_array_equal(got, want)
        else:
            assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    big = torch.zeros((5, 300), dtype=torch.float64, device=gpu)
    big[:, 3:203] = _dev(x[:5, :200], gpu)
    got = get_velocity(big[:, 3:203], sr, **kw).cpu().numpy()           # drop-in name, strided rows
    want = np.stack([get_velocity(r, sr, **kw) for r in x[:5, :200]])
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    one = get_velocity(_dev(x[0], gpu), sr, **kw).cpu().numpy()
    assert one.shape == (n,) and np.abs(one - get_velocity(x[0], sr, **kw)).max() <= 1e-12 * np.abs(want).max()
    # a float32 curve (an RMS envelope): numpy / scipy return float32 for 'gradient' and 'sg'; so does the device path
    x32 = x[:7].astype(np.float32)
    want32 = np.stack([get_velocity(r, sr, **kw) for r in x32])
    got32 = get_velocity(_dev(x32, gpu), sr, **kw)
    assert str(got32.dtype).endswith(str(want32.dtype)), (got32.dtype, want32.dtype)
    assert np.abs(got32.cpu().numpy().astype(np.float64) - want32).max() <= 1e-6 * np.abs(want32).max()


@pytest.mark.parametrize("deriv,acc", sorted(FORNBERG))
def test_findiff_on_device_is_pinned_by_the_published_tables(deriv, acc, gpu):
    """Row N2, 'finDiff' (script/calc.py:636: FinDiff(0, 1/sr, difference, acc=accOrder)): the device stencil against
    Fornberg's published coefficient tables typed into conftest.FORNBERG and applied by hand (central inside, forward
    / mirrored forward on the first / last samples) -- numbers this repository did not compute -- and against a
    polynomial whose derivative is exact at every sample, ends included."""
    from modulation_mfcc_amd import get_velocity
    rng = np.random.default_rng(100 + 10 * deriv + acc)
    sr = 200.0
    nf = len(FORNBERG[(deriv, acc)]["forward"])
    half = len(FORNBERG[(deriv, acc)]["central"]) // 2
    for n in (nf + half, 64, 1001):
        x = rng.standard_normal((9, n)).cumsum(axis=1)
        want = np.stack([fornberg_apply(r, 1 / sr, deriv, acc) for r in x])
        got = get_velocity(_dev(x, gpu), sr, difference=deriv, method="finDiff", accOrder=acc).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    t = np.arange(80) / sr
    poly = np.polynomial.Polynomial(rng.standard_normal(min(2 * half, nf - 1) + 1))
    got = get_velocity(_dev(poly(t)[None, :], gpu), sr, difference=deriv, method="finDiff", accOrder=acc).cpu().numpy()[0]
    exact = poly.deriv(deriv)(t)
    assert np.abs(got - exact).max() <= 1e-8 * max(1.0, np.abs(exact).max())


def test_velocity_of_the_change_curve(gpu):
    """What the UI does (script/main.py:668-713): the derivative of the MFCC-change curve -- here without
    leaving the device: mfcc -> change tail -> get_velocity, all clips of a batch at once."""
    from modulation_mfcc_amd import get_velocity, tail
    kw, y, exp = load_golden("refdefault_am")
    plan = _plan(kw)
    m = plan.mfcc(_dev(np.stack([y, y[::-1].copy()]), gpu))
    ch = tail.mfcc_change_device(plan, m, tStep=0.005, outFiltCutOff=[12])
    v = get_velocity(ch, 1 / 0.005, difference=1, method="gradient")
    hostv = np.stack([get_velocity(c, 1 / 0.005) for c in ch.cpu().numpy()])
    np.testing.assert_array_equal(v.cpu().numpy(), hostv)
    with pytest.raises(ValueError, match="Méthode inconnue"):
        get_velocity(ch, 200.0, method="nope")
    with pytest.raises(ValueError, match="window_length must be less than or equal"):
        get_velocity(ch[:, :5], 200.0, method="sg", width=7, polyOrder=2)


def test_change_tail_errors(gpu):
    from modulation_mfcc_amd import get_MFCCS_change
    kw, y, _ = load_golden("refdefault_am")
    with pytest.raises(ValueError, match="greater than padlen"):
        get_MFCCS_change(y[:500], 10000, tStep=0.005, outFiltCutOff=[12])    # 11 frames <= padlen 21
    with pytest.raises(Exception, match="smaller than the half"):
        get_MFCCS_change(y, 10000, tStep=0.005, outFiltCutOff=[100])
    with pytest.raises(Exception, match="CutOff is None"):
        get_MFCCS_change(y, 10000, tStep=0.005, outFiltCutOff=None)
    # the reference's default outFiltCutOff=[None] dies inside numpy/scipy with a TypeError
    with pytest.raises(TypeError):
        get_MFCCS_change(y, 10000, tStep=0.005)


def _random_cfgs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        n_fft = int(rng.choice([64, 128, 256, 512, 512, 512, 1024, 2048, 2048, 4096]))
        win = int(rng.integers(max(2, n_fft // 4), n_fft + 1))
        hop = int(rng.integers(1, max(2, win)))
        if rng.random() < 0.5:
            hop += hop & 1                      # even hops reach the radix-16 kernels
        sr = int(rng.choice([8000, 10000, 16000, 22050, 44100, 48000]))
        n_mels = int(rng.integers(2, min(129, n_fft // 2)))
        n_mfcc = int(rng.integers(1, min(n_mels, 40) + 1))
        fmin = float(rng.choice([0.0, 20.0, 100.0, 300.0]))
        fmax = float(rng.choice([sr / 2, sr / 2 * 0.9, sr * 0.7, 3000.0]))
        if fmax <= fmin + 50:
            continue
        out.append(dict(sr=sr, n_fft=n_fft, win_length=win, hop_length=max(1, hop), n_mels=n_mels, n_mfcc=n_mfcc,
                        fmin=fmin, fmax=fmax, top_db=float(rng.choice([80.0, 40.0, -1.0])),
                        preemph=float(rng.choice([0.0, 0.0, 0.97]))))
    return out


@pytest.mark.parametrize("idx", range(24))
def test_random_configs_match_oracle(idx, gpu):
    """Sweep of the configuration space (every kernel path, clamp on/off, pre-emphasis, filters
    above Nyquist, odd hops and windows) against the oracle, two ragged clips per configuration."""
    import warnings
    kw = _random_cfgs(24, 20260)[idx]
    rng = np.random.default_rng(idx)
    plan = _plan(kw)
    okw = dict(kw)
    okw["top_db"] = None if kw["top_db"] < 0 else kw["top_db"]
    n = int(rng.integers(kw["n_fft"] // 2, 6 * kw["n_fft"] + 7 * kw["hop_length"]))
    clips = np.stack([O.synth_clip(50 + idx, n, kw["sr"], k) for k in ("am", "quiet_tail")])
    got = plan.mfcc(_dev(clips, gpu)).cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(2):
            want = O.mfcc(clips[i], O.OracleConfig(**okw))
            assert got[i].shape == want.shape
            mfcc_close(got[i], want, f"cfg {idx} ({plan.kernel_path}) {kw} clip {i}")


@pytest.mark.parametrize("n,fl,hop,center", [(4000, 400, 80, True), (4000, 401, 77, True), (1600, 1600, 160, False),
                                              (50, 400, 80, True), (16000, 1600, 160, True),
                                              (48000, 2048, 160, True), (40000, 16384, 1, False),
                                              (70000, 20000, 4000, True)])
def test_rms_frames_on_device(n, fl, hop, center, gpu):
    """Row N3: mm_rms_f32 == librosa.feature.rms(center, pad_mode='constant') (script/calc.py:331)."""
    from modulation_mfcc_amd import rms_batch
    rng = np.random.default_rng(n + fl)
    x = rng.standard_normal((3, n)).astype(np.float32)
    got = rms_batch(_dev(x, gpu), fl, hop, center).cpu().numpy()
    for i in range(3):
        want = O.rms_envelope(x[i], fl, hop, center)
        assert got[i].shape == want.shape
        np.testing.assert_allclose(got[i], want, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_hilbert_envelope_on_device(dt, gpu):
    """Row N3: |scipy.signal.hilbert(x)| (script/calc.py:286) through mm_hilbert_envelope -- the library's own
    Stockham FFT at the clip's own length when that is 2^a 3^b 5^c 7^d, Bluestein chirp-z transforms over a
    power-of-two FFT otherwise -- against scipy in float64: lengths 1 .. 480 000 (every pass radix 2 .. 49 alone,
    first and later, the fused radix-256 / radix-625 passes first, later and behind small radices; primes; products), batches, a strided batch, a batch cut into workspace-bounded calls."""
    import scipy.signal
    from modulation_mfcc_amd import calc
    rng = np.random.default_rng(8)
    tol = 2e-5 if dt == np.float32 else 1e-11
    for n in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 15, 16, 17, 21, 22, 25, 27, 31, 32, 35, 45, 49, 63, 64, 75, 100, 105,
              121, 125, 127, 128, 147, 225, 245, 256, 257, 343, 512, 625, 1000, 1024, 2048, 2401, 4000, 4001, 4096,
              768, 1250, 5625, 6250, 8191, 8192, 12345, 16807, 44100, 65536, 99991, 117649, 160000, 390625, 441000, 480000):
        rows = 3 if n > 20000 else 5
        x = (rng.standard_normal((rows, n)) * np.linspace(0.2, 1.0, n)[None, :]).astype(dt)
        want = np.abs(scipy.signal.hilbert(x.astype(np.float64), axis=1))
        got = calc.hilbert_envelope_batch(_dev(x, gpu))
        assert got.is_cuda and got.shape == want.shape and got.cpu().numpy().dtype == dt
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= tol * max(want.max(), 1e-30), (n, err, want.max())
    # two clips share a complex transform (real part / imaginary part): each is scaled by a power of two first, so a
    # quiet clip next to a loud one keeps ITS OWN relative accuracy; all-zero clips, a clip with one sample set
    for n in (1000, 12345):
        amp = np.array([1.0, 1e-6, 1e3, 0.0, 3e-12, 7.0, 1.0])[:, None]
        x = (rng.standard_normal((7, n)) * amp).astype(dt)
        x[6, :] = 0.0; x[6, n // 3] = 5.0
        want = np.abs(scipy.signal.hilbert(x.astype(np.float64), axis=1))
        got = calc.hilbert_envelope_batch(_dev(x, gpu)).cpu().numpy()
        for r in range(7):
            assert np.abs(got[r] - want[r]).max() <= tol * want[r].max() + 1e-300, (n, r, np.abs(got[r] - want[r]).max(), want[r].max())
    # scipy.signal.hilbert is strictly per row -- a clip with a NaN or an infinity comes out NaN, its pair
    # partner (the clip next to it shares its complex transform) must not: such a clip is kept out of its pair
    for n in (1000, 160000):
        x = rng.standard_normal((6, n)).astype(dt)
        x[1, n // 2] = np.nan                     # partner: row 0
        x[2, 7] = np.inf                          # partner: row 3
        x[5, 0] = -np.inf                         # partner: row 4
        got = calc.hilbert_envelope_batch(_dev(x, gpu)).cpu().numpy()
        for r in (1, 2, 5):
            assert np.isnan(got[r]).all(), (n, r)
        for r in (0, 3, 4):
            want = np.abs(scipy.signal.hilbert(x[r].astype(np.float64)))
            assert np.isfinite(got[r]).all() and np.abs(got[r] - want).max() <= tol * want.max(), (n, r)
    assert len(calc._HILBERT_PLANS) <= calc.HILBERT_MAX_PLANS          # table memory stays bounded over many lengths
    # a strided view (every other row of a bigger batch) and a single clip
    big = _dev(rng.standard_normal((6, 3001)).astype(dt), gpu)
    got = calc.hilbert_envelope_batch(big[::2]).cpu().numpy()
    want = np.abs(scipy.signal.hilbert(big[::2].cpu().numpy().astype(np.float64), axis=1))
    assert np.abs(got - want).max() <= tol * want.max()
    one = calc.hilbert_envelope_batch(big[1]).cpu().numpy()
    assert one.shape == (3001,)
    assert np.abs(one - np.abs(scipy.signal.hilbert(big[1].cpu().numpy().astype(np.float64)))).max() <= tol * want.max()
    # more rows than one workspace-bounded call takes
    old = calc.HILBERT_WS_BYTES
    try:
        calc.HILBERT_WS_BYTES = 3 * 2 * 8192 * (8 if dt == np.float32 else 16)      # two pairs of clips per call
        x = rng.standard_normal((8, 3000)).astype(dt)
        got = calc.hilbert_envelope_batch(_dev(x, gpu)).cpu().numpy()
    finally:
        calc.HILBERT_WS_BYTES = old
    want = np.abs(scipy.signal.hilbert(x.astype(np.float64), axis=1))
    assert np.abs(got - want).max() <= tol * want.max()


def test_amplitude_envelope_on_device(gpu):
    """Row N3, drop-in: calculate_amplitude_envelope (script/calc.py:221-343) computes its envelope on the GPU
    -- 'RMS' through mm_rms_f32, 'Hilb' through a device FFT of the clip length -- for numpy input (numpy out,
    the reference's host output filter) and for a batch of clips on the device (device out, device filter);
    == librosa's rms restatement / scipy.signal.hilbert / scipy's filters."""
    import scipy.signal
    from modulation_mfcc_amd import calculate_amplitude_envelope, get_amplitude, applyFilter, sosfiltfilt_batch
    rng = np.random.default_rng(2)
    x = rng.standard_normal(4000).astype(np.float32)
    amp, t = calculate_amplitude_envelope(x, 8000.0, method="RMS", winLen=0.05, hopLen=0.01)
    assert isinstance(amp, np.ndarray) and amp.dtype == np.float32
    np.testing.assert_allclose(amp, O.rms_envelope(x, 400, 80), rtol=2e-6, atol=1e-7)
    assert len(t) == len(amp) and t[1] == pytest.approx(0.01)
    amp2, _ = get_amplitude(x, 8000.0, method="RMS", winLen=0.05, hopLen=0.01, outFilter="iir", outFiltCutOff=[12])
    want2 = O.apply_filter(O.rms_envelope(x, 400, 80), 100.0, filt="iir", cutOff=[12], filtLen=6, filtType="low", polyOrd=3)
    np.testing.assert_allclose(amp2, want2, rtol=1e-5, atol=1e-7)
    # Hilbert envelope: odd / even / non-power-of-two lengths, float32 and float64 like scipy
    for n, dt in ((4000, np.float32), (4001, np.float32), (160000, np.float32), (12345, np.float64)):
        xs = rng.standard_normal(n).astype(dt)
        amp_h, th = calculate_amplitude_envelope(xs, 8000.0, method="Hilb")
        want = np.abs(scipy.signal.hilbert(xs))
        assert amp_h.shape == want.shape and amp_h.dtype == want.dtype
        np.testing.assert_allclose(amp_h, want, rtol=0, atol=(2e-5 if dt == np.float32 else 1e-11) * want.max())
    # a batch on the device: envelope and IIR / Savitzky-Golay output filters without leaving the GPU
    xb = rng.standard_normal((5, 16000)).astype(np.float32)
    for filt, flen in (("iir", 6), ("sg", 7), ("fir", 9)):
        env, tb = calculate_amplitude_envelope(_dev(xb, gpu), 16000.0, method="RMS", winLen=0.025, hopLen=0.01,
                                               outFilter=filt, outFiltCutOff=[12], outFiltLen=flen)
        assert env.is_cuda and env.shape == (5, 101)
        for i in range(5):
            want = O.apply_filter(O.rms_envelope(xb[i], 400, 160), 100.0, filt=filt, cutOff=[12],       # (float32, as the
                                  filtLen=flen, filtType="low", polyOrd=3)                              # reference filters it)
            np.testing.assert_allclose(env[i].cpu().numpy(), want, rtol=1e-5, atol=1e-7)
    # mm_sosfiltfilt_f64 against scipy on ragged sizes, band-pass, odd order
    for rows, n, order, wn, bt in ((1, 22, 6, 0.2, "low"), (37, 1001, 5, 0.1, "low"), (70, 300, 4, [0.05, 0.3], "band"),
                                   (3, 5000, 8, 0.4, "high")):
        sos = scipy.signal.butter(order, wn, btype=bt, output="sos")
        xr = rng.standard_normal((rows, n)).cumsum(axis=1)
        got = sosfiltfilt_batch(_dev(xr, gpu), sos).cpu().numpy()
        want = scipy.signal.sosfiltfilt(sos, xr, axis=1)
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    with pytest.raises(ValueError, match="greater than padlen"):
        sosfiltfilt_batch(_dev(xr[:, :20], gpu), sos)
    with pytest.raises(Exception, match="smaller than the half"):
        applyFilter(_dev(xr, gpu), 100.0, filt="iir", cutOff=[60])


def test_resampler_design_sensitivity_on_the_ui_default_call(tmp_path, gpu):
    """Row N4, the stated deviation MEASURED (the reference resamples with soxr_hq, absent here; the build's filter is
    a Kaiser sinc to the same published specification): the UI's built-in curve -- a 44.1 kHz file, get_MFCCS_change
    (path, 10000, tStep .005, winLen .025, n_fft 512, maxFreq 10000, ...), script/main.py:732-769 -- through the
    device decode + resampler + MFCC + change tail, against the SAME path fed with the file resampled by two other
    independent high-quality designs (a steeper / deeper Kaiser sinc: pass band to 0.95 of Nyquist, 160 dB; scipy's
    resample_poly default).  What the spread says about parity with the reference on resampled files:
      * INSIDE the pass band (mel bank ending at 4 kHz < 0.913 x 5 kHz), away from the clip's ends, two designs of
        the 125+ dB class give the same MFCCs to float32 round-off (< 2e-6 of max|MFCC|; scipy's 60 dB default: 3e-4);
      * the first / last frames see the filters' different ringing past the clip's ends: up to ~1e-3 of max|MFCC|;
      * with the UI's maxFreq = 10000 (> Nyquist) the mel bank covers the resampler's TRANSITION band
        (4565 - 5000 Hz), where two filters of the same specification differ by many dB: MFCCs move by 1 - 3 % of
        max|MFCC| and the change curve by 10 - 20 % of its maximum.  The reference's own numbers there are a property
        of soxr's transition band; without soxr's coefficients they cannot be reproduced, only bounded."""
    import scipy.signal
    from test_host import _write_wav
    from modulation_mfcc_amd import MfccConfig, get_MFCCS_change, get_plan, load_audio
    from modulation_mfcc_amd.audio_io import design_taps, resample_ratio
    sr_in, sr_out = 44100, 10000
    L, M = resample_ratio(sr_in, sr_out)
    rng = np.random.default_rng(5)
    n = 3 * sr_in
    t = np.arange(n) / sr_in
    x = 0.25 * sum(np.sin(2 * np.pi * 110 * k * t) / k for k in range(1, 30)) * (1 + 0.6 * np.sin(2 * np.pi * 3 * t)) \
        + 0.03 * rng.standard_normal(n) + 0.1 * np.sin(2 * np.pi * (300 + 2000 * t / 3) * t)
    x = np.clip(x, -0.99, 0.99)
    path = str(tmp_path / "ui_default_44k.wav")
    _write_wav(path, x, sr_in, "int", 16)
    xq = load_audio(path)[0].cpu().numpy().astype(np.float64)            # the 16-bit samples as decoded
    h_steep, _ = design_taps(L, M, passband=0.95, stop_db=160.0)
    alts = {"steeper": scipy.signal.resample_poly(xq, L, M, window=h_steep / L).astype(np.float32),
            "scipy_default": scipy.signal.resample_poly(xq, L, M).astype(np.float32)}
    ours = load_audio(path, sr=sr_out)[0]
    assert ours.shape[0] == alts["steeper"].shape[0] == -(-n * L // M)
    ui = dict(tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=512, minFreq=100, removeFirst=1, filtCutoff=12, filtOrd=6,
              diffMethod="grad", outFilter="iir", outFiltCutOff=[12])
    spread = {}
    for maxF in (10000, 4000):
        cfg = MfccConfig.from_reference_call(sr_out, tStep=0.005, winLen=0.025, n_mfcc=13, n_fft=512, minFreq=100, maxFreq=maxF)
        plan = get_plan(cfg)
        tot0, T0 = get_MFCCS_change(path, sr_out, maxFreq=maxF, **ui)                      # decode + resample on the device
        m0 = plan.mfcc(ours[None, :])[0].cpu().numpy()
        for name, y in alts.items():
            tot, T = get_MFCCS_change(y, sr_out, maxFreq=maxF, **ui)
            m = plan.mfcc(_dev(y, gpu)[None, :])[0].cpu().numpy()
            np.testing.assert_array_equal(T, T0)
            per_frame = np.abs(m - m0).max(axis=0) / np.abs(m0).max()
            spread[(maxF, name)] = (float(per_frame[8:-8].max()), float(per_frame.max()),
                                    float(np.abs(tot - tot0).max() / np.abs(tot0).max()))
    print("resampler design spread (interior MFCC, all-frame MFCC, totChange):", spread)
    for name in alts:
        inner, allf, chg = spread[(4000, name)]
        # pass band, interior frames: two 125+ dB designs agree to float32 round-off (measured 2e-7); scipy's default
        # (Kaiser beta 5: ~0.03 dB of pass-band ripple, droop from 4.3 kHz) is not of that class (measured 3.3e-4)
        assert inner <= (2e-6 if name == "steeper" else 1e-3), (name, inner)
        assert allf <= 5e-3, (name, allf)              # clip ends: filter ringing (measured 1e-3 .. 2e-3)
        inner, allf, chg = spread[(10000, name)]
        assert 1e-3 <= allf <= 6e-2, (name, allf)      # the UI default: transition band inside the mel bank (measured 1 - 3 %)
        assert chg <= 0.4, (name, chg)                 # change curve: measured 12 % / 20 % of its maximum


def test_load_and_resample_on_device(tmp_path, gpu):
    """Row N4: WAVE decode and sample-rate conversion on the device (mm_pcm_decode_f32, mm_resample_f32) --
    every encoding against scipy.io.wavfile + libsndfile's scaling, the resampler against scipy's polyphase
    arithmetic with the SAME taps (kernel parity; the taps' quality is checked in tests/test_host.py), the
    drop-in load_channel / get_MFCCS_change on a path (the form the UI uses, script/main.py:750,1049)."""
    import scipy.io.wavfile
    import scipy.signal
    import torch
    from test_host import _write_wav
    from modulation_mfcc_amd import get_MFCCS_change, load_channel, load_audio, load_wav, resample_batch
    from modulation_mfcc_amd.audio_io import design_taps, resample_ratio
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.9, 0.9, (30001, 2))
    for kind, bits, ext in (("int", 8, False), ("int", 16, False), ("int", 24, False), ("int", 32, False),
                            ("float", 32, False), ("float", 64, False), ("int", 24, True)):
        for ch in (1, 2):
            p = str(tmp_path / f"d_{kind}{bits}_{ext}_{ch}.wav")
            _write_wav(p, x[:, :ch], 22050, kind, bits, ext)
            got, sr = load_wav(p)
            assert sr == 22050.0 and got.shape == (ch, 30001) and got.dtype == torch.float32
            if kind == "float":
                want = x[:, :ch].astype(np.float32 if bits == 32 else np.float64).astype(np.float32)
            elif bits == 8:
                want = ((np.clip(np.round(x[:, :ch] * 128 + 128), 0, 255) - 128) / 128).astype(np.float32)
            else:
                q = np.clip(np.round(x[:, :ch] * (1 << (bits - 1))), -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
                want = (q / float(1 << (bits - 1))).astype(np.float32)
            np.testing.assert_array_equal(got.cpu().numpy(), want.T)
            if not ext and not (kind == "int" and bits == 24):
                _, ref = scipy.io.wavfile.read(p)           # same samples as an independent reader sees
                assert ref.reshape(30001, -1).shape[1] == ch
    # resampler kernels == upfirdn with the same taps, both within 2e-6 of the maximum (the float32 rounding of the
    # result).  'f64' (vector pipe, float64 accumulation like scipy) measures 3 - 5e-8; 'auto' = the banded GEMM on the
    # matrix pipe wherever one period fits its LDS tile (exact float32 products, float32 accumulation in two tap-ordered
    # fmaf chains of ~taps / 2 terms) measures 4 - 8e-7 (tools/resample_err.py) -- -122 dB, far under the filter's own
    # 1e-5 pass-band ripple and three orders under what the MFCC tolerance needs (a 1e-5 sample error moves a log-mel
    # value by < 1e-4 dB)
    xs = rng.standard_normal((3, 20000)).astype(np.float32)
    TOL = {"f64": 2e-6, "auto": 2e-6}
    for sr_in, sr_out in ((48000, 16000), (44100, 16000), (44100, 10000), (16000, 10000), (8000, 16000), (22050, 16000), (16000, 16000)):
        L, M = resample_ratio(sr_in, sr_out)
        if L == M:
            np.testing.assert_array_equal(resample_batch(_dev(xs, gpu), sr_in, sr_out).cpu().numpy(), xs)
            continue
        h, half = design_taps(L, M)
        want = scipy.signal.resample_poly(xs.astype(np.float64), L, M, axis=1, window=h.astype(np.float32).astype(np.float64) / L)
        for method in ("auto", "f64"):
            got = resample_batch(_dev(xs, gpu), sr_in, sr_out, method=method).cpu().numpy()
            assert got.shape == want.shape == (3, -(-20000 * L // M))
            assert np.abs(got - want).max() <= TOL[method] * np.abs(want).max(), (sr_in, sr_out, method, np.abs(got - want).max() / np.abs(want).max())
    # clips shorter than the filter (every tap range is clipped at both ends), odd lengths, a single row, upsampling
    for n, sr_in, sr_out in ((1, 44100, 16000), (7, 44100, 16000), (300, 44100, 16000), (301, 48000, 16000), (999, 16000, 44100),
                             (4097, 22050, 16000), (50001, 44100, 16000)):
        xr = rng.standard_normal((2, n)).astype(np.float32)
        L, M = resample_ratio(sr_in, sr_out)
        h, half = design_taps(L, M)
        want = scipy.signal.resample_poly(xr.astype(np.float64), L, M, axis=1, window=h.astype(np.float32).astype(np.float64) / L)
        for method in ("auto", "f64"):
            got = resample_batch(_dev(xr, gpu), sr_in, sr_out, method=method).cpu().numpy()
            assert got.shape == want.shape == (2, -(-n * L // M))
            assert np.abs(got - want).max() <= TOL[method] * max(np.abs(want).max(), 1e-30), (n, sr_in, sr_out, method)
            one = resample_batch(_dev(xr[1], gpu), sr_in, sr_out, method=method).cpu().numpy()
            np.testing.assert_array_equal(one, got[1])
    # a batch of ten-second clips with unaligned row pitch (the 16-byte stores fall back to scalar ones), and rows > tile
    xb = torch.randn((37, 441000 + 3), device=gpu)[:, :441000]
    a_ = resample_batch(xb, 44100, 16000, method="mfma")
    b_ = resample_batch(xb, 44100, 16000, method="f64")
    assert a_.shape == b_.shape == (37, 160000) and float((a_ - b_).abs().max()) <= 2e-6 * float(b_.abs().max())
    # drop-in: a path at the file's own rate gives exactly what the array gives; a resampled path is close to it
    kw, y, exp = load_golden("refdefault_am")
    p = str(tmp_path / "clip.wav")
    _write_wav(p, np.stack([y, 0.5 * y], axis=1), 10000, "float", 32)
    ch = load_channel(p, 10000)
    assert isinstance(ch, np.ndarray) and ch.shape == (2, y.shape[0])
    np.testing.assert_array_equal(ch[0], y)
    a, Ta = get_MFCCS_change(p, 10000, channelN=0, tStep=0.005, outFiltCutOff=[12])
    b, Tb = get_MFCCS_change(y, 10000, tStep=0.005, outFiltCutOff=[12])
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(Ta, Tb)
    p2 = str(tmp_path / "clip48.wav")
    t = np.arange(48000 * 2) / 48000.0
    z = 0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 4 * t)) + 0.01 * rng.standard_normal(t.size)
    _write_wav(p2, z, 48000, "int", 16)
    lo = load_audio(p2, 16000)
    assert lo.shape == (1, 32000)
    c, _ = get_MFCCS_change(p2, 16000, tStep=0.01, outFiltCutOff=[12])
    assert c.shape == (201,) and np.isfinite(c).all()


def test_caller_provided_outputs_are_validated(gpu):
    """plan.mfcc / modspec / rfft hand out.data_ptr() to the C ABI: a wrong shape, dtype, stride or device is
    refused on the host instead of becoming an out-of-bounds device write."""
    import torch
    kw, y, _ = load_golden("c1_am")
    plan = _plan(kw)
    d = _dev(y, gpu)[None, :]
    T = plan.cfg.num_frames(y.shape[0])
    good = torch.empty((1, 13, T), dtype=torch.float32, device=gpu)
    assert plan.mfcc(d, out=good) is good
    for bad in (torch.empty((1, 13, T - 1), device=gpu), torch.empty((1, 13, T), dtype=torch.float64, device=gpu),
                torch.empty((1, 13, 2 * T), device=gpu)[:, :, ::2], torch.empty((1, 13, T))):
        with pytest.raises((ValueError, TypeError)):
            plan.mfcc(d, out=bad)
    with pytest.raises(ValueError):
        plan.modspec(good, out=torch.empty((1, 13, 10), dtype=torch.complex64, device=gpu))
    with pytest.raises(ValueError):
        plan.rfft(torch.zeros((4, 512), device=gpu), 512, out=torch.empty((4, 256), dtype=torch.complex64, device=gpu))
    # clips beyond 2^29 - 8192 samples are refused before anything is launched (32-bit sample offsets in the kernels)
    ws = plan.workspace(1, y.shape[0])
    rc = plan._lib.mm_mfcc_f32(plan._h, d.data_ptr(), 1, (1 << 29) - 8191, 1 << 30, good.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -1


def test_edge_sizes_of_the_side_kernels(gpu):
    """Smallest and oddest inputs of the round-2 entry points: resampling clips of 1 .. 3 samples and by 3 / 1,
    an empty WAVE data chunk, stencils on the shortest admissible curve, one-row batches, non-contiguous input."""
    import scipy.signal
    import torch
    from modulation_mfcc_amd import resample_batch, velocity_batch, sosfiltfilt_batch, rms_batch
    from modulation_mfcc_amd.audio_io import design_taps, resample_ratio
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 17):
        for sr_in, sr_out in ((48000, 16000), (16000, 48000), (44100, 16000)):
            x = rng.standard_normal((2, n)).astype(np.float32)
            got = resample_batch(_dev(x, gpu), sr_in, sr_out).cpu().numpy()
            L, M = resample_ratio(sr_in, sr_out)
            h, _ = design_taps(L, M)
            want = scipy.signal.resample_poly(x.astype(np.float64), L, M, axis=1, window=h.astype(np.float32).astype(np.float64) / L)
            assert got.shape == want.shape == (2, -(-n * L // M))
            np.testing.assert_allclose(got, want, rtol=0, atol=3e-6 * max(1.0, np.abs(want).max()))
    one = resample_batch(_dev(x[0], gpu), 16000, 8000)
    assert one.shape == (9,)
    # strided rows
    big = torch.zeros((3, 50), dtype=torch.float64, device=gpu)
    big[:, ::2] = _dev(rng.standard_normal((3, 25)), gpu)
    v = velocity_batch(big[:, ::2], 100.0)
    np.testing.assert_array_equal(v.cpu().numpy(), np.gradient(big[:, ::2].cpu().numpy(), 0.01, axis=1))
    two = velocity_batch(_dev(np.array([[1.0, 4.0]]), gpu), 10.0)               # n = 2: both outputs are edge rows
    np.testing.assert_array_equal(two.cpu().numpy(), np.gradient(np.array([[1.0, 4.0]]), 0.1, axis=1))
    with pytest.raises(ValueError):
        velocity_batch(_dev(np.array([[1.0]]), gpu), 10.0)
    sos = scipy.signal.butter(2, 0.3, output="sos")
    xr = rng.standard_normal((1, 10))
    np.testing.assert_allclose(sosfiltfilt_batch(_dev(xr, gpu), sos).cpu().numpy(), scipy.signal.sosfiltfilt(sos, xr, axis=1),
                               rtol=1e-9, atol=1e-12)
    r = rms_batch(_dev(np.ones((1, 5), dtype=np.float32), gpu), 4, 2, True)
    np.testing.assert_allclose(r.cpu().numpy(), O.rms_envelope(np.ones(5, dtype=np.float32), 4, 2, True)[None, :], rtol=1e-6)


def test_plans_with_different_lds_sizes_coexist(gpu):
    """The dynamic-LDS limit is a per-function attribute: creating a plan with a small mel table after
    one with a large table must not break launches of the first (n_mels 128 needs more LDS than 40)."""
    from modulation_mfcc_amd import MfccConfig, MfccPlan
    kw, y, exp = load_golden("c1_am")
    big = MfccPlan(MfccConfig(**{**kw, "n_mels": 128, "fmin": 0.0}))
    small = MfccPlan(MfccConfig(**kw))
    d = _dev(y, gpu)[None, :]
    mfcc_close(small.mfcc(d)[0].cpu().numpy(), exp["mfcc"], "small")
    mfcc_close(big.mfcc(d)[0].cpu().numpy(), O.mfcc(y, O.OracleConfig(**{**kw, "n_mels": 128, "fmin": 0.0})), "big")


def test_calls_are_graph_capturable(gpu):
    """include/modmfcc.h promises: no allocation, no synchronisation inside compute calls -- so a
    step (MFCC + modulation spectrum) can be captured into a hipGraph and replayed."""
    import torch
    kw, _, _ = load_golden("c1_am")
    plan = _plan(kw)
    clips = np.stack([O.synth_clip(300 + i, 16000, 16000, "am") for i in range(8)])
    audio = _dev(clips, gpu)
    m_ref = plan.mfcc(audio)
    ms_ref = plan.modspec(m_ref)
    m = torch.empty_like(m_ref)
    ms = torch.empty_like(ms_ref)
    plan.workspace(8, 16000)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                       # warm-up on the side stream
        plan.mfcc(audio, out=m)
        plan.modspec(m, out=ms)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.mfcc(audio, out=m)
        plan.modspec(m, out=ms)
    m.zero_(); ms.zero_()
    audio.copy_(_dev(clips[::-1].copy(), gpu))       # new input, same buffers
    g.replay()
    torch.cuda.synchronize()
    want = plan.mfcc(_dev(clips[::-1].copy(), gpu))
    assert torch.equal(m, want)
    assert torch.equal(ms, plan.modspec(want))


def test_plans_on_every_visible_device(gpu):
    """One process, several GPUs (a host that drives more than one card from one interpreter): the raised dynamic-LDS limit
    of a kernel is a PER-DEVICE function attribute (hipFuncSetAttribute acts on the current device), so the library keeps
    its "already set" flags per device.  A plan on every visible device -- headline configuration (160 KB of LDS), the
    reference default, a typed n_fft -- must launch and agree bit for bit with device 0.  Skipped on a one-GPU box."""
    import torch
    from modulation_mfcc_amd import MfccConfig, MfccPlan, calc
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("needs at least two visible GPUs")
    cfgs = [dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0),
            dict(sr=10000, n_fft=512, win_length=250, hop_length=50, n_mels=128, n_mfcc=13, fmin=100.0, fmax=10000.0),
            dict(sr=16000, n_fft=400, win_length=400, hop_length=160, n_mels=40, n_mfcc=13, fmin=100.0, fmax=8000.0)]
    host = np.stack([O.synth_clip(70 + i, 32000, 16000, ("am", "quiet_tail", "noise")[i % 3]) for i in range(256)])
    for kw in cfgs:
        want = None
        for d in range(n_dev):
            dev = torch.device("cuda", d)
            plan = MfccPlan(MfccConfig(**kw), device=dev)
            x = torch.from_numpy(host).to(dev)
            m, s = plan.mfcc_modspec(x)
            env = calc.hilbert_envelope_batch(x[:4])
            got = (m.cpu(), s.cpu(), env.cpu())
            if want is None:
                want = got
            else:
                for a, b in zip(got, want):
                    assert torch.equal(torch.view_as_real(a) if a.is_complex() else a,
                                       torch.view_as_real(b) if b.is_complex() else b), (kw, d)


def test_sharded_driver_single_rank_nccl(gpu):
    """The N > 1 code path (process group on RCCL, slab layout, gather, root-side modulation spectrum,
    the double-buffered PipelinedGather with its post hook) with a single-rank group: everything but
    the wire.  world_size-2 arithmetic is covered on CPU by tests/test_dist_gloo.py."""
    import os
    import torch
    import torch.distributed as dist
    from modulation_mfcc_amd import MfccConfig
    from modulation_mfcc_amd.dist import PipelinedGather, SlabLayout, mfcc_modspec_sharded
    kw, _, _ = load_golden("c1_am")
    cfg = MfccConfig(**kw)
    plan = _plan(kw)
    clips = np.stack([O.synth_clip(40 + i, 16000, 16000, "am") for i in range(5)])
    d = _dev(clips, gpu)
    want_m = plan.mfcc(d)
    want_ms = plan.modspec(want_m)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29641"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        for on_root in (True, False):
            m, ms = mfcc_modspec_sharded(d, cfg, with_modspec=True, modspec_on_root=on_root)
            assert torch.equal(m, want_m) and torch.equal(ms, want_ms)
        T = cfg.num_frames(16000)
        lay = SlabLayout.make(cfg, 5, 16000, False)
        pg = PipelinedGather(lay.numel, gpu)
        mod_all = [torch.empty_like(want_ms) for _ in range(pg.depth)]

        def post(i):
            plan.modspec(pg.recv_block[i][:, :lay.mfcc_numel].reshape(5, cfg.n_mfcc, T), out=mod_all[i])

        for k in range(5):                          # more steps than buffers
            slab = pg.acquire()
            mv, _ = lay.views(slab)
            plan.mfcc(d, out=mv)
            pg.submit(post=post)
        pg.finish()
        torch.cuda.synchronize()
        for i in range(pg.depth):
            assert torch.equal(pg.recv_block[i][0, :lay.mfcc_numel].view_as(want_m), want_m)
            assert torch.equal(mod_all[i], want_ms)
    finally:
        dist.destroy_process_group()


def test_timing_stage_mask(gpu):
    kw, y, _ = load_golden("c1_am")
    plan = _plan(kw)
    d = _dev(y, gpu)[None, :]
    plan.timing_enable(True, stages=["dct"])
    m = plan.mfcc(d)
    plan.modspec(m)
    plan.timing_enable(False)
    got = plan.timing_read()
    assert set(got) == {"dct"} and got["dct"][1] == 1
    plan.timing_enable(True)
    plan.modspec(plan.mfcc(d))
    plan.timing_enable(False)
    assert {"logmel", "dct", "modspec"} <= set(plan.timing_read())
