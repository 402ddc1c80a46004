"""Cross-check the oracle against an INDEPENDENT librosa-compatible implementation.

TEST INFRASTRUCTURE, container-only: ``transformers.audio_utils`` documents
librosa compatibility for ``mel_filter_bank(norm='slaney', mel_scale='slaney')``
and ``spectrogram(center=True, pad_mode='constant', power=2, log_mel='dB',
db_range=80)``.  It is NOT the reference and does not pin parity; it guards the
restatement in ``mfcc_oracle.py`` against transcription mistakes.  Run:

    python oracle/crosscheck_transformers.py [--write]

Prints one line per case and exits non-zero when any case disagrees by more
than 1e-5 of max|MFCC| (float32 round-off is ~2e-7).  ``--write`` stores the
independent implementation's MFCCs (not the oracle's) as tests/golden_xcheck/*.npz
-- input RECIPE + configuration + expected array -- so that the oracle (CPU suite) and
the HIP path (GPU suite) are also asserted against numbers this repository did not
produce, on machines where transformers is absent.  They are evidence, not a pin:
the reference itself ships no vectors.
"""
import os
import sys

import numpy as np
import scipy.fftpack

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mfcc_oracle as O  # noqa: E402

CASES = [
    # name,            sr,    n_fft, win,  hop, n_mels, n_mfcc, fmin, fmax,  seconds, kind
    ("c1_16k_1s",      16000, 512,   400,  160, 40,     13,     100,  8000,  1.0,  "am"),
    ("c2_16k_10s",     16000, 512,   400,  160, 40,     13,     100,  8000,  10.0, "am"),
    ("ref_default",    10000, 512,   250,  50,  128,    13,     100,  10000, 1.0,  "am"),
    ("c4_48k",         48000, 2048,  1200, 480, 80,     40,     100,  10000, 0.5,  "am"),
    ("quiet_tail",     16000, 512,   400,  160, 40,     13,     100,  8000,  1.0,  "quiet_tail"),
    ("noise_1024",     22050, 1024,  551,  220, 64,     20,     0,    11025, 0.7,  "noise"),
]


def transformers_mfcc(y, sr, n_fft, win, hop, n_mels, n_mfcc, fmin, fmax):
    from transformers import audio_utils as au
    W = au.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels,
                           min_frequency=fmin, max_frequency=fmax, sampling_rate=sr,
                           norm="slaney", mel_scale="slaney")            # [bins, mels]
    window = au.window_function(win, "hann", periodic=True, frame_length=n_fft, center=True)
    S = au.spectrogram(y.astype(np.float64), window, frame_length=n_fft, hop_length=hop,
                       fft_length=n_fft, power=2.0, center=True, pad_mode="constant",
                       mel_filters=W, log_mel="dB", reference=1.0, min_value=1e-10,
                       db_range=80.0, dtype=np.float32)                  # [mels, T]
    M = scipy.fftpack.dct(S, axis=-2, type=2, norm="ortho")[:n_mfcc]
    return W.T.astype(np.float32), M.astype(np.float32)


def main():
    worst = 0.0
    write = "--write" in sys.argv
    outdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden_xcheck")
    for (name, sr, n_fft, win, hop, n_mels, n_mfcc, fmin, fmax, secs, kind) in CASES:
        n = int(sr * secs)
        y = O.synth_clip(7, n, sr, kind)
        cfg = O.OracleConfig(sr=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=n_mels,
                             n_mfcc=n_mfcc, fmin=fmin, fmax=fmax)
        ours = O.mfcc(y, cfg)
        Wt, theirs = transformers_mfcc(y, sr, n_fft, win, hop, n_mels, n_mfcc, fmin, fmax)
        Wo = O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        dW = float(np.max(np.abs(Wo - Wt)))
        assert ours.shape == theirs.shape == (n_mfcc, O.num_frames(n, hop)), (ours.shape, theirs.shape)
        d = float(np.max(np.abs(ours - theirs)))
        rel = d / float(np.max(np.abs(theirs)))
        worst = max(worst, rel)
        if write and theirs.nbytes <= 40000:
            import transformers
            os.makedirs(outdir, exist_ok=True)
            keys = ["sr", "n_fft", "win_length", "hop_length", "n_mels", "n_mfcc", "fmin", "fmax"]
            np.savez_compressed(os.path.join(outdir, name + ".npz"), recipe=np.array([7, n], dtype=np.int64), kind=np.array(kind),
                                cfg_keys=np.array(keys), cfg_vals=np.array([sr, n_fft, win, hop, n_mels, n_mfcc, fmin, fmax], dtype=np.float64),
                                mfcc=theirs, source=np.array(f"transformers {transformers.__version__} audio_utils + scipy.fftpack.dct"))
        print(f"{name:14s} frames={ours.shape[1]:5d} mel max|dW|={dW:.2e} "
              f"mfcc max|d|={d:.3e} rel-to-max={rel:.2e} empty_filters={(Wo.max(1) == 0).sum()}")
    print("worst rel:", worst)
    return 0 if worst < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
