"""modulation_mfcc_amd -- MI355X-native MFCC + modulation-spectrum extractor.

Drop-in for the numeric hot path of aaron-randreth/modulation-mfcc (script/mfcc.py,
script/calc.py); Python host code over hand-written gfx950 HIP kernels (libmodmfcc.so, C ABI in
include/modmfcc.h).  torch is used for device buffers, streams and torch.distributed only.
"""
from .plan import MfccConfig, MfccPlan, get_plan, butter_sos  # noqa: F401
from .batch import mfcc_batch, modspec_batch, mfcc_modspec_batch, rfft_batch, rms_batch  # noqa: F401
from .mfcc import get_MFCCS_change, load_channel, applyFilter, get_amplitude  # noqa: F401
from .calc import (get_velocity, calculate_amplitude_envelope, velocity_batch, amplitude_envelope_batch,  # noqa: F401
                   hilbert_envelope_batch)
from .filters import sosfiltfilt_batch  # noqa: F401
from .audio_io import load_audio, load_wav, resample_batch  # noqa: F401

__version__ = "0.2.0"
