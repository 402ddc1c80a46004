"""ctypes binding of libmodmfcc.so (include/modmfcc.h).

There is NO CPU fallback: when the shared library cannot be loaded, or a plan cannot be created
because no MI355X is visible, the calls raise.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C modulation_mfcc_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MODMFCC_LIB: load another build of the library (development: ablation / stamp builds made by tools/*.sh
# go to their own files instead of overwriting the product library)
LIB_PATH = os.environ.get("MODMFCC_LIB") or os.path.join(_HERE, "libmodmfcc.so")

MM_OK = 0
MM_ERR_INVALID_ARG = -1
MM_ERR_UNSUPPORTED = -2
MM_ERR_HIP = -3
MM_ERR_WORKSPACE = -4
MM_ERR_ALLOC = -5

STAGES = ("logmel", "dct", "modspec", "rfft", "power", "change", "init", "reserved")
MM_NUM_STAGES = 8


class mm_config(C.Structure):
    _fields_ = [
        ("sr", C.c_double),
        ("n_fft", C.c_int32),
        ("win_length", C.c_int32),
        ("hop_length", C.c_int32),
        ("n_mels", C.c_int32),
        ("n_mfcc", C.c_int32),
        ("fmin", C.c_double),
        ("fmax", C.c_double),
        ("preemph", C.c_float),
        ("top_db", C.c_float),
        ("amin", C.c_float),
        ("center", C.c_int32),
        ("n_mod_fft", C.c_int32),
    ]


MM_ST_MAXW, MM_ST_MAXE = 16, 8


class mm_stencil(C.Structure):
    _fields_ = [
        ("n_c", C.c_int32), ("n_edge", C.c_int32), ("edge_w", C.c_int32), ("reserved", C.c_int32),
        ("off", C.c_int32 * MM_ST_MAXW),
        ("c", C.c_double * MM_ST_MAXW),
        ("el", (C.c_double * MM_ST_MAXW) * MM_ST_MAXE),
        ("er", (C.c_double * MM_ST_MAXW) * MM_ST_MAXE),
        ("den_c", C.c_double), ("den_e", C.c_double),
    ]


class MMError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: {detail}" if detail else what)


_vp = C.c_void_p
_i64 = C.c_int64
_cfgp = C.POINTER(mm_config)

# name -> (restype, argtypes); every symbol include/modmfcc.h declares
PROTOTYPES = {
    "mm_version": (C.c_int, []),
    "mm_strerror": (C.c_char_p, [C.c_int]),
    "mm_last_hip_error": (C.c_char_p, []),
    "mm_config_default": (C.c_int, [_cfgp]),
    "mm_config_validate": (C.c_int, [_cfgp]),
    "mm_num_frames": (_i64, [_cfgp, _i64]),
    "mm_num_bins": (C.c_int32, [_cfgp]),
    "mm_mod_fft_len": (C.c_int32, [_cfgp, _i64]),
    "mm_build_window": (C.c_int, [_cfgp, _vp]),
    "mm_build_mel": (C.c_int, [_cfgp, _vp]),
    "mm_build_dct": (C.c_int, [_cfgp, _vp]),
    "mm_build_mel_sweep": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _vp, _vp]),
    "mm_build_mel_runs": (C.c_int, [_cfgp, C.c_int, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp]),
    "mm_build_butter_sos": (C.c_int, [C.c_int, C.c_double, _vp]),
    "mm_plan_create": (C.c_int, [_cfgp, C.POINTER(_vp)]),
    "mm_plan_destroy": (C.c_int, [_vp]),
    "mm_plan_config": (C.c_int, [_vp, _cfgp]),
    "mm_plan_kernel_path": (C.c_int, [_vp]),
    "mm_plan_fused_dct": (C.c_int, [_vp]),
    "mm_plan_set_fuse_dct": (C.c_int, [_vp, C.c_int]),
    "mm_plan_fused_tail": (C.c_int, [_vp, _i64, _i64]),
    "mm_plan_set_fuse_tail": (C.c_int, [_vp, C.c_int]),
    "mm_mfcc_modspec_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, C.c_size_t, _vp]),
    "mm_plan_force_generic": (C.c_int, [_vp, C.c_int]),
    "mm_plan_set_variant": (C.c_int, [_vp, C.c_int]),
    "mm_workspace_bytes": (C.c_size_t, [_vp, _i64, _i64]),
    "mm_mfcc_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, C.c_size_t, _vp]),
    "mm_logmel_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "mm_stft_power_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mm_rfft_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, C.c_int32, _vp, _vp]),
    "mm_modspec_f32": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "mm_mfcc_change_f64": (C.c_int, [_vp, _vp, _i64, _i64, C.c_int32, C.c_int32, _vp, C.c_int32, _vp,
                                     C.c_int32, _vp, _vp, C.c_size_t, _vp]),
    "mm_sosfiltfilt_f64": (C.c_int, [_vp, _i64, _i64, _i64, _vp, C.c_int32, _vp, _vp, C.c_size_t, _vp]),
    "mm_sosfiltfilt_workspace_bytes": (C.c_size_t, [_i64, _i64]),
    "mm_sosfiltfilt_f32_f64": (C.c_int, [_vp, _i64, _i64, _i64, _vp, C.c_int32, _vp, _vp, C.c_size_t, _vp]),
    "mm_stencil_f64": (C.c_int, [C.POINTER(mm_stencil), _vp, _i64, _i64, _i64, _vp, _vp]),
    "mm_change_workspace_bytes": (C.c_size_t, [_vp, _i64, _i64]),
    "mm_change_workspace_bytes_for": (C.c_size_t, [_vp, _i64, _i64, C.c_int32, _vp, C.c_int32, _vp, C.c_int32]),
    "mm_rms_num_frames": (_i64, [_i64, C.c_int32, C.c_int32, C.c_int32]),
    "mm_rms_f32": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "mm_hilbert_create": (C.c_int, [_i64, C.c_int32, C.POINTER(_vp)]),
    "mm_hilbert_destroy": (None, [_vp]),
    "mm_hilbert_fft_size": (_i64, [_vp]),
    "mm_hilbert_workspace_bytes": (C.c_size_t, [_vp, _i64]),
    "mm_hilbert_envelope": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, C.c_size_t, _vp]),
    "mm_hilbert_rfft_workspace_bytes": (C.c_size_t, [_vp, _i64]),
    "mm_hilbert_rfft_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, C.c_size_t, _vp]),
    "mm_pcm_decode_f32": (C.c_int, [_vp, C.c_int32, C.c_int32, _i64, _vp, _i64, _vp]),
    "mm_resample_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, C.c_int32, C.c_int32, C.c_int32, _i64, _vp, _i64, _vp]),
    "mm_resample_banded_f32": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, _vp, _i64, _vp]),
    "mm_devcopy_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mm_timing_enable": (C.c_int, [_vp, C.c_int]),
    "mm_timing_read": (C.c_int, [_vp, _vp, _vp]),
}

_lib = None


def load():
    """Load libmodmfcc.so (once).  Raises ImportError -- never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: its wheel bundles the HIP runtime (torch/lib/libamdhip64.so), libmodmfcc.so was linked against
    # /opt/rocm's.  Whichever is loaded first serves both (same SONAME); loaded the other way round the process
    # ends up with two runtimes and the second one finds no device ("hipGetDevice failed").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`make -C modulation_mfcc_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "modulation_mfcc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError when the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.mm_version() < 121:
        raise ImportError("libmodmfcc.so is older than the Python binding; rebuild it")
    _lib = lib
    return lib


def check(status, what):
    if status == MM_OK:
        return
    lib = load()
    msg = lib.mm_strerror(int(status)).decode()
    detail = lib.mm_last_hip_error().decode() if status == MM_ERR_HIP else ""
    if status == MM_ERR_INVALID_ARG:
        raise ValueError(f"{what}: {msg}")
    if status == MM_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise MMError(status, f"{what}: {msg}", detail)
