"""ctypes binding of libsttfe.so (include/sttfe.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes
import os

# torch must be imported BEFORE libsttfe.so is loaded: torch ships its own libamdhip64.so.7 and the two
# must share ONE HIP runtime (streams and device pointers cross the boundary).  Loading ours first
# would bind /opt/rocm's runtime instead and HIP then reports "no ROCm-capable device".
import torch  # noqa: F401
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# STTFE_LIB_PATH selects another build of the same library (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get("STTFE_LIB_PATH") or os.path.join(_HERE, "libsttfe.so")

_lib = None


class SttfeError(RuntimeError):
    pass


def _declare(lib):
    p = c_void_p
    lib.sttfe_version.restype = c_int
    lib.sttfe_strerror.restype = c_char_p
    lib.sttfe_strerror.argtypes = [c_int]
    lib.sttfe_last_error_detail.restype = c_char_p
    sigs = {
        "sttfe_whisper_tables_bytes": [c_int, POINTER(c_size_t)],
        "sttfe_whisper_tables_build": [c_int, p, c_size_t],
        "sttfe_whisper_mel_bank": [c_int, p],
        "sttfe_whisper_workspace_bytes": [c_int, POINTER(c_size_t)],
        "sttfe_whisper_logmel": [p, p, p, c_int, c_int, c_int, p, p, p, c_int, p, c_size_t, p],
        "sttfe_whisper_logmel_scaled": [p, p, p, c_int, c_int, c_int, p, p, p, c_int, p, p, c_size_t, p],
        "sttfe_whisper_logmel_torch": [p, p, p, c_int, c_int, c_int, p, p, p, c_int, c_int, c_float, p, c_size_t, p],
        "sttfe_whisper_logmel_static": [p, p, p, c_int, c_int, c_int, p, p, p, c_int, p],
        "sttfe_set_whisper_tiles_per_block": [c_int],
        "sttfe_set_whisper_uniform_kernel": [c_int],
        "sttfe_timing_enable": [c_int],
        "sttfe_timing_reset": [],
        "sttfe_timing_set_stride": [c_int],
        "sttfe_timing_seen": [c_char_p, POINTER(c_int)],
        "sttfe_timing_query": [c_char_p, POINTER(ctypes.c_double), POINTER(c_int)],
        "sttfe_kaldi_tables_bytes": [c_int, POINTER(c_size_t)],
        "sttfe_kaldi_tables_build": [c_int, p, c_size_t],
        "sttfe_kaldi_mel_bank": [c_int, p],
        "sttfe_kaldi_workspace_bytes": [c_int, c_int, POINTER(c_size_t)],
        "sttfe_kaldi_fbank_cmvn": [p, p, p, c_int, c_int, c_int, c_int, c_int, p, p, p, c_int, p, c_int, p, c_size_t, p],
        "sttfe_resample_plan_bytes": [c_int, c_int, POINTER(c_size_t)],
        "sttfe_resample_plan_build": [c_int, c_int, p, c_size_t],
        "sttfe_resample_plan_info": [p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)],
        "sttfe_resample_out_length": [c_int, c_int, c_int64, POINTER(c_int64)],
        "sttfe_host_mapped_pointer": [p, POINTER(c_void_p)],
        "sttfe_preprocess_workspace_bytes": [c_int, POINTER(c_size_t)],
        "sttfe_resample_peak": [p, p, p, c_int, c_int, c_int64, p, p, p, p, p, p, c_size_t, p],
        "sttfe_resample_peak_s16": [p, p, p, c_int, c_int, c_int64, p, p, p, p, p, p, c_size_t, p],
        "sttfe_normalize_trim": [p, p, p, c_int, c_int64, c_int, c_int, c_float, p, c_size_t, p],
        "sttfe_waveform_augment": [p, p, p, c_int, c_int64, p, p, p, c_int, p],
        "sttfe_pitch_shift_plan": [p, p, c_int, c_int, c_int, p, p],
        "sttfe_pitch_shift": [p, p, p, c_int, p, p, c_int, p, c_size_t, p],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue          # declared in the header but not built yet -> tests/test_boundary_cpu.py fails
        fn.restype = c_int
        fn.argtypes = args


def lib():
    """The loaded library (loads on first use).  Raises SttfeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SttfeError(
                f"{LIB_PATH} not found: build it with `make -C speech_to_text_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(code: int, what: str = "sttfe"):
    if code != 0:
        msg = lib().sttfe_strerror(code)
        detail = lib().sttfe_last_error_detail()
        raise SttfeError(f"{what} failed: {msg.decode() if msg else code} ({code})"
                         + (f" [{detail.decode()}]" if detail else ""))


def ptr(t):
    """Device/host address of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())
