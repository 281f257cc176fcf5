"""CPU checks of the drop-in boundary: libsttfe.so loads, exports every symbol include/sttfe.h declares,
validates arguments before touching the GPU, and its host-side table builders agree with the oracle.
Also enforces the oracle rule: nothing under speech_to_text_amd/ may import oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from speech_to_text_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sttfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sttfe_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/sttfe.h but not exported by libsttfe.so"
    assert L.sttfe_version() >= 100
    assert L.sttfe_strerror(-1) == b"invalid argument"


def test_ctypes_signatures_cover_the_header():
    L = _lib.lib()
    for n in header_symbols():
        fn = getattr(L, n)
        assert fn.argtypes is not None or n in ("sttfe_version", "sttfe_last_error_detail"), n


def test_argument_validation_needs_no_gpu():
    L = _lib.lib()
    n = ctypes.c_size_t()
    assert L.sttfe_whisper_tables_bytes(0, ctypes.byref(n)) == -1
    assert L.sttfe_whisper_tables_bytes(129, ctypes.byref(n)) == -1
    assert L.sttfe_whisper_workspace_bytes(-1, ctypes.byref(n)) == -1
    assert L.sttfe_whisper_logmel(None, None, None, 1, 80, 3000, None, None, None, 0, None, 0, None) == -1
    assert L.sttfe_whisper_logmel(None, None, None, 0, 80, 3000, None, None, None, 0, None, 0, None) == 0
    assert L.sttfe_whisper_logmel(None, None, None, 1, 80, 3001, None, None, None, 0, None, 0, None) == -1
    assert L.sttfe_kaldi_fbank_cmvn(None, None, None, 1, 80, 10, 0, 0, None, None, None, 4, None, 0, None, 0, None) == -1
    assert L.sttfe_kaldi_fbank_cmvn(None, None, None, 0, 80, 10, 0, 0, None, None, None, 4, None, 0, None, 0, None) == 0
    assert L.sttfe_resample_plan_bytes(48000, 0, ctypes.byref(n)) == -1
    assert L.sttfe_resample_plan_bytes(48000, 16001, ctypes.byref(n)) == 0 and n.value == 32     # reduced rates > 1024:
    # a plan without a table (header only), run by the table-free kernel; beyond 2^22 : 2^22 unsupported
    assert L.sttfe_resample_plan_bytes(2 ** 23 + 1, 2 ** 23, ctypes.byref(n)) == -5
    assert L.sttfe_resample_peak(None, None, None, 1, 1, 10, None, None, None, None, None, None, 0, None) == -1
    assert L.sttfe_normalize_trim(None, None, None, 1, 10, 1, 1, ctypes.c_float(0.01), None, 0, None) == -1
    assert L.sttfe_normalize_trim(None, None, None, 1, 10, 0, 0, ctypes.c_float(0.01), None, 0, None) == 0


@pytest.mark.parametrize("n_mels", [80, 128])
def test_whisper_mel_bank_matches_oracle(n_mels):
    from oracle.whisper import mel_filter_bank_slaney
    dense = np.empty((n_mels, 201), dtype=np.float32)
    _lib.check(_lib.lib().sttfe_whisper_mel_bank(n_mels, dense.ctypes.data_as(ctypes.c_void_p)))
    ref = mel_filter_bank_slaney(n_mels).astype(np.float32)            # (201, n_mels)
    assert np.abs(dense.T - ref).max() <= 1e-9


def test_generated_mel_bands_describe_the_bank():
    """csrc/whisper_mel_bands.h (tools/gen_mel_bands.py) holds the first / last non-zero bin of every mel row: the 12-wave
    kernel's band-sparse projection skips every product outside them.  The tables builder checks the header against the bank
    it builds and says so in the blob's header (word 3, bits 0 and 1: bands verified, weight-pair stream appended): set for the two Whisper extractors, clear for any other bin
    count (those take the MFMA projection).  The header itself is what the generator would write today."""
    import struct
    L = _lib.lib()
    for n_mels, want in ((80, 3), (128, 3), (40, 0), (64, 0)):
        n = ctypes.c_size_t()
        _lib.check(L.sttfe_whisper_tables_bytes(n_mels, ctypes.byref(n)))
        blob = ctypes.create_string_buffer(n.value)
        _lib.check(L.sttfe_whisper_tables_build(n_mels, blob, n))
        magic, nm, afrag_floats, flags = struct.unpack("Iiii", blob.raw[:16])
        assert nm == n_mels and (flags & 3) == want
    text = open(os.path.join(ROOT, "speech_to_text_amd", "csrc", "whisper_mel_bands.h")).read()
    for n_mels in (80, 128):
        dense = np.empty((n_mels, 201), dtype=np.float32)
        _lib.check(L.sttfe_whisper_mel_bank(n_mels, dense.ctypes.data_as(ctypes.c_void_p)))
        lo = [int(np.nonzero(r)[0][0]) for r in dense]
        hi = [int(np.nonzero(r)[0][-1]) for r in dense]
        assert f"lo[{n_mels}] = {{{', '.join(map(str, lo))}}};" in text
        assert f"hi[{n_mels}] = {{{', '.join(map(str, hi))}}};" in text
        assert all(int(np.count_nonzero(r)) == h - l + 1 for r, l, h in zip(dense, lo, hi))
        # the weight-pair stream behind the A fragments: every mel's pairs (W[m][k], W[m][k + 1]) / 4 in the order of the
        # MFMA form's two accumulation chains (bin offsets 0 4 8 12 2 6 10 14 of every 16-bin group of the mel tile's k range)
        plan = {80: ([0, 12, 28, 56, 104], [4, 8, 8, 16, 24]), 128: ([0, 8, 16, 28, 40, 60, 88, 124], [4, 4, 4, 4, 8, 12, 12, 20])}[n_mels]
        want = []
        for m in range(n_mels):
            ks, ns = plan[0][m // 16], plan[1][m // 16]
            for g in range(ns // 4):
                for h in range(2):
                    for kq in range(4):
                        k = ks + 16 * g + 4 * kq + 2 * h
                        if ks + 16 * g + 15 < lo[m] or ks + 16 * g > hi[m] or k + 1 < lo[m] or k > hi[m]:
                            continue
                        want += [dense[m, k] * 0.25, dense[m, k + 1] * 0.25 if k + 1 < 201 else 0.0]
        n = ctypes.c_size_t()
        _lib.check(L.sttfe_whisper_tables_bytes(n_mels, ctypes.byref(n)))
        blob = ctypes.create_string_buffer(n.value)
        _lib.check(L.sttfe_whisper_tables_build(n_mels, blob, n))
        afrag_floats = struct.unpack("Iiii", blob.raw[:16])[2]
        stream = np.frombuffer(blob.raw, dtype=np.float32)[-len(want):]
        assert n.value % 4 == 0 and len(blob.raw) // 4 - len(want) >= afrag_floats
        assert np.array_equal(stream, np.asarray(want, dtype=np.float32))


def test_kaldi_mel_bank_matches_oracle():
    from oracle.kaldi import mel_filter_bank_kaldi
    dense = np.empty((80, 257), dtype=np.float32)
    _lib.check(_lib.lib().sttfe_kaldi_mel_bank(80, dense.ctypes.data_as(ctypes.c_void_p)))
    assert np.abs(dense.T - mel_filter_bank_kaldi(80).astype(np.float32)).max() <= 1e-7


@pytest.mark.parametrize("sr", [48000, 32000, 44100, 22050, 8000, 96000])
def test_resample_plan_matches_oracle_kernel(sr):
    from oracle.resample import sinc_resample_kernel
    from speech_to_text_amd.preprocess import ResamplePlan
    plan = ResamplePlan(sr, 16000)
    kern, width, orig, new = sinc_resample_kernel(sr, 16000)
    assert (plan.orig, plan.new, plan.width, plan.n_taps) == (orig, new, width, kern.shape[1])
    assert np.abs(plan.taps - kern).max() <= 1e-7
    n_out = ctypes.c_int64()
    for n in (0, 1, 2, 3, 479, 480001):
        _lib.check(_lib.lib().sttfe_resample_out_length(sr, 16000, n, ctypes.byref(n_out)))
        assert n_out.value == -(-new * n // orig)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "speech_to_text_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may import the oracle only inside cpu_baseline()
    head, _, rest = bench.partition("def cpu_baseline")
    body, _, tail = rest.partition("\ndef ")
    assert "oracle" not in head and not re.search(r"^\s*(from|import)\s+oracle\b", tail, flags=re.M)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsttfe.so")
    with pytest.raises(_lib.SttfeError, match="no CPU fallback"):
        _lib.lib()


def test_extractors_refuse_to_run_without_a_gpu():
    import torch
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fe = HipWhisperFeatureExtractor(80)
    with pytest.raises(_lib.SttfeError, match="no CPU path"):
        fe(np.zeros(16000, np.float32), sampling_rate=16000)
    with pytest.raises(ValueError):
        fe(np.zeros(16000, np.float32), sampling_rate=8000)
