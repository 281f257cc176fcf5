"""The measurement support of the C ABI (no counterpart in the reference): per-kernel HIP-event timing, sampled
every n-th launch (``sttfe_timing_set_stride``) with the total launch count kept (``sttfe_timing_seen``) -- what
bench.py's roofline figure is made of."""
import ctypes

import pytest
import torch

from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def query(L, name):
    ms, n, seen = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.sttfe_timing_query(name, ctypes.byref(ms), ctypes.byref(n)))
    _lib.check(L.sttfe_timing_seen(name, ctypes.byref(seen)))
    return ms.value, n.value, seen.value


def test_sampled_event_timing_counts_and_durations():
    L = _lib.lib()
    fe = HipWhisperFeatureExtractor(80, device="cuda:0")
    clips = synth_batch(5, [16000] * 8)
    try:
        _lib.check(L.sttfe_timing_enable(64))
        for stride, launches in ((1, 6), (4, 9)):
            _lib.check(L.sttfe_timing_set_stride(stride))          # also forgets earlier launches
            _lib.check(L.sttfe_timing_reset())
            for _ in range(launches):
                fe(clips, sampling_rate=16000)
            torch.cuda.synchronize()
            ms, timed, seen = query(L, b"whisper_stft_mel12")       # 80 bins, HF mode, <= 3072 clips: the 12-wave kernel alone
            assert seen == launches and timed == (launches + stride - 1) // stride
            assert 0.0 < ms / timed < 50.0                          # a few microseconds to milliseconds per launch
            assert query(L, b"whisper_clamp_fill")[1:] == (timed, seen)
            assert query(L, b"whisper_stft_mel")[1:] == (0, 0)      # (the general kernel is not launched for such a call)
            assert query(L, b"kaldi_fbank")[1:] == (0, 0)
        assert L.sttfe_timing_set_stride(0) == -1
        n = ctypes.c_int()
        assert L.sttfe_timing_seen(b"no_such_kernel", ctypes.byref(n)) == -1
    finally:
        L.sttfe_timing_set_stride(1)
        L.sttfe_timing_enable(0)
    for _ in range(2):
        fe(clips, sampling_rate=16000)                              # disabled: nothing is recorded
    assert query(L, b"whisper_stft_mel12")[1] == 0
