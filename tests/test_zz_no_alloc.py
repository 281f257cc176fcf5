"""sttfe_whisper_logmel_static (the entry point that allocates nothing) against sttfe_whisper_logmel, through the C ABI.

Bit-exact bar (same kernels, same arithmetic): two calls in a row on one GPU, the second with a smaller batch, so
that whatever the first call left in the library's static device workspace must not leak into the second.
Collected last (file name): the round's end runs `pytest -x`."""
import ctypes

import pytest
import torch

from speech_to_text_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def _call_static(fe, pcm, offsets, lengths, mask_table=None):
    from speech_to_text_amd import _lib
    L = _lib.lib()
    device = pcm.device
    B = int(offsets.shape[0])
    out = torch.full((B, fe.feature_size, fe.nb_max_frames), float("nan"), dtype=torch.float32, device=device)
    n_masks = 0 if mask_table is None else int(mask_table.shape[1])
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        rc = L.sttfe_whisper_logmel_static(_lib.ptr(pcm), _lib.ptr(offsets), _lib.ptr(lengths), B, fe.feature_size,
                                           fe.nb_max_frames, _lib.ptr(fe._tables_on(device)), _lib.ptr(out),
                                           _lib.ptr(mask_table), n_masks, ctypes.c_void_p(stream))
    return rc, out


def test_static_entry_point_matches_workspace_entry_point():
    from speech_to_text_amd import _lib
    from speech_to_text_amd.processors import HipWhisperFeatureExtractor, pack_clips
    dev = torch.device("cuda:0")
    fe = HipWhisperFeatureExtractor(80, device=dev)
    # call 1: a ragged batch with loud clips (large max keys, flagged tiles); call 2: fewer, quieter, shorter clips --
    # stale keys of call 1 would raise their clamp floor and change their padding constant
    lens1 = [480000, 52001, 300, 96000, 123457, 31999, 480000, 7480, 1, 250000, 16000, 200]
    clips1 = synth_batch(301, lens1)
    clips2 = [c[: max(1, len(c) // 3)] * 1e-3 for c in synth_batch(302, [90000, 480000, 4000, 160, 33333])]
    tbl1 = torch.tensor([[[1, 100, 117], [1, 2990, 3000], [0, 5, 20], [0, 70, 80]]] * len(lens1), dtype=torch.int32,
                        device=dev)
    for clips, tbl in ((clips1, tbl1), (clips2, None)):
        pcm, offsets, lengths = pack_clips(list(clips), dev, fe.n_samples)
        ref = fe.extract(pcm, offsets, lengths, mask_table=tbl).clone()
        rc, got = _call_static(fe, pcm, offsets, lengths, tbl)
        assert rc == 0, _lib.lib().sttfe_last_error_detail()
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
    # the small batch once more (the array now holds its own keys from the previous call): same bits
    pcm, offsets, lengths = pack_clips(list(clips2), dev, fe.n_samples)
    rc, again = _call_static(fe, pcm, offsets, lengths)
    assert rc == 0 and torch.equal(again, fe.extract(pcm, offsets, lengths))


def test_static_entry_point_limits():
    from speech_to_text_amd import _lib
    L = _lib.lib()
    # beyond the static workspace: refused before anything is touched (null pointers never dereferenced)
    rc = L.sttfe_whisper_logmel_static(None, None, None, 4097, 80, 3000, None, None, None, 0, None)
    assert rc == -5 and b"static workspace" in L.sttfe_last_error_detail()
    assert L.sttfe_whisper_logmel_static(None, None, None, 0, 80, 3000, None, None, None, 0, None) == 0
    assert L.sttfe_whisper_logmel_static(None, None, None, -1, 80, 3000, None, None, None, 0, None) == -1
