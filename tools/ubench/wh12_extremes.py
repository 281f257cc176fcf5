#!/usr/bin/env python3
"""12-wave kernel (band-sparse VALU projection) against the general kernel (MFMA projection) on amplitude extremes:
signals scaled from 1e-22 to 1e15, silence, impulses, DC, with and without the per-clip scale of
sttfe_whisper_logmel_scaled -- where do the two stop being bit-identical (denormal products, overflow)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor, pack_clips
from speech_to_text_amd.synth import synth_batch
L = _lib.lib()
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
for n_mels in (80, 128):
    fe = HipWhisperFeatureExtractor(n_mels, device=dev)
    for amp in (1e-22, 1e-20, 1e-19, 3e-19, 1e-18, 1e-17, 1e-15, 1e-10, 1e-5, 1.0, 1e5, 1e10, 1e15, 1e17, 1e19):
        lens = [48000, 31000, 16000, 7481, 160000]
        clips = [np.asarray(c, np.float32) * np.float32(amp) for c in synth_batch(3, lens)]
        clips.append(np.zeros(20000, np.float32))
        imp = np.zeros(20000, np.float32); imp[777] = amp; clips.append(imp)
        clips.append(np.full(20000, amp, np.float32))
        pcm, off, ln = pack_clips(clips, dev, 160000)
        for sc in (None, 1.0 / amp):
            if sc is not None and not (1e-18 < sc < 1e18):
                continue
            scale = None if sc is None else torch.full((len(clips),), sc, dtype=torch.float32, device=dev)
            res = []
            for on in (0, 1):
                assert L.sttfe_set_whisper_uniform_kernel(on) == 0
                out = torch.full((len(clips), n_mels, 1000), -3.0, dtype=torch.float32, device=dev)
                fe.extract(pcm, off, ln, out=out, clip_scale=scale)
                torch.cuda.synchronize()
                res.append(out)
            same = torch.equal(res[0].view(torch.int32), res[1].view(torch.int32))
            nbad = int((res[0].view(torch.int32) != res[1].view(torch.int32)).sum())
            print(f"n_mels {n_mels} amp {amp:g} scale {sc}: {'identical' if same else 'DIFFERENT in %d cells' % nbad}"
                  f" finite {bool(torch.isfinite(res[1]).all())} min {float(res[1].min()):.4f} max {float(res[1].max()):.4f}", flush=True)
L.sttfe_set_whisper_uniform_kernel(1)
