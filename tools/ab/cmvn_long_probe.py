#!/usr/bin/env python3
"""kaldi_cmvn time on batches of longer clips for the three float64 CMVN forms (register columns / LDS panel / split
pair): 128 clips of 10-20 s (t_max ~ 2000 frames) and 64 clips of 20-30 s (t_max ~ 3000).  One process per knob."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np, torch
    from speech_to_text_amd import _lib
    from speech_to_text_amd.processors import HipSpeech2TextFeatureExtractor
    from speech_to_text_amd.synth import synth_batch_device
    dev = torch.device("cuda:0")
    L = _lib.lib()
    fe = HipSpeech2TextFeatureExtractor(device=dev)
    for B, lo, hi in ((256, 32000, 160000), (128, 160000, 320000), (64, 320000, 480000)):
        lens = np.random.default_rng(5).integers(lo, hi + 1, size=B)
        n_max = int(lens.max())
        pcm = synth_batch_device(9, B, n_max, dev).reshape(-1)
        off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
        ln = torch.tensor(lens, dtype=torch.int32, device=dev)
        t_max = max(fe.num_frames(int(n)) for n in lens)
        L.sttfe_timing_enable(256)
        L.sttfe_timing_set_stride(1 << 30)
        for _ in range(60):
            fe.extract(pcm, off, ln, t_max)
        torch.cuda.synchronize()
        L.sttfe_timing_reset()
        L.sttfe_timing_set_stride(1)
        for _ in range(10):
            fe.extract(pcm, off, ln, t_max)
        torch.cuda.synchronize()
        t, k = ctypes.c_double(), ctypes.c_int()
        res = []
        for name in (b"kaldi_fbank", b"kaldi_cmvn"):
            L.sttfe_timing_query(name, ctypes.byref(t), ctypes.byref(k))
            res.append(t.value / max(k.value, 1))
        L.sttfe_timing_enable(0)
        print(f"  {B:4d} clips, t_max {t_max:5d}: fbank64 {res[0]:.4f} ms  cmvn {res[1]:.4f} ms", flush=True)
else:
    for name, env in (("register columns", {}), ("LDS panel (<= 2048 frames, else split)", {"STTFE_KALDI_LDS_CMVN": "1"}),
                      ("split pair", {"STTFE_KALDI_SPLIT_CMVN": "1"})):
        print(name, flush=True)
        rc = subprocess.call([sys.executable, __file__, "child"], env=dict(os.environ, **env))
        if rc:
            sys.exit(rc)
