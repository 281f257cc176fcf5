#!/usr/bin/env python3
"""A/B kernel timing: python3 tools/ab_bench.py lib1.so lib2.so ...  (each in its own process via STTFE_LIB_PATH).
Child mode: python3 tools/ab_bench.py --child [batch] -> prints avg stft_mel / clamp launch ms for several knobs."""
import ctypes, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    for lib in sys.argv[1:]:
        env = dict(os.environ, STTFE_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"== {lib}\n{r.stdout}{r.stderr[-2000:] if r.returncode else ''}", flush=True)
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch
from speech_to_text_amd import _lib
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
B, n_mels = 1024, 80
dev = torch.device("cuda:0")
L = _lib.lib()
fe = HipWhisperFeatureExtractor(n_mels, device=dev)
pcm = synth_batch_device(1, B, 480000, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * 480000).to(dev)
ln = torch.full((B,), 480000, dtype=torch.int32, device=dev)
out = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=dev)
knobs = [int(a) for a in os.environ.get("AB_TPB", "7,9,16,21").split(",")]
for tpb in knobs:
    if hasattr(L, "sttfe_set_whisper_tiles_per_block"):
        L.sttfe_set_whisper_tiles_per_block(tpb)
    # the event pool exists before the settle launches: creating it afterwards idles the GPU for ~2 ms and the
    # clocks drop (the next ~15 launches run up to 10 % slower)
    L.sttfe_timing_enable(256)
    if hasattr(L, "sttfe_timing_set_stride"):
        L.sttfe_timing_set_stride(1 << 30)
    for _ in range(60):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    L.sttfe_timing_reset()
    if hasattr(L, "sttfe_timing_set_stride"):
        L.sttfe_timing_set_stride(1)
    for _ in range(8):
        fe.extract(pcm, off, ln, out=out)
    torch.cuda.synchronize()
    t, n = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_stft_mel", ctypes.byref(t), ctypes.byref(n))
    t12, n12 = ctypes.c_double(), ctypes.c_int()
    if L.sttfe_timing_query(b"whisper_stft_mel12", ctypes.byref(t12), ctypes.byref(n12)) == 0:      # both STFT launches of a call
        t.value += t12.value
    t2, n2 = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(b"whisper_clamp_fill", ctypes.byref(t2), ctypes.byref(n2))
    L.sttfe_timing_enable(0)
    ms = t.value / 8
    print(f"tpb={tpb:3d} stft_mel {ms:.3f} ms/step ({n.value//8} launches)  clamp {t2.value/8:.3f} ms  "
          f"-> {B*2.88e6/ (ms*1e-3)/1e9:.0f} GB/s = {B*2.88e6/(ms*1e-3)/8e12*100:.1f}% of 8 TB/s", flush=True)
