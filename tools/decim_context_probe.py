#!/usr/bin/env python3
"""What a memory-bound kernel pays for its PREDECESSOR: the 48 -> 16 kHz decimator (config 3's clips) and a plain
2 : 1 streaming add of the same bytes, each timed (events around the one launch) alone back to back, behind a kernel that
leaves ~490 MB of freshly written lines (what the previous step's feature kernels leave), and behind a read-only pass
over the same bytes.    python3 tools/decim_context_probe.py [int16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_to_text_amd.preprocess import GpuAudioPreprocessor
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device

dev = torch.device("cuda:0")
B = 512
lens = common_voice_lengths(np.random.default_rng(103000), B) * 3
n_max = (int(lens.max()) + 63) // 64 * 64
pcm = synth_batch_device(103000, B, n_max, dev, sample_rate=48000).reshape(-1)
if len(sys.argv) > 1 and sys.argv[1] == "int16":
    pcm = torch.clamp(torch.round(pcm * 32767.0), -32768, 32767).to(torch.int16)
off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
host = [int(v) for v in lens]
pre = GpuAudioPreprocessor(normalize=False, trim_silence=False, min_duration=0.0, max_duration=1e9, device=dev)
n_in = int(lens.sum())
n_out = int(((lens + 2) // 3).sum())
dec_bytes = pcm.element_size() * n_in + 4 * n_out
n = n_out
x = torch.randn(2 * n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
big = torch.empty(512 * 80 * 3000, dtype=torch.float32, device=dev)            # config 3's feature tensor: 491 MB


def decim():
    pre.run(pcm, off, ln, host, 1, 48000, sync_lengths=False)


def add():
    torch.add(x[:n], x[n:], out=y)


def timed(fn, before):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(25):
        torch.cuda._sleep(3_000_000)          # the host queues everything below while the GPU spins: no host time in the events
        if before is not None:
            before()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


contexts = [("behind an idle spin (nothing dirty, caches as the last launch left them)", None), ("behind itself", "self"), ("behind a 491 MB fill (dirty lines)", lambda: big.fill_(1.0)),
            ("behind a 491 MB read-only sum", lambda: big.sum())]
for name, fn, nbytes in (("decimator 3:1", decim, dec_bytes), ("torch add 2:1", add, 12 * n)):
    for cname, before in contexts:
        ms = timed(fn, fn if before == "self" else before)
        print(f"{name} ({nbytes / 1e6:.0f} MB) {cname:40s}: {ms:.4f} ms  {nbytes / ms / 1e9:.2f} TB/s", flush=True)
