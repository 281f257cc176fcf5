"""Diagnostic for the fused padding fill of whisper_stft_mel_kernel: polls spent waiting for clips (workspace word 2 of
the ticket area) and kernel times on config 4's batch.   python3 tools/dbg_fuse.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_to_text_amd import _lib
from speech_to_text_amd.synth import common_voice_lengths, synth_batch_device
dev = torch.device("cuda:0")
L = _lib.lib()
B, n_mels = 1024, 128
lens = common_voice_lengths(np.random.default_rng(104000), B)
n_max = (int(lens.max()) + 63) // 64 * 64
pcm = synth_batch_device(104000, B, n_max, dev).reshape(-1)
off = (torch.arange(B, dtype=torch.int64) * n_max).to(dev)
ln = torch.tensor(lens, dtype=torch.int32, device=dev)
out = torch.empty((B, n_mels, 3000), dtype=torch.float32, device=dev)
nb = ctypes.c_size_t()
L.sttfe_whisper_tables_bytes(n_mels, ctypes.byref(nb))
host = (ctypes.c_ubyte * nb.value)()
L.sttfe_whisper_tables_build(n_mels, host, nb.value)
tab = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
wb = ctypes.c_size_t()
L.sttfe_whisper_workspace_bytes(B, ctypes.byref(wb))
ws = torch.empty(wb.value, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def run():
    rc = L.sttfe_whisper_logmel(ctypes.c_void_p(pcm.data_ptr()), ctypes.c_void_p(off.data_ptr()), ctypes.c_void_p(ln.data_ptr()), B, n_mels, 3000,
                                ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, 0, ctypes.c_void_p(ws.data_ptr()), wb.value, ctypes.c_void_p(stream))
    assert rc == 0, rc
for _ in range(5):
    run()
torch.cuda.synchronize()
words = ws.view(torch.int32)[B * 64: B * 64 + 16].cpu().numpy()
done = ws.view(torch.int32)[B * 64 + 16: B * 64 + 16 + B].cpu().numpy()
print("ticket area:", words[:4], " fused flag", words[1], " polls", words[2], " done counters min/max", done.min(), done.max())
L.sttfe_timing_enable(64)
for _ in range(10):
    run()
torch.cuda.synchronize()
for k in (b"whisper_stft_mel", b"whisper_clamp_fill"):
    t, n = ctypes.c_double(), ctypes.c_int()
    L.sttfe_timing_query(k, ctypes.byref(t), ctypes.byref(n))
    print(k.decode(), "%.4f ms avg over %d" % (t.value / max(n.value, 1), n.value))
