import numpy as np, torch, sys
sys.path.insert(0, '.')
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch_device
dev = torch.device('cuda:0')
B, N = 256, 480000
pcm2d = synth_batch_device(101000, B, N, dev)
fe = HipWhisperFeatureExtractor(80, device=dev)
off = (torch.arange(B, dtype=torch.int64) * N).to(dev)
ln = torch.full((B,), N, dtype=torch.int32, device=dev)
out = fe.extract(pcm2d.reshape(-1), off, ln)
mx = out.amax(dim=(1, 2), keepdim=True)
fl = mx - 2.0
atfloor = (out <= fl + 1e-7)
print('cells at floor: %.4f %%' % (100 * atfloor.float().mean().item()))
t = atfloor[:, :, :2976].reshape(B, 80, 62, 48).any(dim=3).any(dim=1)
print('tiles with clamped cells: %.2f %% ; clips with any: %d of %d' % (100 * t.float().mean().item(), int(t.any(dim=1).sum()), B))
