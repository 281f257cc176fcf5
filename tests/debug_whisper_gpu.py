# manual GPU debugging helper (not collected by pytest)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_to_text_amd.processors import HipWhisperFeatureExtractor
from speech_to_text_amd.synth import synth_batch
from oracle.whisper import whisper_features
for n_mels in (80, 128):
    fe = HipWhisperFeatureExtractor(n_mels, device="cuda:0")
    lengths = [480000, 82400, 16000, 199, 1, 560]
    clips = synth_batch(101, lengths)
    got = fe(clips, sampling_rate=16000).input_features.cpu().numpy()
    exp = whisper_features(clips, n_mels)
    d = np.abs(got - exp)
    print(n_mels, "max err per clip", d.reshape(len(clips), -1).max(axis=1))
    i = np.unravel_index(np.argmax(d), d.shape)
    print("worst at", i, got[i], exp[i])
    print("nan", np.isnan(got).sum())
    print("err by frame block (clip0):", d[0].max(axis=0)[:2880].reshape(-1, 240).max(axis=1))
    print("err by mel (clip0):", d[0].max(axis=1)[::8])
