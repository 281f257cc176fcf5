"""Oracle for row K of SURVEY.md section 8a: Speech2Text Kaldi fbank + utterance CMVN.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what the reference gets from
``feature_extractor(audio_np, sampling_rate=16000, return_tensors="pt", truncation=..., max_length=...)``
at ``src/data.py:346-358``.  The arithmetic is Hugging Face
``Speech2TextFeatureExtractor`` in its NumPy fallback (torchaudio is absent both in
this container and, for the goldens, by construction):

  HF:models/speech_to_text/feature_extraction_speech_to_text.py:90-102   kaldi mel bank, povey window
  HF:...speech_to_text.py:112         x * 2**15
  HF:...speech_to_text.py:124-138     spectrogram(frame 400, hop 160, fft 512, center=False,
                                      preemphasis 0.97, remove_dc_offset, mel_floor eps, log)
  HF:audio_utils.py:953-1017          the framing loop restated here without the Python loop
  HF:...speech_to_text.py:141-163     utterance_cmvn (float32 mean / population std, pad rows <- 0)
  HF:...speech_to_text.py:270-303     truncation to max_length frames, padding, attention_mask

Pinned by tests/golden/s2t_*.npz (outputs of the HF extractor run in this container).
"""
from __future__ import annotations

import numpy as np

SAMPLING_RATE = 16000
FRAME = 400
HOP = 160
FFT = 512
N_BINS = FFT // 2 + 1
MEL_FLOOR = 1.192092955078125e-07


def _hz_to_mel_kaldi(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)   # HF:audio_utils.py:468


def mel_filter_bank_kaldi(n_mels: int = 80, n_bins: int = N_BINS, fmin: float = 20.0,
                          fmax: float = 8000.0, sr: int = SAMPLING_RATE) -> np.ndarray:
    """(n_bins, n_mels) float64; kaldi scale, triangles in mel space, no norm.

    HF:audio_utils.py:700-715 with the arguments of HF:...speech_to_text.py:90-99.
    """
    mel_pts = np.linspace(_hz_to_mel_kaldi(fmin), _hz_to_mel_kaldi(fmax), n_mels + 2)
    bin_width = sr / ((n_bins - 1) * 2)
    fft_mels = _hz_to_mel_kaldi(bin_width * np.arange(n_bins))
    diff = np.diff(mel_pts)
    slopes = mel_pts[None, :] - fft_mels[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def povey_window(n: int = FRAME) -> np.ndarray:
    """HF:audio_utils.py:786-787 with periodic=False: np.hanning(n) ** 0.85."""
    return np.power(np.hanning(n), 0.85)


def num_frames(n_samples: int) -> int:
    """snip-edges frame count (HF:audio_utils.py:963 with center=False)."""
    if n_samples < FRAME:
        return 0
    return 1 + (n_samples - FRAME) // HOP


def kaldi_power(wave: np.ndarray) -> np.ndarray:
    """1-D float32 waveform in [-1, 1] -> (T, 257) float64 power spectrum of the Kaldi frames."""
    x = (np.asarray(wave, dtype=np.float32).reshape(-1) * (2 ** 15)).astype(np.float64)
    T = num_frames(x.shape[0])
    if T == 0:
        return np.zeros((0, N_BINS), dtype=np.float64)
    idx = np.arange(FRAME)[None, :] + HOP * np.arange(T)[:, None]
    buf = x[idx]                                                  # (T, 400) float64
    buf = buf - buf.mean(axis=1, keepdims=True)                   # remove_dc_offset (:979-980)
    pre = buf.copy()
    pre[:, 1:] -= 0.97 * buf[:, :-1]                              # preemphasis (:982-984)
    pre[:, 0] *= 1.0 - 0.97
    pre *= povey_window()[None, :]
    spec = np.fft.rfft(pre, n=FFT, axis=1).astype(np.complex64)   # stored as complex64 (:966)
    return np.abs(spec, dtype=np.float64) ** 2.0                  # (:993)


def kaldi_fbank(wave: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """1-D float32 waveform in [-1, 1] -> (T, n_mels) float32 log-mel (natural log)."""
    power = kaldi_power(wave)
    if power.shape[0] == 0:
        return np.zeros((0, n_mels), dtype=np.float32)
    mel = np.maximum(MEL_FLOOR, power @ mel_filter_bank_kaldi(n_mels))   # (:998)
    return np.log(mel).astype(np.float32)                         # (:1002,1015)


def utterance_cmvn(x: np.ndarray, input_length: int, padding_value: float = 0.0, normalize_means: bool = True,
                   normalize_vars: bool = True) -> np.ndarray:
    """HF:...speech_to_text.py:141-163, same NumPy calls (float32 in, float32 out)."""
    if normalize_means:
        mean = x[:input_length].mean(axis=0)
        x = np.subtract(x, mean)
    if normalize_vars:
        std = x[:input_length].std(axis=0)
        x = np.divide(x, std)
    if input_length < x.shape[0]:
        x[input_length:] = padding_value
    return x.astype(np.float32)


def speech2text_features(clips, n_mels: int = 80, max_length: int | None = None,
                         truncation: bool = False, pad_to: int | None = None, normalize_means: bool = True,
                         normalize_vars: bool = True):
    """List of waveforms -> (features (B, T_pad, n_mels) f32, attention_mask (B, T_pad) int32, lengths).

    ``truncation``/``max_length`` follow HF:...speech_to_text.py:275-283 (truncate the
    FEATURE frames to max_length before CMVN).  With a single clip and no padding the
    reference call (src/data.py:349-355) returns T_pad == T.  For a list the frames are
    right-padded with zeros to the longest clip (or ``pad_to``), which is what the
    reference's collator does afterwards (src/data.py:468-487).
    """
    feats = [kaldi_fbank(c, n_mels) for c in clips]
    if truncation and max_length is not None:
        feats = [f[:max_length] for f in feats]
    lengths = [f.shape[0] for f in feats]
    t_pad = max(lengths) if pad_to is None else pad_to
    out = np.zeros((len(feats), t_pad, n_mels), dtype=np.float32)
    mask = np.zeros((len(feats), t_pad), dtype=np.int32)
    for i, (f, n) in enumerate(zip(feats, lengths)):
        if n > 0:
            out[i, :n] = utterance_cmvn(f, n, normalize_means=normalize_means, normalize_vars=normalize_vars)
        mask[i, :n] = 1
    return out, mask, np.asarray(lengths, dtype=np.int32)
