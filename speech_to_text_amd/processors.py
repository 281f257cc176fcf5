"""Feature extractors with the Hugging Face call shape, running on libsttfe.so (MI355X).

Drop-in for ``processor.feature_extractor`` in the reference (``src/data.py:338-358``,
``inference.py:148-153``) and for ``TorchWhisperFeatureExtractor`` (``src/processors.py:24-246``):

    fe = HipWhisperFeatureExtractor(feature_size=80)
    feats = fe(raw_speech, sampling_rate=16000, return_tensors="pt").input_features   # (B, 80, 3000)

Numerics follow the HF extractors (the parity target of BASELINE.json), NOT the HTK-scale mel of
``src/processors.py:75-83``.  Results stay on the GPU unless ``return_tensors="np"``.
There is no CPU path: without ``libsttfe.so`` or without a GPU these classes raise.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib


class BatchFeature(dict):
    """dict with attribute access, like transformers.BatchFeature (``.input_features``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, *args, **kwargs):
        return BatchFeature({k: (v.to(*args, **kwargs) if torch.is_tensor(v) else v) for k, v in self.items()})


def _as_clip_list(raw_speech) -> List:
    """Normalise the accepted input forms to a list of 1-D arrays / tensors.

    Accepted (src/processors.py:150,169-176; HF:...whisper.py:278-294): 1-D np.ndarray, list of floats,
    list of 1-D arrays, 2-D np.ndarray (batch, n), list of torch tensors on any device
    (multi-channel tensors are averaged to mono, src/processors.py:94-96).
    """
    if torch.is_tensor(raw_speech):
        raw_speech = [raw_speech] if raw_speech.dim() == 1 else list(raw_speech)
    elif isinstance(raw_speech, np.ndarray):
        if raw_speech.ndim > 2:
            raise ValueError("Only mono-channel audio is supported")
        raw_speech = [raw_speech] if raw_speech.ndim == 1 else list(raw_speech)
    elif isinstance(raw_speech, (list, tuple)):
        if len(raw_speech) == 0:
            return []
        if not isinstance(raw_speech[0], (np.ndarray, list, tuple)) and not torch.is_tensor(raw_speech[0]):
            raw_speech = [np.asarray(raw_speech, dtype=np.float32)]
    else:
        raise TypeError(f"unsupported raw_speech type {type(raw_speech)}")
    return list(raw_speech)


def pack_clips(clips: Sequence, device: torch.device, max_samples: Optional[int] = None):
    """Concatenate clips into one device float32 buffer -> (pcm, offsets int64, lengths int32) on device.

    Host arrays go through one pinned staging buffer and one async H2D copy; device tensors are
    copied device-to-device.
    """
    lens, host_idx = [], []
    for i, c in enumerate(clips):
        if torch.is_tensor(c):
            if c.dim() > 1:
                c = c.float().mean(dim=0)
                clips[i] = c
            n = int(c.shape[-1])
        else:
            c = np.asarray(c, dtype=np.float32).reshape(-1)
            clips[i] = c
            n = c.shape[0]
            host_idx.append(i)
        lens.append(n if max_samples is None else min(n, max_samples))
    # 64-sample alignment keeps every clip 256-byte aligned; 448 samples of slack after the last
    # clip are never read (kernels bound every load by lengths[b]) but keep torch's allocator honest.
    starts, total = [], 0
    for n in lens:
        starts.append(total)
        total += (n + 63) // 64 * 64
    pcm = torch.empty(total + 64, dtype=torch.float32, device=device)
    if host_idx:
        stage = torch.empty(total + 64, dtype=torch.float32, pin_memory=True)
        sn = stage.numpy()
        for i in host_idx:
            sn[starts[i]:starts[i] + lens[i]] = clips[i][:lens[i]]
        if len(host_idx) == len(clips):
            pcm.copy_(stage, non_blocking=True)
        else:
            for i in host_idx:
                pcm[starts[i]:starts[i] + lens[i]].copy_(stage[starts[i]:starts[i] + lens[i]], non_blocking=True)
    for i, c in enumerate(clips):
        if torch.is_tensor(c):
            pcm[starts[i]:starts[i] + lens[i]].copy_(c.reshape(-1)[:lens[i]].to(torch.float32))
    offsets = torch.tensor(starts, dtype=torch.int64).to(device, non_blocking=True)
    lengths = torch.tensor(lens, dtype=torch.int32).to(device, non_blocking=True)
    return pcm, offsets, lengths


def _resolve_device(device) -> torch.device:
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.SttfeError("no GPU visible: the sttfe extractors have no CPU path")
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SttfeError(f"device {device} is not a GPU: the sttfe extractors have no CPU path")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class HipWhisperFeatureExtractor:
    """Whisper log-mel extractor (HF ``WhisperFeatureExtractor`` call shape) on MI355X."""

    model_input_names = ["input_features"]

    def __init__(self, feature_size: int = 80, sampling_rate: int = 16000, hop_length: int = 160,
                 chunk_length: int = 30, n_fft: int = 400, padding_value: float = 0.0,
                 device: Optional[Union[str, torch.device]] = None, **kwargs):
        if (sampling_rate, hop_length, n_fft, chunk_length) != (16000, 160, 400, 30):
            raise ValueError("the HIP kernels implement sampling_rate=16000, hop_length=160, n_fft=400, "
                             "chunk_length=30 (the Whisper preprocessor configuration)")
        if not 1 <= feature_size <= 128:
            raise ValueError("feature_size must be in [1, 128]")
        self.feature_size = feature_size
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.n_fft = n_fft
        self.win_length = n_fft
        self.chunk_length = chunk_length
        self.padding_value = padding_value
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.expected_frames = self.nb_max_frames
        self._device_arg = device
        self._tables = {}          # device index -> uint8 tensor (tables blob)
        self._host_tables = None

    # the device tensors are rebuilt lazily, so instances pickle (DataLoader workers, tests/test_dataloader.py:376-382)
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_tables"] = {}
        return d

    @property
    def device(self) -> torch.device:
        return _resolve_device(self._device_arg)

    @property
    def mel_filters(self) -> np.ndarray:
        """(201, feature_size) float32 bank, same orientation as HF's ``mel_filters``."""
        dense = np.empty((self.feature_size, 201), dtype=np.float32)
        _lib.check(_lib.lib().sttfe_whisper_mel_bank(self.feature_size, dense.ctypes.data_as(ctypes.c_void_p)))
        return dense.T.copy()

    def _tables_on(self, device: torch.device) -> torch.Tensor:
        t = self._tables.get(device.index)
        if t is None:
            if self._host_tables is None:
                L = _lib.lib()
                n = ctypes.c_size_t()
                _lib.check(L.sttfe_whisper_tables_bytes(self.feature_size, ctypes.byref(n)))
                host = torch.empty(n.value, dtype=torch.uint8)
                _lib.check(L.sttfe_whisper_tables_build(self.feature_size, _lib.ptr(host), n.value))
                self._host_tables = host
            t = self._host_tables.to(device)
            self._tables[device.index] = t
        return t

    def extract(self, pcm: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor,
                mask_table: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                clip_scale: Optional[torch.Tensor] = None, n_frames: Optional[int] = None,
                host_lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """Device-resident entry: packed PCM + per-clip offsets/lengths -> (B, M, 3000) float32.

        ``n_frames`` (a multiple of 4, default 3000): the COMPACT output contract for callers that keep "frames + padding
        constant" instead of the 30 s tensor (the offline feature cache): only the first ``n_frames``
        columns are produced.  They are bit-identical to the same columns of the 3000-frame result as long as every
        clip's audio ends before them -- ``n_frames >= whisper_compact_frames(lengths)``, which the caller guarantees
        (it knows the lengths on the host; the last 4 columns are then padding for every clip, so ``out[b, 0, -1]`` is
        clip b's padding constant).  With a shorter ``n_frames`` the columns are still those of the 30 s result (the
        kernel keeps reading the clip's real samples and reflects at 30 s, not at ``160 * n_frames``) but the last
        columns are no longer padding and ``out[b, 0, -1]`` is not the padding constant: pass ``host_lengths`` (the
        clip lengths, known on the host) to have the bound checked here -- it cannot be checked on the device without
        a synchronisation.

        ``clip_scale`` (device float32 (B,)): the features of ``clip_scale[b] * clip_b`` without touching the
        samples (``sttfe_whisper_logmel_scaled``) -- the peak normalisation of ``GpuAudioPreprocessor.run(...,
        defer_normalize=True)`` folded into the kernel's epilogue."""
        device = pcm.device
        B = int(offsets.shape[0])
        L = _lib.lib()
        if n_frames is None:
            n_frames = int(out.shape[2]) if (out is not None and out.dim() == 3) else self.nb_max_frames
        T = int(n_frames)
        if T < 4 or T > self.nb_max_frames or T % 4:
            raise ValueError(f"n_frames must be a multiple of 4 in [4, {self.nb_max_frames}], got {n_frames}")
        if host_lengths is not None and T < self.nb_max_frames:
            from .feature_cache import whisper_compact_frames
            need = whisper_compact_frames(host_lengths, self.nb_max_frames)
            if need > T:
                raise ValueError(f"compact output of {T} frames is too short for the longest clip (needs {need}): "
                                 "its last columns would not be padding")
        if out is None:
            out = torch.empty((B, self.feature_size, T), dtype=torch.float32, device=device)
        elif tuple(out.shape) != (B, self.feature_size, T) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError(f"out must be a contiguous float32 tensor of shape {(B, self.feature_size, T)}")
        ws_bytes = ctypes.c_size_t()
        _lib.check(L.sttfe_whisper_workspace_bytes(B, ctypes.byref(ws_bytes)))
        ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=device)
        n_masks = 0
        if mask_table is not None:
            assert mask_table.dtype == torch.int32 and mask_table.dim() == 3 and mask_table.shape[2] == 3
            mask_table = mask_table.to(device).contiguous()
            n_masks = int(mask_table.shape[1])
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            if clip_scale is None:
                _lib.check(L.sttfe_whisper_logmel(_lib.ptr(pcm), _lib.ptr(offsets), _lib.ptr(lengths), B,
                                                  self.feature_size, T,
                                                  _lib.ptr(self._tables_on(device)), _lib.ptr(out),
                                                  _lib.ptr(mask_table), n_masks, _lib.ptr(ws), ws.numel(),
                                                  ctypes.c_void_p(stream)), "sttfe_whisper_logmel")
            else:
                if clip_scale.dtype != torch.float32 or clip_scale.numel() != B or clip_scale.device != device:
                    raise ValueError("clip_scale must be a float32 tensor of B elements on the PCM's device")
                clip_scale = clip_scale.contiguous()
                _lib.check(L.sttfe_whisper_logmel_scaled(_lib.ptr(pcm), _lib.ptr(offsets), _lib.ptr(lengths), B,
                                                         self.feature_size, T,
                                                         _lib.ptr(self._tables_on(device)), _lib.ptr(out),
                                                         _lib.ptr(mask_table), n_masks, _lib.ptr(clip_scale),
                                                         _lib.ptr(ws), ws.numel(), ctypes.c_void_p(stream)),
                           "sttfe_whisper_logmel_scaled")
        return out

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "pt",
                 padding=True, truncation: bool = True, max_length: Optional[int] = None,
                 mask_table: Optional[torch.Tensor] = None, **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a "
                f"sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input was "
                f"sampled with {self.sampling_rate} and not {sampling_rate}.")
        clips = _as_clip_list(raw_speech)
        device = self.device
        if not clips:
            feats = torch.empty((0, self.feature_size, self.nb_max_frames), dtype=torch.float32, device=device)
        else:
            pcm, offsets, lengths = pack_clips(clips, device, self.n_samples)
            feats = self.extract(pcm, offsets, lengths, mask_table)
        if return_tensors == "np":
            feats = feats.cpu().numpy()
        return BatchFeature({"input_features": feats})


class TorchWhisperFeatureExtractor(HipWhisperFeatureExtractor):
    """Name/signature-compatible stand-in for ``src/processors.py:TorchWhisperFeatureExtractor``.

    Same constructor arguments and ``__call__(raw_speech, sampling_rate, return_tensors, padding)``;
    returns a ``SimpleNamespace(input_features=...)`` like the reference (src/processors.py:241-246)
    and raises ``ValueError`` on a wrong sampling rate (src/processors.py:165-166).  Both branches keep the
    reference class's own semantics (``sttfe_whisper_logmel_torch``), which differ from HF's:

    * ``padding=True`` (src/processors.py:180-221): pad / truncate to 30 s, per-clip max over all 3001 STFT frames,
      then crop to 3000 (HF drops the last frame before taking the max);
    * ``padding=False`` (:222-239): no padding -- reflect at the clip's own end, ``1 + n // 160`` frames, per-clip
      max over them, crop to 3000, pad to the longest clip of the batch with ``padding_value``.

    One deliberate difference: the mel bank is HF's (slaney scale), the parity target of this build, not the
    HTK-scale bank of src/processors.py:75-83.  ``padding=False`` needs clips of more than 200 samples (torch.stft's
    reflect padding raises below that, and so does this class); clips longer than 30 s are transformed to their own
    end for the per-clip maximum and cropped to 3000 frames afterwards, as in the reference.
    """

    def extract_torch(self, pcm: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor, n_frames: int,
                      variable_length: bool) -> torch.Tensor:
        """Device-resident entry -> (B, M, n_frames) float32 with the reference class's semantics."""
        device = pcm.device
        B = int(offsets.shape[0])
        L = _lib.lib()
        out = torch.empty((B, self.feature_size, n_frames), dtype=torch.float32, device=device)
        ws_bytes = ctypes.c_size_t()
        _lib.check(L.sttfe_whisper_workspace_bytes(B, ctypes.byref(ws_bytes)))
        ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            _lib.check(L.sttfe_whisper_logmel_torch(_lib.ptr(pcm), _lib.ptr(offsets), _lib.ptr(lengths), B,
                                                    self.feature_size, n_frames, _lib.ptr(self._tables_on(device)),
                                                    _lib.ptr(out), None, 0, 1 if variable_length else 0,
                                                    ctypes.c_float(self.padding_value), _lib.ptr(ws), ws.numel(),
                                                    ctypes.c_void_p(stream)), "sttfe_whisper_logmel_torch")
        return out

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "pt",
                 padding: Optional[bool] = True):
        if sampling_rate is None:
            sampling_rate = self.sampling_rate
        if sampling_rate != self.sampling_rate:
            raise ValueError(f"Expected sampling_rate={self.sampling_rate}, got {sampling_rate}. Resample beforehand.")
        clips = _as_clip_list(raw_speech)
        device = self.device
        if not clips:
            feats = torch.empty((0, self.feature_size, self.nb_max_frames if padding else 0), dtype=torch.float32,
                                device=device)
        elif padding:
            pcm, offsets, lengths = pack_clips(clips, device, self.n_samples + self.hop_length)
            feats = self.extract_torch(pcm, offsets, lengths, self.nb_max_frames, False)
        else:
            lens = [int(c.shape[-1]) if torch.is_tensor(c) else int(np.asarray(c).reshape(-1).shape[0]) for c in clips]
            if min(lens) <= self.n_fft // 2:
                raise ValueError(f"padding=False needs clips longer than {self.n_fft // 2} samples (reflect padding of "
                                 f"the STFT); got {min(lens)}")
            # clips longer than 30 s: the reference transforms the whole clip, takes the maximum over ALL its frames and
            # crops to 3000 afterwards (src/processors.py:226-239); so does the kernel (tiles behind the output only
            # feed the per-clip maximum)
            frames = min(max(1 + n // self.hop_length for n in lens), self.expected_frames)
            pcm, offsets, lengths = pack_clips(clips, device, max(lens))
            feats = self.extract_torch(pcm, offsets, lengths, (frames + 3) // 4 * 4, True)[:, :, :frames]
        if return_tensors != "pt":
            feats = feats.detach().cpu().numpy()
        return SimpleNamespace(input_features=feats)


class HipSpeech2TextFeatureExtractor:
    """Kaldi fbank + utterance CMVN extractor (HF ``Speech2TextFeatureExtractor`` call shape) on MI355X.

    ``fe(raw_speech, sampling_rate=16000, return_tensors="pt", truncation=True, max_length=N)`` ->
    ``.input_features`` (B, T, 80) float32 and ``.attention_mask`` (B, T) int32, as used at
    ``src/data.py:346-358``.  A batch is zero-padded to its longest clip (what the reference's collator
    does afterwards, ``src/data.py:468-487``).
    """

    model_input_names = ["input_features", "attention_mask"]

    def __init__(self, feature_size: int = 80, sampling_rate: int = 16000, num_mel_bins: int = 80,
                 padding_value: float = 0.0, do_ceptral_normalize: bool = True, normalize_means: bool = True,
                 normalize_vars: bool = True, device: Optional[Union[str, torch.device]] = None,
                 stft_precision: str = "float64", **kwargs):
        if sampling_rate != 16000:
            raise ValueError("the HIP kernels implement sampling_rate=16000")
        if not 1 <= num_mel_bins <= 128:
            raise ValueError("num_mel_bins must be in [1, 128]")
        if stft_precision not in ("float64", "float32"):
            raise ValueError("stft_precision must be 'float64' (HF NumPy-path arithmetic, the default) or 'float32' "
                             "(torchaudio-Kaldi-path arithmetic, faster)")
        self.feature_size = feature_size
        self.num_mel_bins = num_mel_bins
        self.sampling_rate = sampling_rate
        self.padding_value = padding_value
        self.do_ceptral_normalize = bool(do_ceptral_normalize)     # False: raw ln(mel), HF:...speech_to_text.py:256-262
        self.normalize_means = bool(normalize_means)     # individually, as HF's utterance_cmvn (:141-163)
        self.normalize_vars = bool(normalize_vars)
        self.stft_precision = stft_precision
        self.return_attention_mask = True
        self.n_fft = 512
        self.hop_length = 160
        self.win_length = 400
        self._device_arg = device
        self._tables = {}
        self._host_tables = None

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_tables"] = {}
        return d

    @property
    def device(self) -> torch.device:
        return _resolve_device(self._device_arg)

    @property
    def mel_filters(self) -> np.ndarray:
        dense = np.empty((self.num_mel_bins, 257), dtype=np.float32)
        _lib.check(_lib.lib().sttfe_kaldi_mel_bank(self.num_mel_bins, dense.ctypes.data_as(ctypes.c_void_p)))
        return dense.T.copy()

    @staticmethod
    def num_frames(n_samples: int, max_length: Optional[int] = None) -> int:
        t = 1 + (n_samples - 400) // 160 if n_samples >= 400 else 0
        return min(t, max_length) if max_length else t

    def _tables_on(self, device: torch.device) -> torch.Tensor:
        t = self._tables.get(device.index)
        if t is None:
            if self._host_tables is None:
                L = _lib.lib()
                n = ctypes.c_size_t()
                _lib.check(L.sttfe_kaldi_tables_bytes(self.num_mel_bins, ctypes.byref(n)))
                host = torch.empty(n.value, dtype=torch.uint8)
                _lib.check(L.sttfe_kaldi_tables_build(self.num_mel_bins, _lib.ptr(host), n.value))
                self._host_tables = host
            t = self._host_tables.to(device)
            self._tables[device.index] = t
        return t

    def extract(self, pcm: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor, t_max: int,
                max_frames: int = 0, mask_table: Optional[torch.Tensor] = None,
                attention_dtype: torch.dtype = torch.int32, apply_cmvn: Optional[bool] = None,
                stft_precision: Optional[str] = None):
        """Device-resident entry -> (features (B, t_max, M) f32, attention_mask (B, t_max)).

        ``apply_cmvn`` defaults to the extractor's ``do_ceptral_normalize``; ``stft_precision`` to the extractor's
        (``"float64"``: STTFE_KALDI_F64, max-abs <= 1e-4 vs the HF NumPy path; ``"float32"``: the faster kernel)."""
        if apply_cmvn is None:
            apply_cmvn = self.do_ceptral_normalize
        flags = (0 if apply_cmvn else 1) | (2 if (stft_precision or self.stft_precision) == "float64" else 0)
        flags |= (0 if self.normalize_means else 4) | (0 if self.normalize_vars else 8)     # STTFE_KALDI_NO_MEAN / _NO_VAR
        device = pcm.device
        B = int(offsets.shape[0])
        L = _lib.lib()
        out = torch.empty((B, t_max, self.num_mel_bins), dtype=torch.float32, device=device)
        attn = torch.empty((B, t_max), dtype=attention_dtype, device=device)
        ws_bytes = ctypes.c_size_t()
        _lib.check(L.sttfe_kaldi_workspace_bytes(B, self.num_mel_bins, ctypes.byref(ws_bytes)))
        ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=device)
        n_masks = 0
        if mask_table is not None:
            assert mask_table.dtype == torch.int32 and mask_table.dim() == 3 and mask_table.shape[2] == 3
            mask_table = mask_table.to(device).contiguous()
            n_masks = int(mask_table.shape[1])
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            _lib.check(L.sttfe_kaldi_fbank_cmvn(_lib.ptr(pcm), _lib.ptr(offsets), _lib.ptr(lengths), B,
                                                self.num_mel_bins, t_max, int(max_frames or 0), flags,
                                                _lib.ptr(self._tables_on(device)), _lib.ptr(out), _lib.ptr(attn),
                                                attn.element_size(), _lib.ptr(mask_table), n_masks,
                                                _lib.ptr(ws), ws.numel(), ctypes.c_void_p(stream)),
                       "sttfe_kaldi_fbank_cmvn")
        return out, attn

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "pt",
                 padding=True, truncation: bool = False, max_length: Optional[int] = None,
                 mask_table: Optional[torch.Tensor] = None, **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a "
                f"sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input was "
                f"sampled with {self.sampling_rate} and not {sampling_rate}.")
        clips = _as_clip_list(raw_speech)
        device = self.device
        max_frames = int(max_length) if (truncation and max_length) else 0
        if not clips:
            feats = torch.empty((0, 0, self.num_mel_bins), dtype=torch.float32, device=device)
            attn = torch.empty((0, 0), dtype=torch.int32, device=device)
        else:
            pcm, offsets, lengths = pack_clips(clips, device)
            t_max = max(self.num_frames(int(c.shape[-1]), max_frames or None) for c in clips)
            feats, attn = self.extract(pcm, offsets, lengths, t_max, max_frames, mask_table)
        if return_tensors == "np":
            feats, attn = feats.cpu().numpy(), attn.cpu().numpy()
        return BatchFeature({"input_features": feats, "attention_mask": attn})
