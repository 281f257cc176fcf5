"""Collator + GPU front-end: the reference's ``AudioCollator`` contract with its "cuda" mode completed.

``AudioCollator`` keeps the constructor, call signature, return keys, dtypes, padding values and error
behaviour of the reference's class (``src/data.py:417-529``; pinned by the reference's
``tests/test_dataloader.py:525-845``).  In the reference the ``"cuda"`` branch only pads raw audio
(``src/data.py:519-527``) and nothing consumes that batch (``train.py:291`` indexes
``batch['input_features']``); ``GpuFrontEnd`` is that missing consumer: it turns the raw-audio batch
into the dict the ``"cpu"`` branch returns -- ``input_features`` / ``labels`` / ``text``
(/ ``attention_mask``) -- with every array operation in HIP kernels (``libsttfe.so``).

Usage in the reference's training loop (``train.py:1449-1464``, ``train.py:288-297``)::

    collator = AudioCollator(config.data, model_type, transpose_features)      # device = "cuda"
    loader = DataLoader(dataset, batch_size, collate_fn=collator, num_workers=4)    # workers: decode only
    front = GpuFrontEnd(config.data, model_type, transpose_features, tokenizer=processor.tokenizer,
                        split="train")
    for raw in loader:
        batch = front(raw)                      # main process, GPU
        model(input_features=batch["input_features"], labels=batch["labels"], ...)

DataLoader workers must not touch the GPU, so ``AudioCollator`` stays host-only and picklable; the GPU
work happens in the main process.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from .augment import GpuWaveformAugmenter, SpecAugmentSampler, WaveformAugmentSampler
from .processors import (BatchFeature, HipSpeech2TextFeatureExtractor, HipWhisperFeatureExtractor,
                         _resolve_device)

logger = logging.getLogger(__name__)


class AudioCollator:
    """Drop-in for the reference's ``AudioCollator`` (``src/data.py:417-529``)."""

    def __init__(self, config: Any, model_type: str = "whisper", transpose_features: bool = False):
        self.config = config
        self.device = config.processing_and_augmentation_device            # 'cpu' or 'cuda'
        self.model_type = model_type.lower()
        self.transpose_features = transpose_features
        logger.info(f"Initialized AudioCollator with device: {self.device}, model_type: {self.model_type}, "
                    f"labels padding: -100, transpose_features: {self.transpose_features}")

    def __call__(self, batch: List[Optional[Dict[str, Any]]]) -> Optional[Dict[str, Any]]:
        batch = [sample for sample in batch if sample is not None]          # src/data.py:455
        if not batch:
            logger.info("Empty batch after filtering None samples, skipping...")   # src/data.py:459
            return None

        if self.device == "cpu":
            # samples hold (freq, time) features of any length (src/data.py:468-477)
            feats = [torch.as_tensor(s["input_features"]).T for s in batch]
            time_lengths = [f.shape[0] for f in feats]
            input_features = pad_sequence(feats, batch_first=True).transpose(1, 2)    # (B, freq, T_max)
            attention_mask = None
            if self.model_type == "speech2text":                           # src/data.py:480-487
                max_len = input_features.shape[2]
                attention_mask = torch.ones(len(batch), max_len, dtype=torch.long)
                for i, length in enumerate(time_lengths):
                    if length < max_len:
                        attention_mask[i, length:] = 0
            if self.transpose_features:                                    # src/data.py:493-496
                input_features = input_features.transpose(1, 2)
            labels = pad_sequence([torch.as_tensor(s["labels"]) for s in batch], batch_first=True,
                                  padding_value=-100)                      # src/data.py:500
            result = {"input_features": input_features, "labels": labels,
                      "text": [s.get("text", "") for s in batch]}
            if attention_mask is not None:
                result["attention_mask"] = attention_mask
            return result
        elif self.device == "cuda":
            # src/data.py:519-527, plus "audio_lengths": zero padding hides where a clip ends, and the
            # Speech2Text frame count depends on it (a key the reference's batch lacks; extra keys are
            # ignored by its consumers).
            clips = [torch.as_tensor(s["audio"]).reshape(-1) for s in batch]
            audio = pad_sequence(clips, batch_first=True).unsqueeze(1)
            return {"audio": audio, "texts": [s["text"] for s in batch],
                    "audio_lengths": torch.tensor([c.shape[0] for c in clips], dtype=torch.int32)}
        else:
            raise ValueError(f"Unsupported processing device: {self.device}. Must be 'cpu' or 'cuda'.")


class GpuFrontEnd:
    """Raw-audio batch (``AudioCollator`` "cuda" mode) -> model inputs, on the GPU.

    Produces what the reference's CPU path produces per clip in ``CommonVoiceDataset.__getitem__``
    (``src/data.py:331-380``: feature extraction, SpecAugment on the train split, tokenisation) followed
    by the collator's "cpu" branch (``src/data.py:462-518``: dynamic time padding, attention mask,
    optional transpose, labels padded with -100), but batched: one feature kernel pass with the masks,
    padding, attention mask and (for Speech2Text) the time-major layout fused in.

    ``tokenizer`` (optional) is the HF tokenizer of the processor; tokenisation stays on the host.
    """

    def __init__(self, config: Any, model_type: str = "whisper", transpose_features: bool = False,
                 tokenizer: Any = None, split: str = "train", feature_size: Optional[int] = None,
                 device: Optional[Union[str, torch.device]] = None,
                 max_source_positions: Optional[int] = None, max_target_positions: Optional[int] = None,
                 noise_source: str = "device", pitch_mode: str = "as_constructed"):
        self.config = config
        self.model_type = model_type.lower()
        if self.model_type not in ("whisper", "speech2text"):
            raise ValueError(f"Unsupported model type: {model_type}")
        self.transpose_features = transpose_features
        self.tokenizer = tokenizer
        self.split = split
        self.sample_rate = int(getattr(config, "sample_rate", 16000))
        self._device_arg = device
        aug = getattr(config, "augmentation", None)
        # src/data.py:222: augmentation object only when augmentation.enabled; :360 train split only
        enabled = aug is not None and getattr(aug, "enabled", False)
        self.sampler = SpecAugmentSampler(aug) if enabled else None
        # src/data.py:328-329: waveform-domain augmentations sit between the preprocessor and the extractor
        self.wave_sampler = WaveformAugmentSampler(aug, self.sample_rate, noise_source, pitch_mode) if enabled else None
        self.wave_augmenter = GpuWaveformAugmenter(self.wave_sampler) if enabled else None
        if self.model_type == "whisper":
            self.feature_extractor = HipWhisperFeatureExtractor(feature_size or 80, device=device)
        else:
            self.feature_extractor = HipSpeech2TextFeatureExtractor(num_mel_bins=feature_size or 80, device=device)
        # src/data.py:251-258: Speech2Text truncation limits come from the model config
        self.max_source_positions = max_source_positions
        self.max_target_positions = max_target_positions
        self._ingest = None                  # same-rate copy / convert pass for pinned batches (zero-copy ingest)
        self._held: List[Any] = []           # (event, pinned batch tensor) pairs still being read by the GPU

    @property
    def device(self) -> torch.device:
        return _resolve_device(self._device_arg)

    # ---------------------------------------------------------------- labels (host, src/data.py:365-380,500)
    def _labels(self, texts: Sequence[str]) -> Optional[torch.Tensor]:
        if self.tokenizer is None:
            return None
        ids = []
        for text in texts:
            if self.model_type == "speech2text" and self.max_target_positions is not None:
                enc = self.tokenizer(text, return_tensors="pt", truncation=True, max_length=self.max_target_positions)
            else:
                enc = self.tokenizer(text, return_tensors="pt")
            ids.append(torch.as_tensor(enc.input_ids).reshape(-1))
        return pad_sequence(ids, batch_first=True, padding_value=-100)

    # ---------------------------------------------------------------- features
    def _spec_shape(self, n_samples: int):
        """(n_freq, n_time) of one clip's features, as the reference's per-clip extractor call returns them."""
        fe = self.feature_extractor
        if self.model_type == "whisper":
            return fe.feature_size, fe.nb_max_frames
        return fe.num_mel_bins, fe.num_frames(int(n_samples), int(self.max_source_positions or 0) or None)

    def features(self, pcm: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor,
                 host_lengths: Sequence[int], mask_table: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Packed 16 kHz PCM on the device -> {"input_features"[, "attention_mask"]} (device tensors).

        ``mask_table``: SpecAugment masks already drawn by the caller (``__call__`` draws them clip by clip,
        interleaved with the waveform draws, as the reference does); None = draw them here."""
        fe = self.feature_extractor
        augment = self.sampler is not None and self.split == "train"
        B = len(host_lengths)
        if self.model_type == "whisper":
            table = mask_table
            if table is None and augment:
                table = self.sampler.mask_table([(fe.feature_size, fe.nb_max_frames)] * B)
            feats = fe.extract(pcm, offsets, lengths, mask_table=table)               # (B, M, 3000)
            if self.transpose_features:
                feats = feats.transpose(1, 2)
            return {"input_features": feats}
        max_frames = int(self.max_source_positions or 0)
        frames = [fe.num_frames(int(n), max_frames or None) for n in host_lengths]
        t_max = max(frames) if frames else 0
        table = mask_table
        if table is None and augment:
            table = self.sampler.mask_table([(fe.num_mel_bins, t) for t in frames])
        feats, attn = fe.extract(pcm, offsets, lengths, t_max, max_frames, mask_table=table,
                                 attention_dtype=torch.int64)                         # (B, T_max, M)
        if not self.transpose_features:
            feats = feats.transpose(1, 2)             # the "(batch, freq, time)" form of src/data.py:477
        return {"input_features": feats, "attention_mask": attn}

    def __call__(self, batch: Optional[Dict[str, Any]]) -> Optional[Dict[str, Any]]:
        if batch is None:
            return None
        device = self.device
        audio = batch["audio"]
        B, n_max = int(audio.shape[0]), int(audio.shape[-1])
        if "audio_lengths" in batch:
            host_lengths = [int(v) for v in torch.as_tensor(batch["audio_lengths"]).tolist()]
        else:
            host_lengths = [n_max] * B                # reference batch without lengths: padding counts as audio
        lengths = torch.tensor(host_lengths, dtype=torch.int32).to(device, non_blocking=True)
        offsets = (torch.arange(B, dtype=torch.int64) * n_max).to(device, non_blocking=True)
        texts = list(batch.get("texts", batch.get("text", [""] * B)))
        if (audio.device.type == "cpu" and audio.is_pinned() and audio.is_contiguous()
                and audio.dtype in (torch.float32, torch.int16) and audio.numel() == B * n_max):
            # A pinned batch (DataLoader(pin_memory=True), src/config.py:77, train.py:1457-1464; float32 as the reference
            # collates it, or int16 decoded PCM): the same-rate copy kernel pulls the clips' VALID samples over PCIe
            # itself (sttfe_host_mapped_pointer) -- no H2D copy of the padded (B, n_max) tensor, whose padding is half
            # of a Common-Voice batch, and no staging buffer.  The batch tensor is kept alive until the kernel is done.
            if self._ingest is None:
                from .preprocess import GpuAudioPreprocessor
                self._ingest = GpuAudioPreprocessor(sample_rate=self.sample_rate, normalize=False, trim_silence=False,
                                                    min_duration=0.0, max_duration=1e30, device=device)
            p = self._ingest.run(audio.reshape(-1), offsets, lengths, host_lengths, 1, self.sample_rate,
                                 sync_lengths=False, device=device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            self._held = [(e, t) for e, t in self._held if not e.query()] + [(ev, audio)]
            pcm, offsets, lengths = p.pcm, p.offsets, p.lengths
        else:
            pcm = audio.reshape(B, n_max).to(device=device, dtype=torch.float32, non_blocking=True).contiguous()
            pcm = pcm.reshape(-1)
        table = None
        if self.wave_sampler is not None and self.split == "train":
            # One clip after the other, as CommonVoiceDataset.__getitem__ does (src/data.py:328-329, :360-361):
            # the waveform draws of clip b, then its SpecAugment draws, from the same two generators.
            from .preprocess import PackedAudio
            draws, rows = [], []
            for n in host_lengths:
                d = self.wave_sampler.draw_clip(n)
                draws.append(d)
                rows.append(self.sampler.draw_clip(*self._spec_shape(self.wave_sampler.out_length(n, d))))
            packed = self.wave_augmenter.apply(
                PackedAudio(pcm, offsets, lengths, list(host_lengths), self.sample_rate, [True] * B), draws)
            if not all(packed.keep):
                # torchaudio's Fade raises when the fade is longer than the clip, and the reference lets it propagate
                raise RuntimeError("fade longer than the clip (fade_duration vs min_duration)")
            pcm, offsets, lengths, host_lengths = packed.pcm, packed.offsets, packed.lengths, packed.host_lengths
            table_np = np.zeros((B, max(self.sampler.masks_per_clip, 1), 3), dtype=np.int32)
            for b, masks in enumerate(rows):
                if masks:
                    table_np[b, :len(masks)] = masks
            table = torch.from_numpy(table_np)
        result: Dict[str, Any] = self.features(pcm, offsets, lengths, host_lengths, mask_table=table)
        labels = self._labels(texts)
        if labels is not None:
            result["labels"] = labels.to(device, non_blocking=True)
        result["text"] = texts
        return result
