"""SpecAugment for the GPU collate mode: the host draws the mask bounds, the kernels apply them.

Counterpart of ``AudioAugmentationCPU.apply_spectrogram_augmentations`` (reference
``src/data.py:185-194``) with ``torchaudio.transforms.TimeMasking(time_mask_param=...)`` /
``FrequencyMasking(freq_mask_param=...)`` (``src/data.py:129-130``).  A mask is four integers, so the
random draws stay on the host, in the reference's order and from the same generators (Python
``random`` for the gate, the global torch CPU generator for the bounds): seeding both reproduces the
reference's masks bit for bit.  The (B, n_masks, 3) int32 table {axis, start, end} is handed to
``sttfe_whisper_logmel`` / ``sttfe_kaldi_fbank_cmvn``, which write 0.0 over the masked spans in the
same pass that normalises the features.

Draw order per clip (torchaudio 2.5.1 ``functional.mask_along_axis`` with p = 1.0, mask_value = 0.0)::

    random.random() < spec_augment_probability              # gate, consumed even when it fails
    for each time mask, then each frequency mask:
        value     = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (axis_len - value)
        start, end = floor(min_value), floor(min_value) + floor(value)
"""
from __future__ import annotations

import ctypes
import logging
import random
import struct
from dataclasses import dataclass
from math import gcd
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

logger = logging.getLogger(__name__)

AXIS_FREQ = 0        # STTFE_AXIS_FREQ
AXIS_TIME = 1        # STTFE_AXIS_TIME


class SpecAugmentSampler:
    """Draws SpecAugment masks in the reference's RNG order.  ``config`` = AugmentationConfig-like."""

    def __init__(self, config: Any = None, **overrides):
        get = lambda k, d: overrides.get(k, getattr(config, k, d) if config is not None else d)
        self.spec_augment = bool(get("spec_augment", True))
        self.probability = float(get("spec_augment_probability", 0.4))
        self.time_mask_max_size = int(get("time_mask_max_size", 20))
        self.freq_mask_max_size = int(get("freq_mask_max_size", 30))
        self.num_time_masks = int(get("num_time_masks", 2))
        self.num_freq_masks = int(get("num_freq_masks", 2))

    @property
    def masks_per_clip(self) -> int:
        return self.num_time_masks + self.num_freq_masks

    @staticmethod
    def _draw(axis_len: int, mask_param: int) -> Tuple[int, int]:
        if mask_param < 1:                       # mask_along_axis returns the input untouched, no draws
            return 0, 0
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (axis_len - value)
        start = int(min_value.long().item())
        return start, start + int(value.long().item())

    def draw_clip(self, n_freq: int, n_time: int) -> List[Tuple[int, int, int]]:
        """Masks of one (n_freq, n_time) clip: [] when the gate fails, else time masks then freq masks.

        All uniform draws of the clip come from one ``torch.rand(2 k)`` call: the CPU generator hands out the same
        numbers, in the same order and leaving the same state, as 2 k successive ``torch.rand(1)`` calls
        (tests/test_specaugment_cpu.py pins that), and the float32 products are the same element-wise."""
        if not (self.spec_augment and random.random() < self.probability):
            return []
        plan = [(AXIS_TIME, n_time, self.time_mask_max_size)] * self.num_time_masks \
            + [(AXIS_FREQ, n_freq, self.freq_mask_max_size)] * self.num_freq_masks
        live = [(a, n, p) for a, n, p in plan if p >= 1]      # mask_param < 1: input untouched, no draws
        masks: List[Tuple[int, int, int]] = []
        if live:
            u = torch.rand(2 * len(live))
            value = u[0::2] * torch.tensor([float(p) for _, _, p in live])
            min_value = u[1::2] * (torch.tensor([float(n) for _, n, _ in live]) - value)
            starts, spans = min_value.long().tolist(), value.long().tolist()
        it = iter(range(len(live)))
        for a, n, p in plan:
            if p < 1:
                masks.append((a, 0, 0))
            else:
                i = next(it)
                masks.append((a, starts[i], starts[i] + spans[i]))
        return masks

    def mask_table(self, shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
        """One draw per clip, in batch order -> (B, masks_per_clip, 3) int32 host tensor.

        Clips whose gate failed get no-op rows (start == end == 0).  The whole batch is drawn at once: the gates are
        the only users of Python's generator in this loop, and the clips that pass take their uniform numbers from ONE
        ``torch.rand`` call -- the CPU generator hands them out sequentially, so clip after clip gets the numbers (and
        the generator the state) that per-clip ``draw_clip`` calls would give; the float32 products are element-wise
        the same (tests/test_specaugment_cpu.py holds this method to ``draw_clip``).  256 clips: 2 ms -> 0.1 ms.
        """
        B, M = len(shapes), max(self.masks_per_clip, 1)
        tbl = torch.zeros((B, M, 3), dtype=torch.int32)
        if B == 0:
            return tbl
        plan = [(AXIS_TIME, self.time_mask_max_size)] * self.num_time_masks \
            + [(AXIS_FREQ, self.freq_mask_max_size)] * self.num_freq_masks
        passed = [b for b in range(B) if self.spec_augment and random.random() < self.probability]
        if not passed or not plan:
            return tbl
        axes = torch.tensor([a for a, _ in plan], dtype=torch.int32)
        tbl[passed, :len(plan), 0] = axes                                     # (param < 1: the row stays (axis, 0, 0))
        live = [i for i, (_, p) in enumerate(plan) if p >= 1]                 # mask_param < 1: input untouched, no draws
        if not live:
            return tbl
        shp = torch.tensor([[int(shapes[b][0]), int(shapes[b][1])] for b in passed], dtype=torch.float32)   # (n_freq, n_time)
        axis_len = torch.stack([shp[:, 1] if plan[i][0] == AXIS_TIME else shp[:, 0] for i in live], dim=1)  # (passed, live)
        params = torch.tensor([float(plan[i][1]) for i in live])
        u = torch.rand(2 * len(live) * len(passed)).view(len(passed), len(live), 2)      # per clip: value, min_value, value, ...
        value = u[:, :, 0] * params
        min_value = u[:, :, 1] * (axis_len - value)
        starts, spans = min_value.long(), value.long()
        rows = torch.tensor(passed)
        cols = torch.tensor(live)
        tbl[rows[:, None], cols[None, :], 1] = starts.to(torch.int32)
        tbl[rows[:, None], cols[None, :], 2] = (starts + spans).to(torch.int32)
        return tbl


# ------------------------------------------------------------------ waveform-domain augmentations
AUG_NOISE, AUG_GAIN, AUG_FADE = 1, 2, 4          # STTFE_AUG_*
PITCH_NONE = -2 ** 31                            # STTFE_PITCH_NONE
PITCH_MODES = {"as_constructed": 0, "functional": 1, "off": None}      # STTFE_PITCH_AS_CONSTRUCTED / _FUNCTIONAL
PITCH_MIN_SAMPLES = 257                          # torch.stft's reflect padding needs n > n_fft / 2


@dataclass
class WaveformDraw:
    """The random draws of one clip (reference ``src/data.py:155-181``); None = the gate did not pass."""
    noise_factor: Optional[float] = None
    noise: Optional[torch.Tensor] = None         # (n,) host standard-normal samples when noise_source == "host"
    speed_factor: Optional[float] = None
    pitch_steps: Optional[int] = None            # semitones, random.randint(*pitch_shift_range)
    gain: Optional[float] = None
    fade_samples: Optional[int] = None


class WaveformAugmentSampler:
    """Draws the time-domain augmentation parameters in the reference's RNG order.

    Counterpart of ``AudioAugmentationCPU.apply_time_domain_augmentations`` (``src/data.py:152-183``):
    per clip, in order, ``random.random()`` gates for noise / speed / pitch / volume / fade, each followed
    (when it passes) by its value draw: ``torch.randn_like(audio)``, ``torch.randint(len(factors), ())``,
    ``random.randint(*pitch_shift_range)``, ``random.uniform(*volume_range_db)``, none for fade.

    ``noise_source="host"`` draws the noise itself from the torch CPU generator, exactly like the reference
    (bit parity; n floats per noisy clip cross PCIe); ``"device"`` leaves it to the GPU generator (same
    distribution, different stream) -- the default, since the noise is 4 bytes per sample.
    ``pitch_mode``: ``"as_constructed"`` (default) applies what the reference's ``PitchShift(sample_rate, n_steps=0)``
    object computes once its ``n_steps`` attribute is overwritten -- the phase-vocoder stretch without the
    resample, whose ratio the transform fixed at construction -- ``"functional"`` applies
    ``torchaudio.functional.pitch_shift`` with the drawn steps, ``"off"`` consumes the draws and leaves the clip
    alone (``sttfe_pitch_shift``, include/sttfe.h).
    """

    def __init__(self, config: Any = None, sample_rate: int = 16000, noise_source: str = "device",
                 pitch_mode: str = "as_constructed", **overrides):
        get = lambda k, d: overrides.get(k, getattr(config, k, d) if config is not None else d)
        if noise_source not in ("host", "device"):
            raise ValueError("noise_source must be 'host' or 'device'")
        if pitch_mode not in PITCH_MODES:
            raise ValueError("pitch_mode must be one of %s" % (sorted(PITCH_MODES),))
        self.noise_source = noise_source
        self.pitch_mode = pitch_mode
        self.sample_rate = int(sample_rate)
        self.add_noise = bool(get("add_noise", True))
        self.noise_probability = float(get("noise_probability", 0.3))
        self.noise_factor = float(get("noise_factor", 0.02))
        self.speed_perturbation = bool(get("speed_perturbation", True))
        self.speed_probability = float(get("speed_probability", 0.3))
        lo, hi = get("speed_factor_range", (0.9, 1.1))
        self.speed_factors = np.linspace(lo, hi, int(get("speed_num_steps", 3))).tolist()   # src/data.py:145-149
        # (any speed factor is accepted, as by torchaudio: reduced ratios beyond 1024:1024 -- int(factor * sample_rate)
        # sharing no large divisor with sample_rate -- run through the table-free resampler, see csrc/tables.cpp.
        # Constructing the sampler touches neither the library nor the GPU: it is host configuration + RNG state.)
        self.pitch_shift = bool(get("pitch_shift", True))
        self.pitch_probability = float(get("pitch_probability", 0.3))
        self.pitch_shift_range = tuple(get("pitch_shift_range", (-2, 2)))
        self.volume_perturbation = bool(get("volume_perturbation", True))
        self.volume_probability = float(get("volume_probability", 0.3))
        self.volume_range_db = tuple(get("volume_range_db", (-10, 10)))
        self.fade_inout = bool(get("fade_inout", True))
        self.fade_probability = float(get("fade_probability", 0.2))
        self.fade_duration = float(get("fade_duration", 0.1))

    def draw_clip(self, n_samples: int) -> WaveformDraw:
        d = WaveformDraw()
        if self.add_noise and random.random() < self.noise_probability:
            d.noise_factor = self.noise_factor
            if self.noise_source == "host":
                d.noise = torch.randn(1, int(n_samples)).reshape(-1)       # == randn_like of the (1, n) clip
        if self.speed_perturbation and random.random() < self.speed_probability:
            d.speed_factor = float(self.speed_factors[int(torch.randint(len(self.speed_factors), ()))])
        if self.pitch_shift and random.random() < self.pitch_probability:
            d.pitch_steps = random.randint(*self.pitch_shift_range)
        if self.volume_perturbation and random.random() < self.volume_probability:
            d.gain = 10 ** (random.uniform(*self.volume_range_db) / 20.0)
        if self.fade_inout and random.random() < self.fade_probability:
            d.fade_samples = int(self.fade_duration * self.sample_rate)
        return d

    def speed_rates(self, factor: float) -> Tuple[int, int]:
        """(source, target) of torchaudio.functional.speed's resample, reduced by their gcd."""
        source, target = int(factor * self.sample_rate), self.sample_rate
        g = gcd(source, target)
        return source // g, target // g

    def out_length(self, n: int, d: WaveformDraw) -> int:
        """Clip length after the draw's speed perturbation (ceil(target * n / source))."""
        if d.speed_factor is None or d.speed_factor == 1.0:
            return int(n)
        src, dst = self.speed_rates(d.speed_factor)
        return int(n) if src == dst else (dst * int(n) + src - 1) // src


def _fbits(v: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(v)))[0]


class GpuWaveformAugmenter:
    """Applies a batch of ``WaveformDraw`` to packed PCM on the GPU (``sttfe_waveform_augment`` + resampler).

    Order as in the reference: noise -> speed -> pitch -> volume -> fade.
    """

    def __init__(self, sampler: WaveformAugmentSampler):
        self.sampler = sampler
        self._plans: Dict[Tuple[int, int], Any] = {}

    def _plan(self, src: int, dst: int):
        from .preprocess import ResamplePlan
        p = self._plans.get((src, dst))
        if p is None:
            p = ResamplePlan(src, dst)
            self._plans[(src, dst)] = p
        return p

    def pitch_shift(self, packed, steps: Sequence[Optional[int]], mode: Optional[str] = None):
        """In place: clip b of ``packed`` shifted by ``steps[b]`` semitones (None = untouched) with
        ``sttfe_pitch_shift``.  Raises RuntimeError for an active clip of fewer than 257 samples (torch.stft
        raises there in the reference)."""
        from . import _lib
        mode_id = PITCH_MODES[mode or self.sampler.pitch_mode]
        B = len(steps)
        if mode_id is None or B == 0 or all(s is None for s in steps):
            return packed
        L = _lib.lib()
        lens = np.asarray(packed.host_lengths, dtype=np.int32)
        if len(lens) != B:
            raise ValueError("one pitch step entry per clip is required")
        n_steps = np.asarray([PITCH_NONE if s is None else int(s) for s in steps], dtype=np.int64).astype(np.int32)
        short = [b for b in range(B) if steps[b] is not None and lens[b] < PITCH_MIN_SAMPLES]
        if short:
            raise RuntimeError("pitch shift needs clips of at least %d samples (clips %s)" % (PITCH_MIN_SAMPLES, short))
        layout = np.zeros((B, 8), dtype=np.int64)
        ws_bytes = ctypes.c_size_t()
        _lib.check(L.sttfe_pitch_shift_plan(lens.ctypes.data_as(ctypes.c_void_p), n_steps.ctypes.data_as(ctypes.c_void_p),
                                            B, int(packed.sample_rate), mode_id,
                                            layout.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ws_bytes)),
                   "sttfe_pitch_shift_plan")
        device = packed.pcm.device
        ws = torch.empty(max(ws_bytes.value, 16), dtype=torch.uint8, device=device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        layout_dev = torch.from_numpy(layout).to(device, non_blocking=True)
        with torch.cuda.device(device):
            _lib.check(L.sttfe_pitch_shift(_lib.ptr(packed.pcm), _lib.ptr(packed.offsets), _lib.ptr(packed.lengths), B,
                                           layout.ctypes.data_as(ctypes.c_void_p), _lib.ptr(layout_dev), mode_id,
                                           _lib.ptr(ws), ws.numel(), stream), "sttfe_pitch_shift")
        return packed

    def apply(self, packed, draws: Sequence[WaveformDraw]):
        """``packed``: preprocess.PackedAudio (mono, device).  Returns a PackedAudio (new buffers when any clip's
        speed changed; otherwise the same buffers, modified in place).  Clips whose fade is longer than the clip
        are dropped (``keep`` False): torchaudio's Fade raises there and the reference skips the sample."""
        from . import _lib
        from .preprocess import _ALIGN, PackedAudio
        L = _lib.lib()
        device = packed.pcm.device
        B = len(draws)
        if B == 0:
            return packed
        lens = list(packed.host_lengths)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        params_np = np.zeros((B, 4), dtype=np.int32)
        keep = list(packed.keep)
        noise_dev = noise_off = None
        any_noise = any(d.noise_factor is not None for d in draws)
        if any_noise:
            starts, total = [], 0
            for b, d in enumerate(draws):
                starts.append(total)
                if d.noise_factor is not None:
                    total += (lens[b] + _ALIGN - 1) // _ALIGN * _ALIGN
            if self.sampler.noise_source == "host":
                stage = torch.zeros(total + _ALIGN, dtype=torch.float32, pin_memory=torch.cuda.is_available())
                for b, d in enumerate(draws):
                    if d.noise_factor is not None:
                        stage[starts[b]:starts[b] + lens[b]] = d.noise[:lens[b]]
                noise_dev = stage.to(device, non_blocking=True)
            else:
                noise_dev = torch.randn(total + _ALIGN, dtype=torch.float32, device=device)
            noise_off = torch.tensor(starts, dtype=torch.int64).to(device, non_blocking=True)
        for b, d in enumerate(draws):
            flags = 0
            if d.noise_factor is not None:
                flags |= AUG_NOISE
                params_np[b, 1] = _fbits(d.noise_factor)
            if d.gain is not None:
                flags |= AUG_GAIN
                params_np[b, 2] = _fbits(d.gain)
            if d.fade_samples is not None:
                flags |= AUG_FADE
                params_np[b, 3] = int(d.fade_samples)
            params_np[b, 0] = flags
        params_dev = torch.from_numpy(params_np).to(device, non_blocking=True)
        speed_groups: Dict[Tuple[int, int], List[int]] = {}
        for b, d in enumerate(draws):
            if d.speed_factor is not None and d.speed_factor != 1.0:
                src, dst = self.sampler.speed_rates(d.speed_factor)
                if src != dst:
                    speed_groups.setdefault((src, dst), []).append(b)

        def run(p, stages):
            with torch.cuda.device(device):
                _lib.check(L.sttfe_waveform_augment(_lib.ptr(p.pcm), _lib.ptr(p.offsets), _lib.ptr(p.lengths), B,
                                                    int(max(p.host_lengths) if B else 0), _lib.ptr(params_dev),
                                                    _lib.ptr(noise_dev) if noise_dev is not None else None,
                                                    _lib.ptr(noise_off) if noise_off is not None else None,
                                                    int(stages), stream), "sttfe_waveform_augment")

        pitch_steps = [d.pitch_steps for d in draws]
        any_pitch = self.sampler.pitch_mode != "off" and any(s is not None for s in pitch_steps)
        if not speed_groups and not any_pitch:
            run(packed, AUG_NOISE | AUG_GAIN | AUG_FADE)
            out = packed
        elif not speed_groups:
            if any_noise:
                run(packed, AUG_NOISE)
            out = self.pitch_shift(packed, pitch_steps)
            run(out, AUG_GAIN | AUG_FADE)
        else:
            if any_noise:
                run(packed, AUG_NOISE)
            # speed: every group through the resampler into a fresh packing; untouched clips are copied
            new_lens = [self.sampler.out_length(lens[b], draws[b]) for b in range(B)]
            starts, total = [], 0
            for n in new_lens:
                starts.append(total)
                total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
            pcm_out = torch.empty(total + _ALIGN, dtype=torch.float32, device=device)
            out_offsets = torch.tensor(starts, dtype=torch.int64).to(device, non_blocking=True)
            out_lengths = torch.empty(B, dtype=torch.int32, device=device)
            ws_bytes = ctypes.c_size_t()
            groups = dict(speed_groups)
            rest = [b for b in range(B) if not any(b in g for g in speed_groups.values())]
            if rest:
                groups[(1, 1)] = rest
            for (src, dst), idx in groups.items():
                sel = torch.tensor(idx, dtype=torch.int64).to(device, non_blocking=True)
                in_off, in_len = packed.offsets[sel].contiguous(), packed.lengths[sel].contiguous()
                o_off, o_len = out_offsets[sel].contiguous(), torch.empty(len(idx), dtype=torch.int32, device=device)
                _lib.check(L.sttfe_preprocess_workspace_bytes(len(idx), ctypes.byref(ws_bytes)))
                ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=device)
                plan = self._plan(src, dst) if (src, dst) != (1, 1) else None
                with torch.cuda.device(device):
                    _lib.check(L.sttfe_resample_peak(_lib.ptr(packed.pcm), _lib.ptr(in_off), _lib.ptr(in_len), len(idx),
                                                     1, int(max(lens[b] for b in idx)),
                                                     _lib.ptr(plan.on(device)) if plan else None,
                                                     _lib.ptr(plan.host) if plan else None,
                                                     _lib.ptr(pcm_out), _lib.ptr(o_off), _lib.ptr(o_len),
                                                     _lib.ptr(ws), ws.numel(), stream), "sttfe_resample_peak")
                out_lengths[sel] = o_len
            out = PackedAudio(pcm_out, out_offsets, out_lengths, new_lens, packed.sample_rate, keep)
            if any_pitch:
                self.pitch_shift(out, pitch_steps)
            run(out, AUG_GAIN | AUG_FADE)
        for b, d in enumerate(draws):
            if d.fade_samples is not None and d.fade_samples > out.host_lengths[b]:
                out.keep[b] = False
        return out
