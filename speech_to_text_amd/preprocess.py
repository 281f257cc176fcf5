"""Batched waveform preprocessing on the GPU: mono mix, resample, peak normalise, silence trim, gate.

Device-side counterpart of ``AudioPreprocessor`` (reference ``src/data.py:42-116``; the same steps are
repeated in ``inference.py:100-132``).  The reference runs these per clip on the CPU inside DataLoader
workers; here a whole batch of decoded PCM goes through ``sttfe_resample_peak`` +
``sttfe_normalize_trim`` (``include/sttfe.h``) and stays in HBM, addressed by device
``offsets`` / ``lengths`` arrays that the feature kernels consume directly (trimming moves no samples).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Union

from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from .processors import _resolve_device

_ALIGN = 64          # samples; keeps every clip start 256-byte aligned


@dataclass
class PackedAudio:
    """A batch of mono clips in one device buffer."""
    pcm: torch.Tensor            # float32 (total,)
    offsets: torch.Tensor        # int64 (B,) device: clip b starts at pcm[offsets[b]]
    lengths: torch.Tensor        # int32 (B,) device
    host_lengths: List[int]      # the same lengths on the host
    sample_rate: int
    keep: List[bool]             # duration gate (src/data.py:95-98); False = the reference returns None
    scale: Optional[torch.Tensor] = None      # float32 (B,) device: pending peak normalisation (defer_normalize)

    @property
    def durations(self) -> List[float]:
        return [n / self.sample_rate for n in self.host_lengths]

    def clip(self, b: int) -> torch.Tensor:
        """Clip b as a (1, n) tensor view (the shape ``preprocess`` returns in the reference)."""
        o = int(self.offsets[b].item())
        return self.pcm[o:o + self.host_lengths[b]].unsqueeze(0)


class ResamplePlan:
    """Host + device copies of one (orig_sr, new_sr) polyphase kernel (torchaudio Resample defaults)."""

    def __init__(self, orig_sr: int, new_sr: int):
        L = _lib.lib()
        n = ctypes.c_size_t()
        _lib.check(L.sttfe_resample_plan_bytes(int(orig_sr), int(new_sr), ctypes.byref(n)), "sttfe_resample_plan_bytes")
        self.host = torch.empty(n.value, dtype=torch.uint8)
        _lib.check(L.sttfe_resample_plan_build(int(orig_sr), int(new_sr), _lib.ptr(self.host), n.value))
        vals = [ctypes.c_int() for _ in range(5)]
        _lib.check(L.sttfe_resample_plan_info(_lib.ptr(self.host), *[ctypes.byref(v) for v in vals]))
        self.orig, self.new, self.width, self.n_taps, self.taps_row = [v.value for v in vals]
        self._dev: Dict[int, torch.Tensor] = {}

    def on(self, device: torch.device) -> torch.Tensor:
        t = self._dev.get(device.index)
        if t is None:
            t = self.host.to(device)
            self._dev[device.index] = t
        return t

    def out_length(self, n_in: int) -> int:
        return (self.new * n_in + self.orig - 1) // self.orig

    @property
    def taps(self) -> np.ndarray:
        """(new, n_taps) float32 kernel, same layout as torchaudio's (new, 1, n_taps) squeezed."""
        off = 32 + 8 * self.new
        rows = self.host[off:].numpy().view(np.float32).reshape(self.new, self.taps_row)
        return rows[:, :self.n_taps].copy()


def _as_planar(c):
    """One clip -> ((channels, n) array or tensor, is_int16).  int16 stays int16 (decoded PCM), the rest is float32."""
    if torch.is_tensor(c):
        a = c.detach()
        a = a.reshape(1, -1) if a.dim() == 1 else a.reshape(a.shape[0], -1)
        return (a, True) if a.dtype == torch.int16 else (a.to(torch.float32), False)
    a = np.asarray(c)
    s16 = a.dtype == np.int16
    if not s16:
        a = a.astype(np.float32, copy=False)
    return (a.reshape(1, -1) if a.ndim == 1 else a.reshape(a.shape[0], -1)), s16


def _layout(arrs):
    lens, chans = [], None
    for a in arrs:
        if chans is None:
            chans = int(a.shape[0])
        elif chans != int(a.shape[0]):
            raise ValueError("pack_planar: all clips of one call must have the same channel count")
        lens.append(int(a.shape[1]))
    starts, total = [], 0
    for n in lens:
        starts.append(total)
        total += (chans * n + _ALIGN - 1) // _ALIGN * _ALIGN
    return lens, chans or 1, starts, total


def _fill_stage(stage: torch.Tensor, arrs, idx, starts, lens, chans):
    sn = stage.numpy()
    for i in idx:
        a = arrs[i].numpy() if torch.is_tensor(arrs[i]) else arrs[i]
        sn[starts[i]:starts[i] + chans * lens[i]] = a.reshape(-1)


def pack_planar(clips: Sequence, device: torch.device):
    """Clips of identical channel count -> (pcm, offsets int64, lengths int32, host lengths, channels).

    Clip b is stored as ``channels`` planar rows of ``lengths[b]`` samples (what ``sttfe_resample_peak``
    expects).  Host arrays go through one pinned staging buffer.  When every clip is int16 (decoded PCM) the
    buffer stays int16 -- half the bytes over PCIe -- and ``sttfe_resample_peak_s16`` converts on the fly;
    mixed input is converted to float32 on the host (x / 32768).
    """
    pairs = [_as_planar(c) for c in clips]
    s16 = bool(pairs) and all(f for _, f in pairs)
    arrs = []
    for a, f in pairs:
        if f and not s16:
            a = (a.to(torch.float32) if torch.is_tensor(a) else a.astype(np.float32)) / 32768.0
        arrs.append(a)
    dtype = torch.int16 if s16 else torch.float32
    lens, chans, starts, total = _layout(arrs)
    pcm = torch.empty(total + _ALIGN, dtype=dtype, device=device)
    host_idx = [i for i, a in enumerate(arrs) if not torch.is_tensor(a) or not a.is_cuda]
    if host_idx:
        stage = torch.empty(total + _ALIGN, dtype=dtype, pin_memory=torch.cuda.is_available())
        _fill_stage(stage, arrs, host_idx, starts, lens, chans)
        if len(host_idx) == len(arrs):
            pcm.copy_(stage, non_blocking=True)
        else:
            for i in host_idx:
                s, e = starts[i], starts[i] + chans * lens[i]
                pcm[s:e].copy_(stage[s:e], non_blocking=True)
    for i, a in enumerate(arrs):
        if torch.is_tensor(a) and a.is_cuda:
            pcm[starts[i]:starts[i] + chans * lens[i]].copy_(a.reshape(-1))
    offsets = torch.tensor(starts, dtype=torch.int64).to(device, non_blocking=True)
    lengths = torch.tensor(lens, dtype=torch.int32).to(device, non_blocking=True)
    return pcm, offsets, lengths, lens, chans


class PinnedIngestRing:
    """Decoded-PCM ingest: a ring of pinned staging buffers and a copy stream.

    DataLoader workers only decode (``torchaudio.load``, reference ``src/data.py:313``; ``pin_memory`` /
    ``num_workers`` at ``train.py:1457-1464``); the main process calls ``submit(clips)`` as batches arrive --
    one host memcpy into the next pinned slot and one asynchronous H2D copy on the ring's own stream -- and
    ``take(ticket)`` when the GPU front-end is ready for the batch: the compute stream then waits on the copy's
    event, not the host.  With ``slots`` >= 2 the copy of batch k+1 runs under the kernels of batch k.
    int16 clips stay int16 on the wire (see ``pack_planar``).
    """

    def __init__(self, device: Union[str, torch.device], slots: int = 3, fill_threads: int = 4):
        self.device = _resolve_device(device)
        self.stream = torch.cuda.Stream(self.device)
        # the memcpy into the pinned slot releases the GIL: a few threads fill contiguous groups of clips, and each
        # group's H2D copy is queued as soon as it is filled (the single-threaded fill was the bound for int16 PCM)
        self._pool = ThreadPoolExecutor(max_workers=int(fill_threads)) if int(fill_threads) > 1 else None
        self._groups = max(2 * int(fill_threads), 1)
        self._slots: List[Optional[torch.Tensor]] = [None] * max(int(slots), 1)
        self._events: List[Optional[torch.cuda.Event]] = [None] * len(self._slots)
        self._next = 0

    def submit(self, clips: Sequence):
        """Host clips (numpy / CPU tensors, all int16 or all float) -> ticket for ``take``."""
        pairs = [_as_planar(c) for c in clips]
        s16 = bool(pairs) and all(f for _, f in pairs)
        arrs = [a if (s16 or not f) else (a.to(torch.float32) if torch.is_tensor(a) else a.astype(np.float32)) / 32768.0
                for a, f in pairs]
        dtype = torch.int16 if s16 else torch.float32
        lens, chans, starts, total = _layout(arrs)
        k = self._next
        self._next = (k + 1) % len(self._slots)
        if self._events[k] is not None:
            self._events[k].synchronize()          # the slot's previous copy has left the pinned buffer
        slot = self._slots[k]
        if slot is None or slot.dtype != dtype or slot.numel() < total + _ALIGN:
            slot = torch.empty(max(total + _ALIGN, 1), dtype=dtype, pin_memory=True)
            self._slots[k] = slot
        with torch.cuda.stream(self.stream):
            pcm = torch.empty(total + _ALIGN, dtype=dtype, device=self.device)
            if self._pool is None or len(arrs) < 2:
                _fill_stage(slot, arrs, range(len(arrs)), starts, lens, chans)
                pcm.copy_(slot[:total + _ALIGN], non_blocking=True)
            else:
                # contiguous groups of about equal size; copies are queued in order as the fills complete
                bounds, target, acc = [0], total / self._groups, 0
                for i in range(len(arrs)):
                    acc += chans * lens[i]
                    if acc >= target * len(bounds) and i + 1 < len(arrs):
                        bounds.append(i + 1)
                bounds.append(len(arrs))
                futs = [self._pool.submit(_fill_stage, slot, arrs, range(bounds[g], bounds[g + 1]), starts, lens, chans)
                        for g in range(len(bounds) - 1)]
                for g, f in enumerate(futs):
                    f.result()
                    lo = starts[bounds[g]]
                    hi = starts[bounds[g + 1]] if bounds[g + 1] < len(arrs) else total + _ALIGN
                    pcm[lo:hi].copy_(slot[lo:hi], non_blocking=True)
            offsets = torch.tensor(starts, dtype=torch.int64).to(self.device, non_blocking=True)
            lengths = torch.tensor(lens, dtype=torch.int32).to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._events[k] = ev
        return (pcm, offsets, lengths, lens, chans, ev)

    def take(self, ticket):
        """-> (pcm, offsets, lengths, host_lengths, channels), ready for ``GpuAudioPreprocessor.run`` on the
        current stream."""
        pcm, offsets, lengths, lens, chans, ev = ticket
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in (pcm, offsets, lengths):
            t.record_stream(cur)                   # allocated on the copy stream, consumed on the compute stream
        return pcm, offsets, lengths, lens, chans


class GpuAudioPreprocessor:
    """``AudioPreprocessor`` (src/data.py:42-116) for whole batches, on the GPU.

    ``config`` is the reference's project config (``config.data.sample_rate`` ...), a ``DataConfig``,
    or None with keyword overrides.
    """

    def __init__(self, config: Any = None, *, device: Optional[Union[str, torch.device]] = None, **overrides):
        data = getattr(config, "data", config)
        get = lambda k, d: overrides.get(k, getattr(data, k, d) if data is not None else d)
        self.sample_rate = int(get("sample_rate", 16000))
        self.min_duration = float(get("min_duration", 0.5))
        self.max_duration = float(get("max_duration", 30.0))
        self.normalize = bool(get("normalize", True))
        self.trim_silence = bool(get("trim_silence", True))
        self.silence_threshold = float(get("silence_threshold", 0.01))
        self._device_arg = device
        self._plans: Dict[int, ResamplePlan] = {}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_plans"] = {}
        return d

    @property
    def device(self) -> torch.device:
        return _resolve_device(self._device_arg)

    def plan(self, original_sample_rate: int) -> Optional[ResamplePlan]:
        if int(original_sample_rate) == self.sample_rate:
            return None
        p = self._plans.get(int(original_sample_rate))
        if p is None:
            p = ResamplePlan(int(original_sample_rate), self.sample_rate)
            self._plans[int(original_sample_rate)] = p
        return p

    def run(self, pcm_in: torch.Tensor, in_offsets: torch.Tensor, in_lengths: torch.Tensor,
            host_lengths: Sequence[int], channels: int, original_sample_rate: int,
            sync_lengths: Optional[bool] = None, defer_normalize: bool = False, copy: bool = False,
            device: Optional[torch.device] = None) -> PackedAudio:
        """Device-resident entry: packed (planar) PCM at ``original_sample_rate`` -> PackedAudio at 16 kHz.

        ``defer_normalize`` (with ``normalize``): the samples are left unnormalised and ``PackedAudio.scale`` carries
        1 / (max|x| + 1e-8) per clip (1 for silent clips, as src/data.py:79 skips them) for
        ``HipWhisperFeatureExtractor.extract(..., clip_scale=...)``, which folds it into the feature kernel -- one
        pass over the waveform less.  Silence trimming then compares raw samples with ``peak * threshold`` instead of
        normalised ones with ``(peak / d) * threshold``: the bounds may differ by a sample where a sample sits exactly
        at the threshold.

        **In place:** mono float32 input at the target rate (no resample, no channel mix, no int16 conversion) is not
        copied -- ``PackedAudio.pcm`` IS ``pcm_in`` and peak normalisation overwrites the caller's buffer.  A caller
        that keeps decoded PCM resident across epochs passes ``pcm_in.clone()`` (or ``defer_normalize=True``, which
        leaves the samples alone).

        ``sync_lengths`` (default: only when trimming) copies the trimmed lengths back to the host,
        which is what the duration gate and Speech2Text's dynamic padding need.

        **Zero-copy ingest:** ``pcm_in`` may be a PINNED host tensor (``device=`` then names the GPU; offsets / lengths
        are device tensors as always): the kernel reads the clips' valid samples over PCIe itself
        (``sttfe_host_mapped_pointer``), so the padding of a collated ``(B, n_max)`` batch never crosses the bus and no
        staging copy is made.  ``copy=True`` forces the packed copy for device input too (never aliases ``pcm_in``).
        """
        host_in = pcm_in.device.type == "cpu"
        if host_in:
            if device is None or not pcm_in.is_pinned():
                raise ValueError("host input must be pinned memory and needs device=")
            device = torch.device(device)
        else:
            device = pcm_in.device
        L = _lib.lib()
        B = len(host_lengths)
        plan = self.plan(original_sample_rate)
        out_lens = [plan.out_length(n) if plan else int(n) for n in host_lengths]
        starts, total = [], 0
        for n in out_lens:
            starts.append(total)
            total += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        s16 = pcm_in.dtype == torch.int16
        if not s16 and pcm_in.dtype != torch.float32:
            raise TypeError("packed PCM must be float32 or int16")
        alias = plan is None and channels == 1 and not s16 and not copy and not host_in
        in_ptr = _lib.ptr(pcm_in)
        if host_in:
            mapped = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(L.sttfe_host_mapped_pointer(in_ptr, ctypes.byref(mapped)), "sttfe_host_mapped_pointer")
            in_ptr = mapped
        if alias:
            pcm_out, out_offsets = pcm_in, in_offsets.clone()
        else:
            pcm_out = torch.empty(total + _ALIGN, dtype=torch.float32, device=device)
            out_offsets = torch.tensor(starts, dtype=torch.int64).to(device, non_blocking=True)
        out_lengths = torch.empty(B, dtype=torch.int32, device=device)
        ws_bytes = ctypes.c_size_t()
        _lib.check(L.sttfe_preprocess_workspace_bytes(B, ctypes.byref(ws_bytes)))
        ws = torch.empty(max(ws_bytes.value, 1), dtype=torch.uint8, device=device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        max_in = max(host_lengths) if B else 0
        with torch.cuda.device(device):
            entry = L.sttfe_resample_peak_s16 if s16 else L.sttfe_resample_peak
            _lib.check(entry(in_ptr, _lib.ptr(in_offsets), _lib.ptr(in_lengths), B,
                             int(channels), int(max_in),
                             _lib.ptr(plan.on(device)) if plan else None,
                             _lib.ptr(plan.host) if plan else None,
                             _lib.ptr(pcm_out), _lib.ptr(out_offsets), _lib.ptr(out_lengths),
                             _lib.ptr(ws), ws.numel(), stream), "sttfe_resample_peak")
            _lib.check(L.sttfe_normalize_trim(_lib.ptr(pcm_out), _lib.ptr(out_offsets), _lib.ptr(out_lengths), B,
                                              int(max(out_lens) if B else 0), int(self.normalize and not defer_normalize),
                                              int(self.trim_silence), ctypes.c_float(self.silence_threshold),
                                              _lib.ptr(ws), ws.numel(), stream), "sttfe_normalize_trim")
        self.last_peaks = ws[:4 * B].view(torch.float32)      # max|x| per clip before normalisation (debug / tests)
        if sync_lengths is None:
            sync_lengths = self.trim_silence
        final_lens = [int(v) for v in out_lengths.cpu().tolist()] if sync_lengths else out_lens
        # src/data.py:75,91,95: duration = n / sample_rate after resampling (after trimming when it trimmed)
        keep = [self.min_duration <= n / self.sample_rate <= self.max_duration for n in final_lens]
        scale = None
        if defer_normalize and self.normalize:
            peaks = self.last_peaks
            scale = torch.where(peaks > 0, 1.0 / (peaks + 1e-8), torch.ones_like(peaks))
        return PackedAudio(pcm_out, out_offsets, out_lengths, final_lens, self.sample_rate, keep, scale)

    def preprocess_batch(self, clips: Sequence, original_sample_rate: int) -> PackedAudio:
        """List of (C, n) / (n,) arrays or tensors, all at ``original_sample_rate`` -> PackedAudio."""
        device = self.device
        pcm, offsets, lengths, lens, chans = pack_planar(list(clips), device)
        return self.run(pcm, offsets, lengths, lens, chans, original_sample_rate)

    def preprocess(self, audio, original_sample_rate: int) -> Optional[Dict[str, Any]]:
        """Single-clip form with the reference's signature and return value (src/data.py:54-108):
        ``{"audio": (1, n) tensor (on the GPU), "sample_rate", "duration"}`` or None when gated out."""
        try:
            packed = self.preprocess_batch([audio], original_sample_rate)
        except _lib.SttfeError:
            raise
        except Exception:
            return None
        if not packed.keep[0]:
            return None
        return {"audio": packed.clip(0), "sample_rate": self.sample_rate, "duration": packed.durations[0]}

    def __repr__(self) -> str:
        return (f"GpuAudioPreprocessor(sample_rate={self.sample_rate}, normalize={self.normalize}, "
                f"trim_silence={self.trim_silence})")
