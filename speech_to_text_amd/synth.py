"""Seeded synthetic PCM for tests and benchmarks.

A clip is the sum of three linear chirps ``a*sin(2*pi*(f0 + k*t)*t + phi)`` with
a ~ U[0.05, 0.4], f0 ~ U[80, 3000] Hz, k ~ U[-200, 200] Hz/s, plus white noise with
sigma ~ U[0.002, 0.05], clipped to [-1, 1], float32.  One clip in ten gets a 0.2-1 s block
of exact zeros (the case that exercises the log floor / max-8 clamp).
"""
from __future__ import annotations

import numpy as np


def synth_clip(rng: np.random.Generator, n_samples: int, sample_rate: int = 16000,
               zero_block: bool | None = None) -> np.ndarray:
    t = np.arange(n_samples, dtype=np.float64) / sample_rate
    x = np.zeros(n_samples, dtype=np.float64)
    for _ in range(3):
        a = rng.uniform(0.05, 0.4)
        f0 = rng.uniform(80.0, 3000.0)
        k = rng.uniform(-200.0, 200.0)
        phi = rng.uniform(0.0, 2.0 * np.pi)
        x += a * np.sin(2.0 * np.pi * (f0 + k * t) * t + phi)
    x += rng.uniform(0.002, 0.05) * rng.standard_normal(n_samples)
    x = np.clip(x, -1.0, 1.0)
    if zero_block is None:
        zero_block = rng.uniform() < 0.1
    if zero_block and n_samples > sample_rate // 4:
        dur = min(int(rng.uniform(0.2, 1.0) * sample_rate), n_samples // 2)
        start = int(rng.integers(0, n_samples - dur))
        x[start:start + dur] = 0.0
    return x.astype(np.float32)


def synth_batch(seed: int, lengths, sample_rate: int = 16000):
    """List of clips with the given sample counts, all drawn from one seeded generator."""
    rng = np.random.default_rng(seed)
    return [synth_clip(rng, int(n), sample_rate) for n in lengths]


def common_voice_lengths(rng: np.random.Generator, batch: int, sample_rate: int = 16000) -> np.ndarray:
    """Clip durations ~ N(5.15, 1.77) s clipped to [1.08, 10.58] (reference notebooks/01_eda.ipynb stats)."""
    d = np.clip(rng.normal(5.15, 1.77, size=batch), 1.08, 10.58)
    return (d * sample_rate).astype(np.int64)


def synth_batch_device(seed: int, batch: int, n_samples: int, device, sample_rate: int = 16000,
                       chunk: int = 64):
    """Same signal family generated on the GPU with torch -> (batch, n_samples) float32.

    Used by bench.py for the BASELINE-sized inputs (1024 x 480 000 samples would take minutes with
    NumPy).  Parameters come from a seeded host generator; the noise from a seeded device generator.
    """
    import torch
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty((batch, n_samples), dtype=torch.float32, device=device)
    t = torch.arange(n_samples, dtype=torch.float64, device=device) / sample_rate
    for b0 in range(0, batch, chunk):
        nb = min(chunk, batch - b0)
        x = torch.zeros((nb, n_samples), dtype=torch.float64, device=device)
        for _ in range(3):
            a = torch.tensor(rng.uniform(0.05, 0.4, nb), device=device)[:, None]
            f0 = torch.tensor(rng.uniform(80.0, 3000.0, nb), device=device)[:, None]
            k = torch.tensor(rng.uniform(-200.0, 200.0, nb), device=device)[:, None]
            phi = torch.tensor(rng.uniform(0.0, 2 * np.pi, nb), device=device)[:, None]
            x += a * torch.sin(2 * np.pi * (f0 + k * t[None, :]) * t[None, :] + phi)
        sigma = torch.tensor(rng.uniform(0.002, 0.05, nb), device=device)[:, None]
        x += sigma * torch.randn((nb, n_samples), dtype=torch.float64, device=device, generator=gen)
        out[b0:b0 + nb] = x.clamp_(-1.0, 1.0).to(torch.float32)
    return out
