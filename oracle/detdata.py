"""Deterministic data / weight factory shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Everything is derived from a hand-written splitmix64 counter generator so the
numbers are identical on every machine, independent of torch / numpy RNG
streams.  Values are uniform; weights are scaled so activations stay O(1).
"""
from __future__ import annotations

import zlib
import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def det_uniform(key: str | int, shape, lo=-1.0, hi=1.0) -> torch.Tensor:
    """uniform [lo, hi) float32 tensor, a pure function of (key, shape)."""
    seed = zlib.crc32(key.encode()) if isinstance(key, str) else int(key)
    n = int(np.prod(shape)) if len(shape) else 1
    ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
    bits = _splitmix64(_splitmix64(ctr))
    u = (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24 random bits -> exactly representable in fp32
    out = (lo + (hi - lo) * u).astype(np.float32).reshape(shape)
    return torch.from_numpy(out)


def det_normalish(key, shape, std=1.0) -> torch.Tensor:
    """sum of 4 uniforms: bell shaped, unit-ish variance * std (used for latents / noise)."""
    acc = sum(det_uniform(f'{key}/{i}', shape) for i in range(4))
    return acc * (std * (3.0 / 4.0) ** 0.5)


def det_randint(key, shape, lo, hi) -> torch.Tensor:
    u = det_uniform(key, shape, 0.0, 1.0).double()
    return (lo + torch.floor(u * (hi - lo))).clamp(max=hi - 1).long()


# --------------------------------------------------------------------------
# weights: fill a reference-shaped state_dict deterministically (and de-zero
# the zero-init parameters, which would otherwise hide bugs - SURVEY.md §7.1)
# --------------------------------------------------------------------------

def det_state_dict(shapes: dict[str, tuple], tag: str = 'w') -> dict[str, torch.Tensor]:
    """shapes: name -> shape (reference `state_dict` keys).  Returns fp32 tensors."""
    sd = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name == 'rotary_emb.freqs':
            dim = shape[0] * 2
            sd[name] = 1. / (10000 ** (torch.arange(0, dim, 2).float() / dim))
            continue
        if name.endswith('to_time_cond.0.weights'):
            sd[name] = det_normalish(f'{tag}/{name}', shape)  # fixed random fourier buffer
            continue
        fan_in = shape[-1] if len(shape) > 1 else None
        if name.endswith('to_ada_ln_zero.bias'):
            sd[name] = -2.0 + det_uniform(f'{tag}/{name}', shape, -0.5, 0.5)
        elif name.endswith(('gamma', 'layernorm_gamma', 'layerscale')):
            sd[name] = det_uniform(f'{tag}/{name}', shape, -0.2, 0.2)
        elif name.endswith('pseudo_queries'):
            sd[name] = det_uniform(f'{tag}/{name}', shape, -0.5, 0.5)
        elif name.endswith('bias'):
            sd[name] = det_uniform(f'{tag}/{name}', shape, -0.1, 0.1)
        elif name.endswith(('to_film.weight', 'to_ada_ln_zero.weight')):
            sd[name] = det_uniform(f'{tag}/{name}', shape, -1.0, 1.0) * (0.5 / fan_in ** 0.5)
        elif name in ('text_embed.weight',):
            sd[name] = det_uniform(f'{tag}/{name}', shape, -1.0, 1.0)
        elif fan_in is not None:
            sd[name] = det_uniform(f'{tag}/{name}', shape, -1.0, 1.0) * (1.7 / fan_in ** 0.5)
        else:
            sd[name] = det_uniform(f'{tag}/{name}', shape, -0.5, 0.5)
    return sd


# --------------------------------------------------------------------------
# synthetic batches
# --------------------------------------------------------------------------

def canonical_sample(key: str, num_text_tokens=256, dim_latent=384, n_inst=32, latent_len=4, text_len=24, last_text_len=23):
    """64 parts alternating text(24) / latent(4,384); packs to 1025 tokens."""
    parts = []
    for i in range(n_inst):
        tl = text_len if i < n_inst - 1 else last_text_len
        parts.append(det_randint(f'{key}/t{i}', (tl,), 0, num_text_tokens))
        parts.append(det_normalish(f'{key}/l{i}', (latent_len, dim_latent)))
    return parts


def canonical_batch(batch: int, key='canon', **kw):
    return [canonical_sample(f'{key}/{b}', **kw) for b in range(batch)]


def two_modality_sample(key: str, num_text_tokens=256, dim_latents=(384, 192), lens=(4, 2), n_inst=32, text_len=25, last_text_len=24):
    """even instances type 0 (4,384), odd type 1 (2,192); packs to 1025."""
    parts = []
    for i in range(n_inst):
        tl = text_len if i < n_inst - 1 else last_text_len
        parts.append(det_randint(f'{key}/t{i}', (tl,), 0, num_text_tokens))
        ty = i % 2
        parts.append((ty, det_normalish(f'{key}/l{i}', (lens[ty], dim_latents[ty]))))
    return parts


def ragged_batch(key: str, batch: int, num_text_tokens: int, dim_latents, lens_choices=((3,), (5,), (2,))):
    """small ragged batch: varying text lengths / instance counts / latent lengths, incl. a
    text-only sample and a sample that starts with a modality."""
    out = []
    for b in range(batch):
        parts = []
        n_inst = [2, 0, 3, 1][b % 4]
        if b % 4 == 2:
            n_text_first = 0
        else:
            n_text_first = 3 + b
        if n_text_first:
            parts.append(det_randint(f'{key}/{b}/t0', (n_text_first,), 0, num_text_tokens))
        for i in range(n_inst):
            ty = (i + b) % len(dim_latents)
            L = lens_choices[(i + b) % len(lens_choices)][0]
            lat = det_normalish(f'{key}/{b}/l{i}', (L, dim_latents[ty]))
            parts.append((ty, lat) if len(dim_latents) > 1 else lat)
            parts.append(det_randint(f'{key}/{b}/t{i + 1}', (2 + (i * 3 + b) % 5,), 0, num_text_tokens))
        if n_inst == 0:
            parts.append(det_randint(f'{key}/{b}/tx', (6,), 0, num_text_tokens))
        out.append(parts)
    return out


def count_instances(batch) -> list[int]:
    return [sum(1 for p in s if isinstance(p, tuple) or (torch.is_tensor(p) and p.is_floating_point())) for s in batch]


def det_times(key: str, batch) -> torch.Tensor:
    counts = count_instances(batch)
    m = max(counts) if counts else 0
    return det_uniform(key, (len(batch), m), 0.02, 0.98)


def det_noise(key: str, batch, num_modalities: int) -> dict[int, torch.Tensor]:
    """noise per modality type, flat (R_type, dim_latent) in scan order (batch-major, then in-sample
    order) - the order `process_type_flat` concatenates instances in (MP:642)."""
    rows = {t: [] for t in range(num_modalities)}
    for s in batch:
        for p in s:
            if isinstance(p, tuple):
                rows[p[0]].append(p[1])
            elif torch.is_tensor(p) and p.is_floating_point():
                rows[0].append(p)
    out = {}
    for t, lst in rows.items():
        if not lst:
            continue
        R = sum(x.shape[0] for x in lst)
        out[t] = det_normalish(f'{key}/n{t}', (R, lst[0].shape[-1]))
    return out
