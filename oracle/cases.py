"""Named parity cases: config + deterministic inputs.  TEST INFRASTRUCTURE ONLY.

Each case is a pure function of its name (oracle/detdata.py), so the golden
fixture only has to store the REFERENCE's outputs plus input checksums.
"""
from __future__ import annotations

import torch

from . import detdata as D
from .transfusion_oracle import OracleConfig

CASES = {
    # name: (cfg kwargs, batch kind, batch size)
    'tiny1':   (dict(num_text_tokens=256, dim=64,  depth=2, dim_latents=(48,),      heads=2, dim_head=64), 'ragged', 4),
    'small2':  (dict(num_text_tokens=256, dim=128, depth=4, dim_latents=(32, 16),   heads=2, dim_head=64), 'ragged', 4),
    'mid2':    (dict(num_text_tokens=256, dim=256, depth=4, dim_latents=(384, 192), heads=4, dim_head=64), 'two_modality', 2),
    'head8':   (dict(num_text_tokens=256, dim=128, depth=2, dim_latents=(16,),      heads=4, dim_head=8),  'ragged', 3),   # dim_head of train_toy.py / the reference's tests
    'canon512': (dict(num_text_tokens=256, dim=512, depth=8, dim_latents=(384,),    heads=8, dim_head=64), 'canonical', 2),
    # the dimensions of BASELINE configs 3 / 4: depth 24 = AttentionResidual over 25 hiddens + 12 skip pairs; two modality types at dim 768
    'cfg3_1024': (dict(num_text_tokens=256, dim=1024, depth=24, dim_latents=(384,),     heads=8, dim_head=64), 'canonical', 1),
    'cfg4_768':  (dict(num_text_tokens=256, dim=768,  depth=16, dim_latents=(384, 192), heads=8, dim_head=64), 'two_modality', 1),
}

# pure-text cases (Transfusion.forward_text): name -> (cfg kwargs, batch, tokens per row incl. the shifted one)
TEXT_CASES = {
    'text1': (dict(num_text_tokens=256, dim=128, depth=4, dim_latents=(32,), heads=2, dim_head=64), 3, 98),
}


def build_text_case(name: str):
    kw, b, n1 = TEXT_CASES[name]
    cfg = OracleConfig(**kw)
    text = D.det_randint(f'{name}/text', (b, n1), 0, cfg.num_text_tokens)
    text[1, -17:] = -1                                     # a padded row: ignore_index labels, id 0 embedded (T:2608)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=name)
    return cfg, sd, text


# pure-modality cases (Transfusion.forward_modality): name -> (cfg kwargs, batch, axial shape, modality type)
MODALITY_CASES = {
    'flow1': (dict(num_text_tokens=256, dim=128, depth=4, dim_latents=(48, 32), heads=2, dim_head=64), 3, (5, 7), 1),
}


def build_modality_case(name: str):
    kw, b, shape, ty = MODALITY_CASES[name]
    cfg = OracleConfig(**kw)
    dl = cfg.dim_latents[ty]
    x = D.det_normalish(f'{name}/x', (b, *shape, dl))
    noise = D.det_normalish(f'{name}/n', (b, *shape, dl))
    times = D.det_uniform(f'{name}/t', (b,), 0.05, 0.95)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=name)
    return cfg, sd, x, times, noise, ty


TRAINABLE_EXCLUDE = ('rotary_emb.freqs', 'transformer.to_time_cond.0.weights')


def default_shapes(cfg: OracleConfig):
    if cfg.num_modalities == 1:
        return (4,)
    return tuple((4,) if i == 0 else (2,) for i in range(cfg.num_modalities))


def build_case(name: str):
    kw, kind, b = CASES[name]
    cfg = OracleConfig(**kw)
    if kind == 'ragged':
        batch = D.ragged_batch(f'{name}/b', b, cfg.num_text_tokens, cfg.dim_latents)
    elif kind == 'canonical':
        batch = D.canonical_batch(b, key=f'{name}/b', num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents[0])
    elif kind == 'two_modality':
        batch = [D.two_modality_sample(f'{name}/b/{i}', cfg.num_text_tokens, cfg.dim_latents) for i in range(b)]
    else:
        raise KeyError(kind)
    times = D.det_times(f'{name}/t', batch)
    noise = D.det_noise(f'{name}/n', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=name)
    return cfg, sd, batch, times, noise


def input_checksum(sd, batch, times, noise) -> float:
    acc = 0.0
    for k in sorted(sd):
        acc += float(sd[k].double().abs().sum())
    for s in batch:
        for p in s:
            t = p[1] if isinstance(p, tuple) else p
            acc += float(t.double().abs().sum())
    acc += float(times.double().sum())
    for t in sorted(noise):
        acc += float(noise[t].double().abs().sum())
    return acc


def with_grad(sd):
    return {k: (v.clone().requires_grad_(True) if k not in TRAINABLE_EXCLUDE else v.clone()) for k, v in sd.items()}
