"""Golden vectors for the decode path: the UNMODIFIED reference's `sample_many` (and `sample_one`) on deterministic
weights / prompts / initial noise, greedy text (temperature 0).  Build container only.

    python -m oracle.make_golden_sampling
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .cases import default_shapes
from .ref_runner import build_reference_model
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

CFG = dict(num_text_tokens=256, dim=128, depth=4, dim_latents=(32,), heads=2, dim_head=64)


def sampling_case():
    cfg = OracleConfig(**CFG)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='sampling')
    prompts = [
        D.det_randint('sp/p0', (12,), 0, 256),                                   # text prompt
        (0, D.det_normalish('sp/p1', (4, 32))),                                  # raw modality prompt
        None,                                                                    # empty prompt
        [D.det_randint('sp/p3', (6,), 0, 256), (0, D.det_normalish('sp/p3m', (3, 32)))],   # list prompt ending in a modality
    ]
    noise = D.det_normalish('sp/noise', (8, 32))
    return cfg, sd, prompts, noise


def to_plain(sample):
    out = []
    for p in sample:
        if isinstance(p, tuple):
            out.append(('mod', int(p[0]), p[1].detach().clone()))
        else:
            out.append(('text', p.detach().clone().long()))
    return out


def main():
    cfg, sd, prompts, noise = sampling_case()
    model = build_reference_model(cfg, sd, modality_default_shape=(4,))
    model.eval()
    g = dict(cfg=CFG, runs={})
    for name, kw in [('free', dict()), ('forced', dict(force_modality_at_start=0)), ('forced_nocfg', dict(force_modality_at_start=0, cfg_scale=1.))]:
        kwargs = dict(max_length=12, text_temperature=0., init_modality_noise=noise, modality_steps=4, fixed_modality_shape=(4,), cfg_scale=3.)
        kwargs.update(kw)
        outs = model.sample_many([p if not isinstance(p, list) else list(p) for p in prompts], **kwargs)
        g['runs'][name] = [to_plain(o) for o in outs]
        for i, o in enumerate(outs):
            desc = [('mod', tuple(p[1].shape)) if isinstance(p, tuple) else p.tolist() for p in o]
            print(name, i, desc)
    torch.save(g, os.path.join(OUT, 'sampling.pt'))
    print('saved', os.path.getsize(os.path.join(OUT, 'sampling.pt')), 'bytes')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    main()
