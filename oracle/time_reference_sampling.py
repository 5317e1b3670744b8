"""Time the UNMODIFIED reference's `sample_many` on the reduced configuration of SURVEY.md section 8(d) (dim512/depth8, 8 prompts of the four
README kinds, max_length 32, 16 ODE grid points, cfg 3, greedy) on this container's CPU cores and write tests/golden/reference_sampling_time.json
(`bench.py --sample` prints it beside the native numbers; the reference tree does not exist on the GPU box).  Build container only:

    python -m oracle.time_reference_sampling
"""
from __future__ import annotations

import json
import os
import time

import torch

from .ref_runner import import_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'reference_sampling_time.json')


def main():
    tp = import_reference()
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    model = tp.Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).eval()
    g = torch.Generator().manual_seed(1234)
    prompts = []
    for _ in range(2):
        prompts += [torch.randint(0, 256, (16,), generator=g), (0, torch.randn(4, 384, generator=g)), None,
                    [torch.randint(0, 256, (8,), generator=g), (0, torch.randn(6, 384, generator=g))]]
    noise = torch.randn(16, 384, generator=g)
    res = {}
    for force in (None, 0):
        kw = dict(max_length=32, modality_steps=16, cfg_scale=3., text_temperature=0., init_modality_noise=noise, fixed_modality_shape=(4,))
        if force is not None:
            kw['force_modality_at_start'] = force
        t0 = time.perf_counter()
        out = model.sample_many([p if not isinstance(p, list) else list(p) for p in prompts], **kw)
        dt = time.perf_counter() - t0
        nmod = sum(sum(isinstance(p, tuple) for p in s) for s in out)
        res['reduced_forced' if force is not None else 'reduced'] = {'seconds': dt, 'modality_instances': nmod}
        print(force, dt, nmod)
    json.dump({'kind': 'reference', 'where': f'build container, {os.cpu_count()} vCPU, torch {torch.__version__} CPU fp32 (not the GPU box)',
               'config': 'dim=512 depth=8, 8 prompts (2 x the four README kinds), max_length=32, modality_steps=16, cfg_scale=3, greedy', 'runs': res},
              open(OUT, 'w'), indent=1)


if __name__ == '__main__':
    main()
