"""Generate tests/golden/*.pt from the UNMODIFIED reference (run in the build container only).

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden tiny1

TEST INFRASTRUCTURE ONLY.  Stores the reference's outputs on the deterministic
inputs of oracle/cases.py: losses, logits (sub-sampled rows for the big case),
final embed rows, per-parameter gradient norms + a slice of every gradient
(full gradients for the small cases), and an input checksum.
"""
from __future__ import annotations

import os
import sys
import time

import torch

from .cases import CASES, build_case, default_shapes, input_checksum
from .ref_runner import reference_forward_backward

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def make(name: str):
    cfg, sd, batch, times, noise = build_case(name)
    t0 = time.time()
    ref, model = reference_forward_backward(cfg, sd, batch, times, noise, modality_default_shape=default_shapes(cfg))
    dt = time.time() - t0
    big = cfg.dim >= 256
    row_step = 8 if big else 1
    g = dict(
        case=name,
        reference='lucidrains/transfusion-pytorch v0.19.4, modality_processing=flat, fp32, CPU',
        input_checksum=input_checksum(sd, batch, times, noise),
        loss=ref['loss'].double(), text_loss=ref['text_loss'].double(),
        flow_losses=[f.double() for f in ref['flow_losses']],
        row_step=row_step,
        logits=ref['logits'][:, ::row_step].clone(),
        embed=ref['embed'][:, ::row_step].clone(),
        grad_norms={k: float(v.double().norm()) for k, v in ref['grads'].items()},
        grad_head={k: v.reshape(-1)[:256].clone() for k, v in ref['grads'].items()},
    )
    if cfg.dim < 128:
        g['grads'] = {k: v.clone() for k, v in ref['grads'].items()}
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f'{name}.pt')
    torch.save(g, path)
    print(f'{name}: loss {float(g["loss"]):.6f} text {float(g["text_loss"]):.6f} flow {[float(f) for f in g["flow_losses"]]} '
          f'({dt:.1f}s reference, {os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    names = sys.argv[1:] or list(CASES)
    for n in names:
        make(n)
