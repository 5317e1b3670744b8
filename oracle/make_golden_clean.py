"""Generate tests/golden/clean1.pt from the UNMODIFIED reference with `model_output_clean=True` (T:1297, MP:100-126): the
interleaved training step and `forward_modality` - build container only.   python -m oracle.make_golden_clean

TEST INFRASTRUCTURE ONLY.  Times are kept away from 1 except for one instance at 0.995, which exercises the `eps` floor.
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .cases import CASES, default_shapes
from .ref_runner import build_reference_model, inject_noise
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def clean_case():
    kw, _, b = CASES['small2']
    cfg = OracleConfig(**kw, model_output_clean=True, eps=1e-2)
    batch = D.ragged_batch('clean1/b', b, cfg.num_text_tokens, cfg.dim_latents)
    times = D.det_times('clean1/t', batch) * 0.9 + 0.02
    times[0, 0] = 0.995                                           # 1 - t = 0.005 < eps: the clamp is active
    noise = D.det_noise('clean1/n', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='clean1')
    xm = D.det_normalish('clean1/xm', (3, 6, cfg.dim_latents[1]))
    nm = D.det_normalish('clean1/nm', (3, 6, cfg.dim_latents[1]))
    tm = torch.tensor([0.3, 0.995, 0.71])
    return cfg, sd, batch, times, noise, xm, tm, nm


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def make():
    cfg, sd, batch, times, noise, xm, tm, nm = clean_case()
    model = build_reference_model(cfg, sd, default_shapes(cfg)); model.train()
    with inject_noise(noise):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    g1 = grads_of(model)
    model.zero_grad(set_to_none=True)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: nm.clone()
    try:
        lm = model.forward_modality(xm, times=tm, modality_type=1)
    finally:
        torch.randn_like = orig
    lm.backward()
    g2 = grads_of(model)
    with torch.no_grad():
        pm = model.forward_modality(xm, times=tm, modality_type=1, return_loss=False)
    pack = lambda g: (dict((k, float(v.double().norm())) for k, v in g.items()), dict((k, v.reshape(-1)[:1024].clone()) for k, v in g.items()))
    n1, h1 = pack(g1); n2, h2 = pack(g2)
    out = dict(reference='lucidrains/transfusion-pytorch v0.19.4, model_output_clean=True, eps=1e-2, fp32, CPU',
               loss=loss.detach().double(), text_loss=bd.text.detach().double(), flow_losses=[f.detach().double() for f in bd.flow],
               grad_norms=n1, grad_head=h1, mod_loss=lm.detach().double(), mod_grad_norms=n2, mod_grad_head=h2, mod_pred_noloss=pm.clone())
    path = os.path.join(OUT, 'clean1.pt')
    torch.save(out, path)
    print(f'clean1: loss {float(loss):.6f} flows {[float(f) for f in bd.flow]}; forward_modality {float(lm):.6f} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    make()
