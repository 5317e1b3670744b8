"""Generate tests/golden/flow*.pt from the UNMODIFIED reference's `forward_modality` / `generate_modality_only`
(build container only).

    python -m oracle.make_golden_modality

TEST INFRASTRUCTURE ONLY.  Stores loss, predicted flow, gradient norms + heads of the pure flow path (T:2710-2869) on
the deterministic inputs of oracle/cases.py (noise injected through `torch.randn_like`), and the result of
`generate_modality_only` (T:2871-2923, midpoint ODE) started from a deterministic noise tensor (`torch.randn` patched).
"""
from __future__ import annotations

import os
import sys

import torch

from . import detdata as D
from .cases import MODALITY_CASES, build_modality_case, default_shapes
from .ref_runner import build_reference_model

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
GEN_STEPS = 5
VC_DELTA = 1e-3


def gen_noise(name, b, shape, dl):
    return D.det_normalish(f'{name}/gen', (b, *shape, dl))


def make(name: str):
    cfg, sd, x, times, noise, ty = build_modality_case(name)
    shape = MODALITY_CASES[name][2]
    model = build_reference_model(cfg, sd, tuple(shape if i == ty else (2,) * len(shape) for i in range(cfg.num_modalities)))
    model.train()
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: noise.clone()
    try:
        loss = model.forward_modality(x, times=times, modality_type=ty)
    finally:
        torch.randn_like = orig
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad():
        pred = model.forward_modality(x, times=times, modality_type=ty, return_loss=False)   # no noising: f(x, t)
    g0 = gen_noise(name, 2, shape, cfg.dim_latents[ty])
    orig_randn = torch.randn
    torch.randn = lambda *a, **kw: g0.clone()
    try:
        sampled = model.generate_modality_only(batch_size=2, modality_type=ty, fixed_modality_shape=tuple(shape), modality_steps=GEN_STEPS)
    finally:
        torch.randn = orig_randn
    # forward_modality with an EMA teacher (T:2716-2859): the student is noised at t (1 - delta), the teacher predicts at t + delta on the
    # CLEAN input, and the term is mse(flow target, teacher flow) - constant in the student's parameters (same gradients, larger loss)
    sd_t = D.det_state_dict(cfg.state_dict_shapes(), tag=f'{name}/teacher')
    teacher = build_reference_model(cfg, sd_t, tuple(shape if i == ty else (2,) * len(shape) for i in range(cfg.num_modalities)))
    teacher.eval()
    model.zero_grad(set_to_none=True)
    torch.randn_like = lambda t, **kw: noise.clone()
    try:
        vloss, (vflow, vvel, _) = model.forward_modality(x, times=times, modality_type=ty, velocity_consistency_ema_model=teacher,
                                                         velocity_consistency_delta_time=VC_DELTA, return_loss_breakdown=True)
    finally:
        torch.randn_like = orig
    g = dict(vc_delta=VC_DELTA, vc_loss=vloss.detach().double(), vc_flow=vflow.detach().double(), vc_velocity=vvel.detach().double(),
             case=name, reference='lucidrains/transfusion-pytorch v0.19.4 forward_modality / generate_modality_only, fp32, CPU',
             input_checksum=float(x.double().abs().sum() + noise.double().abs().sum() + times.double().sum()),
             loss=loss.detach().double(), pred_noloss=pred.detach().clone(),
             grad_norms={k: float(v.double().norm()) for k, v in grads.items()},
             grad_head={k: v.reshape(-1)[:1024].clone() for k, v in grads.items()},
             gen_steps=GEN_STEPS, gen=sampled.detach().clone())
    path = os.path.join(OUT, f'{name}.pt')
    torch.save(g, path)
    print(f'{name}: loss {float(g["loss"]):.6f}  {len(grads)} gradients  ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    for n in (sys.argv[1:] or list(MODALITY_CASES)):
        make(n)
