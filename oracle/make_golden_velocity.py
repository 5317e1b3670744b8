"""Generate tests/golden/velocity1.pt from the UNMODIFIED reference: `Transfusion.forward` with `velocity_consistency_ema_model`
(T:3084-3088, T:3378-3418) - build container only.   python -m oracle.make_golden_velocity

TEST INFRASTRUCTURE ONLY.  Student and teacher are two reference models with different deterministic weights; the student's
noise and the teacher's noise are injected through `torch.randn_like` in call order (student types first, then the teacher's).
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .cases import CASES, default_shapes
from .ref_runner import build_reference_model
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
DELTA = 1e-3


def velocity_case():
    kw, _, b = CASES['small2']
    cfg = OracleConfig(**kw)
    batch = D.ragged_batch('velocity1/b', b, cfg.num_text_tokens, cfg.dim_latents)
    times = D.det_times('velocity1/t', batch) * 0.9 + 0.02
    noise = D.det_noise('velocity1/n', batch, cfg.num_modalities)
    noise_t = D.det_noise('velocity1/nt', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='velocity1/student')
    sd_t = D.det_state_dict(cfg.state_dict_shapes(), tag='velocity1/teacher')
    return cfg, sd, sd_t, batch, times, noise, noise_t


def make():
    cfg, sd, sd_t, batch, times, noise, noise_t = velocity_case()
    student = build_reference_model(cfg, sd, default_shapes(cfg)); student.train()
    teacher = build_reference_model(cfg, sd_t, default_shapes(cfg)); teacher.eval()
    queue = [noise[t] for t in sorted(noise)] + [noise_t[t] for t in sorted(noise_t)]
    orig = torch.randn_like

    def fake(t, *a, **k):
        v = queue.pop(0)
        assert tuple(v.shape) == tuple(t.shape), (v.shape, t.shape)
        return v.to(t)

    torch.randn_like = fake
    try:
        loss, bd = student(batch, times=times, velocity_consistency_ema_model=teacher, velocity_consistency_delta_time=DELTA, return_breakdown=True)
    finally:
        torch.randn_like = orig
    assert not queue
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in student.named_parameters() if p.grad is not None}
    g = dict(reference='lucidrains/transfusion-pytorch v0.19.4 forward(velocity_consistency_ema_model=...), fp32, CPU', delta=DELTA,
             loss=loss.detach().double(), text_loss=bd.text.detach().double(), flow_losses=[f.detach().double() for f in bd.flow],
             velocity_losses=[v.detach().double() for v in bd.velocity],
             grad_norms={k: float(v.double().norm()) for k, v in grads.items()},
             grad_head={k: v.reshape(-1)[:1024].clone() for k, v in grads.items()})
    path = os.path.join(OUT, 'velocity1.pt')
    torch.save(g, path)
    print(f'velocity1: loss {float(loss):.6f} velocity {[float(v) for v in bd.velocity]} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    make()
