"""Run the UNMODIFIED reference (`/root/reference`) through `oracle/shims`.

TEST INFRASTRUCTURE ONLY.  Works only where `/root/reference` exists (the build
container); the GPU box never imports this module - it uses the committed
`tests/golden/*.pt` vectors this module produced (see make_golden.py).
"""
from __future__ import annotations

import os
import sys
from contextlib import contextmanager

import torch

REF_ROOT = os.environ.get('TFX_REFERENCE_ROOT', '/root/reference')
SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shims')


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, 'transfusion_pytorch'))


def import_reference():
    assert reference_available(), f'{REF_ROOT} not present'
    for p in (SHIMS, REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import transfusion_pytorch  # noqa
    return transfusion_pytorch


def build_reference_model(cfg, sd, modality_default_shape=None):
    """cfg: oracle OracleConfig; sd: reference-keyed state_dict to load (strict)."""
    tp = import_reference()
    dl = cfg.dim_latents if len(cfg.dim_latents) > 1 else cfg.dim_latents[0]
    model = tp.Transfusion(
        num_text_tokens=cfg.num_text_tokens,
        dim_latent=dl,
        modality_default_shape=modality_default_shape,
        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads),
        modality_processing='flat',     # never 'naive' (times bug MP:390) / 'auto' (timing dependent) - SURVEY §8c
        prob_uncond=0.,
        model_output_clean=getattr(cfg, 'model_output_clean', False), eps=getattr(cfg, 'eps', 1e-2),
    )
    missing, unexpected = model.load_state_dict(sd, strict=True)
    return model


@contextmanager
def inject_noise(noise_by_type: dict):
    """patch `torch.randn_like` so `process_type_flat` (MP:654) draws our noise.  The flat strategy
    calls randn_like once per modality type on the (R, dl) concatenation; we match by shape."""
    orig = torch.randn_like
    pool = {tuple(v.shape): v for v in noise_by_type.values()}
    assert len(pool) == len(noise_by_type), 'noise shapes must be unique per type'

    def fake(t, *a, **k):
        key = tuple(t.shape)
        assert key in pool, f'unexpected randn_like shape {key}'
        return pool[key].to(t)

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def reference_forward_backward(cfg, sd, batch, times, noise, modality_default_shape=None, backward=True):
    model = build_reference_model(cfg, sd, modality_default_shape)
    model.train()
    captured = {}
    # capture the TRAINING path's logits / final embed with forward hooks (T:3280, T:1250)
    h1 = model.to_text_logits.register_forward_hook(lambda m, i, o: captured.__setitem__('logits', o.detach()))
    h2 = model.transformer.norm.register_forward_hook(lambda m, i, o: captured.__setitem__('embed', o.detach()))
    with inject_noise(noise):
        loss, breakdown = model(batch, times=times, return_breakdown=True)
    h1.remove(); h2.remove()
    out = dict(loss=loss.detach(), text_loss=breakdown.text.detach(),
               flow_losses=[f.detach() for f in breakdown.flow], **captured)
    if backward:
        loss.backward()
        out['grads'] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return out, model
