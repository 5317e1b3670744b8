"""Golden for `reconstruction_loss_weight > 0` (T:1522-1525; interleaved: MP:177-200 closures + T:3420-3431; forward_modality: T:2840-2853) from the
UNMODIFIED reference - build container only.   python -m oracle.make_golden_recon

  inter   the `small2` case (two modality types) with reconstruction_loss_weight = 0.1: loss, per-type mean of the per-instance reconstruction
          losses, gradient norms + heads
  fm      forward_modality without encoder / decoder: the term carries gradient (target = the clean latent)
  fm_dec  forward_modality with the frozen encoder / decoder of the f4 golden: the decoder runs under no_grad - a reported value
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .cases import build_case, default_shapes
from .make_golden_f4 import enc_dec, f4_case
from .make_golden_f4b import grads_of, patched, summarize
from .ref_runner import import_reference, inject_noise

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
WEIGHT = 0.1


def make():
    tp = import_reference()
    cfg, sd, batch, times, noise = build_case('small2')
    model = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents, modality_default_shape=default_shapes(cfg), reconstruction_loss_weight=WEIGHT,
                           modality_processing='flat', prob_uncond=0.,
                           transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict(sd, strict=True)
    model.train()
    with inject_noise(noise):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    gn, gh = summarize(grads_of(model))
    gh.update({k: g.reshape(-1)[:64].clone() for k, g in grads_of(model).items() if k.endswith(('to_out.1.weight', 'net.3.weight')) and '.0.' in k})
    recon = [torch.stack(r).mean() if len(r) else torch.zeros(()) for r in bd.recon]
    out = dict(loss=loss.detach(), text_loss=bd.text.detach(), flow_losses=[f.detach() for f in bd.flow], recon=[r.detach() for r in recon], grad_norms=gn, grad_heads=gh)
    model.zero_grad(set_to_none=True)
    # forward_modality, type 0, no encoder / decoder
    xm = D.det_normalish('recon/xm', (2, 5, cfg.dim_latents[0])); nm = D.det_normalish('recon/nm', (2, 5, cfg.dim_latents[0])); tm = torch.tensor([0.3, 0.8])
    with patched('randn_like', [nm]):
        lm, (fl, vl, rl) = model.forward_modality(xm, times=tm, modality_type=0, return_loss_breakdown=True)
    lm.backward()
    gn_m, gh_m = summarize(grads_of(model))
    out.update(fm_loss=lm.detach(), fm_flow=fl.detach(), fm_recon=rl.detach(), fm_grad_norms=gn_m, fm_grad_heads=gh_m)
    # forward_modality with the frozen conv1d encoder / decoder of the f4 golden (channel-first latents)
    cfg4, sd4, _, _, _, xm4, nm4, tm4, _ = f4_case()
    enc, dec = enc_dec()
    m4 = tp.Transfusion(num_text_tokens=cfg4.num_text_tokens, dim_latent=16, channel_first_latent=True, modality_default_shape=(4,), modality_encoder=enc, modality_decoder=dec,
                        reconstruction_loss_weight=WEIGHT, modality_processing='flat', prob_uncond=0., transformer=dict(dim=cfg4.dim, depth=cfg4.depth, dim_head=cfg4.dim_head, heads=cfg4.heads))
    sd4 = {k.replace('latent_to_model_projs.0.', 'latent_to_model_projs.0.1.').replace('model_to_latent_projs.0.', 'model_to_latent_projs.0.0.'): v for k, v in sd4.items()}
    missing, unexpected = m4.load_state_dict(sd4, strict=False)
    assert not unexpected
    m4.train()
    with patched('randn_like', [nm4]):
        l4, (f4, v4, r4) = m4.forward_modality(xm4, times=tm4, return_loss_breakdown=True)
    out.update(dec_loss=l4.detach(), dec_flow=f4.detach(), dec_recon=r4.detach())
    torch.save(out, os.path.join(OUT, 'recon1.pt'))
    print('interleaved: loss', float(loss), 'recon', [float(r) for r in recon], '| forward_modality: loss', float(lm), 'recon', float(rl), '| with decoder: loss', float(l4), 'recon', float(r4))


def recon_inputs():
    cfg, *_ = build_case('small2')
    return D.det_normalish('recon/xm', (2, 5, cfg.dim_latents[0])), D.det_normalish('recon/nm', (2, 5, cfg.dim_latents[0])), torch.tensor([0.3, 0.8])


if __name__ == '__main__':
    make()
