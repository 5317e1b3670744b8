"""Golden for SURVEY 8(f) rank 4, first half: `channel_first_latent=True` + frozen `modality_encoder` / `modality_decoder` (T:1352, T:1405-1418,
T:1481-1489, T:3094-3101, T:1826-1840) from the UNMODIFIED reference - build container only.   python -m oracle.make_golden_f4

Raw modalities are (3, L) "signals"; the encoder is a fixed Conv1d(3 -> 16, kernel 1) producing channel-first latents (16, L), the decoder its
counterpart.  Stored: the interleaved training step (loss, flow loss, gradient norms + heads), forward_modality, generate_modality_only through
the decoder, and `return_only_pred_flows` (channel-first shapes).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import detdata as D
from .ref_runner import import_reference, inject_noise
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
CFG = dict(num_text_tokens=256, dim=128, depth=2, dim_latents=(16,), heads=2, dim_head=64)
GEN_STEPS = 4


def enc_dec():
    enc, dec = nn.Conv1d(3, 16, 1), nn.Conv1d(16, 3, 1)
    with torch.no_grad():
        enc.weight.copy_(D.det_uniform('f4/enc/w', (16, 3, 1), -0.8, 0.8)); enc.bias.copy_(D.det_uniform('f4/enc/b', (16,), -0.2, 0.2))
        dec.weight.copy_(D.det_uniform('f4/dec/w', (3, 16, 1), -0.4, 0.4)); dec.bias.copy_(D.det_uniform('f4/dec/b', (3,), -0.2, 0.2))
    return enc, dec


def f4_case():
    cfg = OracleConfig(**CFG)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='f4')
    raw = lambda key, L: D.det_normalish(key, (3, L))
    batch = [[D.det_randint('f4/t0', (5,), 0, 256), raw('f4/m0', 4), D.det_randint('f4/t1', (3,), 0, 256), raw('f4/m1', 6)],
             [raw('f4/m2', 4), D.det_randint('f4/t2', (7,), 0, 256)],
             [D.det_randint('f4/t3', (9,), 0, 256)]]
    times = D.det_uniform('f4/times', (3, 2), 0.05, 0.95)
    noise = {0: D.det_normalish('f4/noise', (14, 16))}                 # flat (R, dim_latent) rows in scan order, channel-LAST (MP:642)
    xm = D.det_normalish('f4/xm', (2, 3, 5))                           # forward_modality input: raw (b, 3, L)
    nm = D.det_normalish('f4/nm', (2, 16, 5))                          # its noise, channel-first like the encoded tokens (T:2753)
    tm = torch.tensor([0.3, 0.8])
    g0 = D.det_normalish('f4/gen', (2, 4, 16))                         # generate_modality_only noise before the channel-first rearrange (T:2890)
    return cfg, sd, batch, times, noise, xm, nm, tm, g0


def make():
    tp = import_reference()
    cfg, sd, batch, times, noise, xm, nm, tm, g0 = f4_case()
    enc, dec = enc_dec()
    model = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, channel_first_latent=True, modality_default_shape=(4,),
                           modality_encoder=enc, modality_decoder=dec, modality_processing='flat', prob_uncond=0.,
                           transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    # channel-first types wrap their projections in nn.Sequential(Rearrange, Linear) / (Linear, Rearrange): T:1481-1483 -> keys `.0.1.` / `.0.0.`
    sd = {k.replace('latent_to_model_projs.0.', 'latent_to_model_projs.0.1.').replace('model_to_latent_projs.0.', 'model_to_latent_projs.0.0.'): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(('modality_encoder', 'modality_decoder')) for k in missing), (missing, unexpected)
    model.train()
    noise_cf = {0: noise[0].T.contiguous()}                              # the flat packer concatenates channel-first instances as (d, R) ('d *', T:3354)
    with inject_noise(noise_cf):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad(), inject_noise(noise_cf):
        flows = model(batch, times=times, return_only_pred_flows=True)
    model.zero_grad(set_to_none=True)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: nm.clone()
    try:
        lm = model.forward_modality(xm, times=tm)
    finally:
        torch.randn_like = orig
    with torch.no_grad():
        pm = model.forward_modality(xm, times=tm, return_loss=False)
    orig_randn = torch.randn
    torch.randn = lambda *a, **kw: g0.clone()
    try:
        gen = model.generate_modality_only(batch_size=2, fixed_modality_shape=(4,), modality_steps=GEN_STEPS)
    finally:
        torch.randn = orig_randn
    out = dict(reference='lucidrains/transfusion-pytorch v0.19.4, channel_first_latent + frozen Conv1d encoder / decoder, fp32, CPU',
               loss=loss.detach().double(), text_loss=bd.text.detach().double(), flow_losses=[f.detach().double() for f in bd.flow],
               grad_norms={k: float(v.double().norm()) for k, v in grads.items()}, grad_head={k: v.reshape(-1)[:1024].clone() for k, v in grads.items()},
               pred_flow_shapes=[tuple(f.shape) for f in flows[0]], pred_flow0=flows[0][0].detach().clone(),
               mod_loss=lm.detach().double(), mod_pred_noloss=pm.detach().clone(), gen=gen.detach().clone(), gen_steps=GEN_STEPS)
    path = os.path.join(OUT, 'f4_chfirst.pt')
    torch.save(out, path)
    print(f'f4_chfirst: loss {float(loss):.6f} flow {[float(f) for f in bd.flow]} mod {float(lm):.6f} shapes {out["pred_flow_shapes"]} gen {tuple(gen.shape)} '
          f'({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    make()
