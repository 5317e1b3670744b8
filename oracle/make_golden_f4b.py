"""Goldens for SURVEY 8(f) rank 4, second half: `add_pos_emb` (T:1384-1403, T:2781-2796, T:3173-3180) and learnable
`pre_post_transformer_enc_dec` pairs (the paper's U-Net down / up, T:1451-1494, MP:715-745) from the UNMODIFIED reference - build container
only.   python -m oracle.make_golden_f4b

Two models:
  pos   one channel-last modality type (dim_latent 16, 2 axial dims) with the axial positional embedding: interleaved training step,
        forward_modality, greedy `sample_one` (the reference's sampler that adds the embedding during the ODE steps, through forward())
  unet  one channel-first type (4, H, W) with Conv2d(4 -> dim, 3, stride 2) / ConvTranspose2d(dim -> 4) around the transformer AND the positional
        embedding on the down-sampled grid: interleaved training step with text, forward_modality, generate_modality_only
The positional-embedding module on both sides is the repo's restatement (oracle/shims/axial_positional_embedding): its arithmetic is unpinned,
everything around it is the reference's.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import detdata as D
from .ref_runner import import_reference
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
POS_CFG = dict(num_text_tokens=256, dim=128, depth=2, dim_latents=(16,), heads=2, dim_head=64)
UNET_CFG = dict(num_text_tokens=256, dim=128, depth=2, dim_latents=(4,), heads=2, dim_head=64)
GEN_STEPS = 3


def fill_module_(mod, tag, scale=0.3):
    """deterministic values for every parameter of a torch module (the positional-embedding MLPs, the conv pair)"""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            p.copy_(D.det_uniform(f'{tag}/{name}', tuple(p.shape), -scale, scale))


def base_sd(cfg, tag, drop_proj=False):
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=tag)
    if drop_proj:
        sd = {k: v for k, v in sd.items() if not k.startswith(('latent_to_model_projs', 'model_to_latent_projs'))}
    return sd


def pos_case():
    cfg = OracleConfig(**POS_CFG)
    sd = base_sd(cfg, 'f4b/pos')
    lat = lambda key, shape: D.det_normalish(key, (*shape, 16))
    batch = [[D.det_randint('f4b/t0', (5,), 0, 256), (0, lat('f4b/m0', (2, 3))), D.det_randint('f4b/t1', (3,), 0, 256), (0, lat('f4b/m1', (3, 2)))],
             [(0, lat('f4b/m2', (2, 3))), D.det_randint('f4b/t2', (7,), 0, 256)],
             [D.det_randint('f4b/t3', (9,), 0, 256)]]
    times = D.det_uniform('f4b/times', (3, 2), 0.05, 0.95)
    noise = D.det_normalish('f4b/noise', (18, 16))                     # flat (R, dim_latent) rows in scan order (MP:654)
    xm = D.det_normalish('f4b/xm', (2, 2, 3, 16)); nm = D.det_normalish('f4b/nm', (2, 2, 3, 16)); tm = torch.tensor([0.3, 0.8])
    prompt = D.det_randint('f4b/prompt', (6,), 0, 256)
    init_noise = D.det_normalish('f4b/init', (6, 16))
    return cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise


SAMPLE_MAX_LEN = 70
SAMPLE_PROMPT = lambda: D.det_randint('f4b/u/prompt', (5,), 0, 256)
SAMPLE_NOISE = lambda: D.det_normalish('f4b/u/init', (64, 4))


def unet_modules(dim):
    enc, dec = nn.Conv2d(4, dim, 3, 2, 1), nn.ConvTranspose2d(dim, 4, 3, 2, 1, output_padding=1)
    fill_module_(enc, 'f4b/unet/enc', 0.15); fill_module_(dec, 'f4b/unet/dec', 0.05)
    return enc, dec


def unet_case():
    cfg = OracleConfig(**UNET_CFG)
    sd = base_sd(cfg, 'f4b/unet', drop_proj=True)
    img = lambda key, hw: D.det_normalish(key, (4, *hw))
    batch = [[D.det_randint('f4b/u/t0', (5,), 0, 256), (0, img('f4b/u/m0', (8, 8))), D.det_randint('f4b/u/t1', (3,), 0, 256)],
             [(0, img('f4b/u/m1', (4, 8))), D.det_randint('f4b/u/t2', (4,), 0, 256), (0, img('f4b/u/m2', (8, 8)))]]
    times = D.det_uniform('f4b/u/times', (2, 2), 0.05, 0.95)
    noises = [D.det_normalish(f'f4b/u/n{i}', s) for i, s in enumerate([(4, 8, 8), (4, 4, 8), (4, 8, 8)])]      # per instance, scan order (MP:716)
    xm = D.det_normalish('f4b/u/xm', (2, 4, 8, 8)); nm = D.det_normalish('f4b/u/nm', (2, 4, 8, 8)); tm = torch.tensor([0.3, 0.8])
    g0 = D.det_normalish('f4b/u/gen', (2, 8, 8, 4))                      # generate_modality_only noise before the channel-first rearrange (T:2890)
    return cfg, sd, batch, times, noises, xm, nm, tm, g0


class patched:
    """torch.randn_like / torch.randn replaced by a queue of prepared tensors (checked by shape)"""

    def __init__(self, name, queue):
        self.name, self.queue = name, list(queue)

    def __enter__(self):
        self.orig = getattr(torch, self.name)
        def fake(*a, **k):
            v = self.queue.pop(0)
            shape = tuple(a[0].shape) if torch.is_tensor(a[0]) else tuple(a[0] if isinstance(a[0], (tuple, list)) else a)
            assert tuple(v.shape) == shape, (tuple(v.shape), shape)
            return v.clone()
        setattr(torch, self.name, fake)
        return self

    def __exit__(self, *a):
        setattr(torch, self.name, self.orig)


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def summarize(grads):
    """norm of every gradient + the head of a few: enough to pin the backward without storing it all"""
    out = {k: g.norm() for k, g in grads.items()}
    heads = {k: g.reshape(-1)[:64].clone() for k, g in grads.items()
             if k.startswith(('pos_emb_mlp', 'latent_to_model_projs', 'model_to_latent_projs')) or k in ('text_embed.weight', 'transformer.layers.0.1.fn.to_qk.0.weight')}
    return out, heads


def make_pos():
    tp = import_reference()
    cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise = pos_case()
    model = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, modality_default_shape=(2, 3), add_pos_emb=True, modality_num_dim=2,
                           modality_processing='flat', prob_uncond=0.,
                           transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('pos_emb_mlp') for k in missing), (missing, unexpected)
    fill_module_(model.pos_emb_mlp, 'f4b/pos/mlp')
    pos_sd = {k: v.clone() for k, v in model.state_dict().items() if k.startswith('pos_emb_mlp')}
    model.train()
    with patched('randn_like', [noise]):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    gn, gh = summarize(grads_of(model))
    model.zero_grad(set_to_none=True)
    with patched('randn_like', [nm]):
        lm = model.forward_modality(xm, times=tm)
    lm.backward()
    gn_m, gh_m = summarize(grads_of(model))
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        pm = model.forward_modality(xm, times=tm, return_loss=False)
        logits = model(batch, times=times, return_loss=False)
    # greedy sample_one: text is deterministic at temperature 0; the modality noise is injected
    torch.manual_seed(0)
    out = model.sample_one(prompt, max_length=14, text_temperature=0., init_modality_noise=init_noise, modality_steps=GEN_STEPS, fixed_modality_shape=(2, 3),
                           cfg_scale=1., force_modality_at_start=0, cache_kv=True)
    parts = [(p if not isinstance(p, tuple) else ('mod', p[0], p[1])) for p in out]
    torch.save(dict(pos_sd=pos_sd, loss=loss.detach(), text_loss=bd.text.detach(), flow_losses=[f.detach() for f in bd.flow], grad_norms=gn, grad_heads=gh,
                    fm_loss=lm.detach(), fm_grad_norms=gn_m, fm_grad_heads=gh_m, fm_pred=pm, logits=logits, sample=parts),
               os.path.join(OUT, 'f4b_pos.pt'))
    print('pos: loss', float(loss), 'fm', float(lm), 'sample', [p.shape if torch.is_tensor(p) else p[2].shape for p in parts])


def make_pos_clean():
    """`add_pos_emb` together with `model_output_clean` (T:1297): the model-space conversion subtracts the PROJECTED noised tokens
    (`processed.packed`, MP:786-792) - the positional embedding joins the stream afterwards (T:3173-3176) and is not part of the subtrahend"""
    tp = import_reference()
    cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise = pos_case()
    model = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, modality_default_shape=(2, 3), add_pos_emb=True, modality_num_dim=2,
                           modality_processing='flat', prob_uncond=0., model_output_clean=True,
                           transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('pos_emb_mlp') for k in missing), (missing, unexpected)
    fill_module_(model.pos_emb_mlp, 'f4b/pos/mlp')
    pos_sd = {k: v.clone() for k, v in model.state_dict().items() if k.startswith('pos_emb_mlp')}
    model.train()
    with patched('randn_like', [noise]):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    gn, gh = summarize(grads_of(model))
    torch.save(dict(pos_sd=pos_sd, loss=loss.detach(), text_loss=bd.text.detach(), flow_losses=[f.detach() for f in bd.flow], grad_norms=gn, grad_heads=gh),
               os.path.join(OUT, 'f4b_pos_clean.pt'))
    print('pos + model_output_clean: loss', float(loss), 'flow', [float(f) for f in bd.flow])


def make_unet():
    tp = import_reference()
    cfg, sd, batch, times, noises, xm, nm, tm, g0 = unet_case()
    enc, dec = unet_modules(cfg.dim)
    model = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                           pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, modality_processing='flat', prob_uncond=0.,
                           transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(('pos_emb_mlp', 'latent_to_model_projs', 'model_to_latent_projs')) for k in missing), (missing, unexpected)
    fill_module_(model.pos_emb_mlp, 'f4b/unet/mlp')
    ext_sd = {k: v.clone() for k, v in model.state_dict().items() if k.startswith(('pos_emb_mlp', 'latent_to_model_projs', 'model_to_latent_projs'))}
    model.train()
    with patched('randn_like', noises):
        loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    gn, gh = summarize(grads_of(model))
    model.zero_grad(set_to_none=True)
    with patched('randn_like', [nm]):
        lm = model.forward_modality(xm, times=tm)
    lm.backward()
    gn_m, gh_m = summarize(grads_of(model))
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        pm = model.forward_modality(xm, times=tm, return_loss=False)
        logits = model(batch, times=times, return_loss=False)
    with patched('randn', [g0]):
        gen = model.generate_modality_only(batch_size=2, modality_steps=GEN_STEPS)
    # `sample()` as train_mnist_with_unet.py / train_latent_with_text.py call it: the UN-CACHED sample_one (the stride-2 encoder changes the token
    # count, which the cached paths cannot slice).  Greedy text, injected modality noise; cfg_scale = 1 because the reference's guidance branch
    # ALWAYS decodes against a null-text kv cache (T:1972-1988, whatever `cache_kv` says) and then fails on this model (`decode_length` = 64 latent
    # positions against a block of 16 tokens: "shape '[4, 4, 128]' is invalid for input of size 2944").  One thing is pinned that the
    # reference leaves to chance: its un-cached text steps draw RANDOM times for the modalities already in the history (T:1917-1924 passes no
    # `times`); every other call of its samplers conditions them at 1 (T:1996, T:2192), and so does this golden (and the native loop).
    import transfusion_pytorch.transfusion as T
    orig_fn = T.default_modality_length_to_time_fn
    T.default_modality_length_to_time_fn = lambda num_modalities: torch.ones(num_modalities.shape[0], max(int(num_modalities.amax()), 1))
    try:
        out = model.sample_one(SAMPLE_PROMPT(), max_length=SAMPLE_MAX_LEN, text_temperature=0., init_modality_noise=SAMPLE_NOISE(), modality_steps=GEN_STEPS,
                               cfg_scale=1., force_modality_at_start=0, cache_kv=False)
    finally:
        T.default_modality_length_to_time_fn = orig_fn
    parts = [(p if not isinstance(p, tuple) else ('mod', p[0], p[1])) for p in out]
    torch.save(dict(ext_sd=ext_sd, loss=loss.detach(), text_loss=bd.text.detach(), flow_losses=[f.detach() for f in bd.flow], grad_norms=gn, grad_heads=gh,
                    fm_loss=lm.detach(), fm_grad_norms=gn_m, fm_grad_heads=gh_m, fm_pred=pm, logits=logits, gen=gen, sample=parts),
               os.path.join(OUT, 'f4b_unet.pt'))
    print('unet: loss', float(loss), 'fm', float(lm), 'gen', tuple(gen.shape), 'logits', tuple(logits.shape),
          'sample', [tuple(p.shape) if torch.is_tensor(p) else tuple(p[2].shape) for p in parts])


def teacher_parts(cfg):
    """the EMA teacher of the combination goldens: same architecture, its own deterministic weights everywhere"""
    sd = base_sd(cfg, 'f4b/unet/teacher', drop_proj=True)
    enc, dec = nn.Conv2d(4, cfg.dim, 3, 2, 1), nn.ConvTranspose2d(cfg.dim, 4, 3, 2, 1, output_padding=1)
    fill_module_(enc, 'f4b/unet/t_enc', 0.15); fill_module_(dec, 'f4b/unet/t_dec', 0.05)
    return sd, enc, dec


def teacher_noises():
    return [D.det_normalish(f'f4b/u/tn{i}', s) for i, s in enumerate([(4, 8, 8), (4, 4, 8), (4, 8, 8)])]


COMBO_DELTA = 1e-3


def make_unet_combo(clean):
    """SURVEY 8(f) rank 3 x rank 4: `velocity_consistency_ema_model` (T:3084-3088, T:3378-3418) - and, `clean`, `model_output_clean` (MP:786-792,
    T:2770-2810) - on the model whose modality type runs through a learnable conv encoder / decoder pair (MP:715-745).  Student and teacher
    are two reference models with different deterministic weights; `torch.randn_like` hands out the student's per-instance noises, then the
    teacher's (its `return_only_pred_flows` call noises again, T:3016)."""
    tp = import_reference()
    cfg, sd, batch, times, noises, xm, nm, tm, g0 = unet_case()

    def build(sd_, enc, dec, mlp_tag):
        m = tp.Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                           pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, modality_processing='flat', prob_uncond=0.,
                           model_output_clean=clean, transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
        missing, unexpected = m.load_state_dict(sd_, strict=False)
        assert not unexpected and all(k.startswith(('pos_emb_mlp', 'latent_to_model_projs', 'model_to_latent_projs')) for k in missing), (missing, unexpected)
        fill_module_(m.pos_emb_mlp, mlp_tag)
        return m, {k: v.clone() for k, v in m.state_dict().items() if k.startswith(('pos_emb_mlp', 'latent_to_model_projs', 'model_to_latent_projs'))}

    student, ext_sd = build(sd, *unet_modules(cfg.dim), 'f4b/unet/mlp')
    teacher, ext_sd_t = build(*teacher_parts(cfg), 'f4b/unet/t_mlp')
    student.train(); teacher.eval()
    with patched('randn_like', noises + teacher_noises()) as q:
        loss, bd = student(batch, times=times, velocity_consistency_ema_model=teacher, velocity_consistency_delta_time=COMBO_DELTA, return_breakdown=True)
        assert not q.queue
    loss.backward()
    gn, gh = summarize(grads_of(student))
    out = dict(ext_sd=ext_sd, ext_sd_teacher=ext_sd_t, loss=loss.detach(), text_loss=bd.text.detach(), flow_losses=[f.detach() for f in bd.flow],
               velocity_losses=[v.detach() for v in bd.velocity], grad_norms=gn, grad_heads=gh)
    if clean:
        student.zero_grad(set_to_none=True)
        with patched('randn_like', [nm]):
            lm = student.forward_modality(xm, times=tm)
        lm.backward()
        gn_m, gh_m = summarize(grads_of(student))
        with torch.no_grad():
            pm = student.forward_modality(xm, times=tm, return_loss=False)
        with patched('randn', [g0]):
            gen = student.generate_modality_only(batch_size=2, modality_steps=GEN_STEPS)
        import transfusion_pytorch.transfusion as T
        orig_fn = T.default_modality_length_to_time_fn
        T.default_modality_length_to_time_fn = lambda num_modalities: torch.ones(num_modalities.shape[0], max(int(num_modalities.amax()), 1))
        try:                                                                # the un-cached sample_one, as in make_unet (same pinning of the text steps' times)
            smp = student.sample_one(SAMPLE_PROMPT(), max_length=SAMPLE_MAX_LEN, text_temperature=0., init_modality_noise=SAMPLE_NOISE(), modality_steps=GEN_STEPS,
                                     cfg_scale=1., force_modality_at_start=0, cache_kv=False)
        finally:
            T.default_modality_length_to_time_fn = orig_fn
        out.update(fm_loss=lm.detach(), fm_grad_norms=gn_m, fm_grad_heads=gh_m, fm_pred=pm, gen=gen,
                   sample=[(p if not isinstance(p, tuple) else ('mod', p[0], p[1])) for p in smp])
    name = 'f4b_unet_clean_velocity.pt' if clean else 'f4b_unet_velocity.pt'
    torch.save(out, os.path.join(OUT, name))
    print(name, 'loss', float(loss), 'flow', [float(f) for f in bd.flow], 'velocity', [float(v) for v in bd.velocity],
          *(('fm', float(out['fm_loss'])) if clean else ()))


if __name__ == '__main__':
    import sys
    which = sys.argv[1:] or ['pos', 'unet', 'pos_clean', 'unet_velocity', 'unet_clean_velocity']
    if 'unet_velocity' in which:
        make_unet_combo(False)
    if 'unet_clean_velocity' in which:
        make_unet_combo(True)
    if 'pos' in which:
        make_pos()
    if 'unet' in which:
        make_unet()
    if 'pos_clean' in which:
        make_pos_clean()
