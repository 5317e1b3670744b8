"""SURVEY 8(f) rank 4, second half: `add_pos_emb` (axial positional embedding, T:1384-1403 / T:2781-2796 / T:3173-3180) and learnable
`pre_post_transformer_enc_dec` pairs (T:1451-1494, MP:715-745) around the native engine - goldens from the UNMODIFIED reference
(oracle/make_golden_f4b.py -> tests/golden/f4b_pos.pt, f4b_unet.pt).

What is native here: everything between the token rows and the final embedding rows, forward and backward, including the ADD of the
positional rows into the packed stream and the gradients back out to the rows (identity-GEMM mapped RESID epilogue).  What is PyTorch: the
positional MLP (a few hundred rows) and the user's conv encoder / decoder.  Tolerances as in tests/test_model_gpu.py (bf16 vs fp32)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.make_golden_f4b import (COMBO_DELTA, GEN_STEPS, SAMPLE_MAX_LEN, SAMPLE_NOISE, SAMPLE_PROMPT, pos_case,      # noqa: E402
                                    teacher_noises, teacher_parts, unet_case, unet_modules)

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
GRAD_TOL, GRAD_MEAN_TOL, GRAD_HEAD_TOL = 4e-2, 1.2e-2, 6e-2


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def close(a, r, what, tol=2e-3):
    a, r = float(a.detach()) if torch.is_tensor(a) else float(a), float(r)
    print(f'  {what}: native {a:.6f} reference {r:.6f} delta {a - r:+.2e}')
    assert abs(a - r) <= tol * max(1., abs(r)), what


def check_grads(model, norms, heads, what):
    wsum = nsum = 0.
    worst = ('', 0.)
    for k, p in model.named_parameters():
        if k not in norms or float(norms[k]) < 1e-6:
            continue
        assert p.grad is not None, f'{what}: no gradient for {k}'
        gn = float(p.grad.double().norm())
        assert abs(gn - float(norms[k])) <= GRAD_TOL * float(norms[k]), (what, k, gn, float(norms[k]))
        if k in heads:
            r = rel(p.grad.float().reshape(-1)[:64], heads[k])
            if r > worst[1]:
                worst = (k, r)
            assert r <= GRAD_HEAD_TOL, (what, k, r)
            wsum += r * float(norms[k]); nsum += float(norms[k])
    print(f'  {what}: gradient heads worst {worst[0]} {worst[1]:.2e}, norm-weighted mean {wsum / max(nsum, 1e-30):.2e}')
    assert wsum / nsum <= GRAD_MEAN_TOL
    missing = [k for k in norms if float(norms[k]) >= 1e-6 and k not in dict(model.named_parameters())]
    assert not missing, missing


def to_cuda(batch):
    return [[(p[0], p[1].cuda()) if isinstance(p, tuple) else p.cuda() for p in s] for s in batch]


def test_axial_positional_embedding_matches_reference_golden():
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.optim import FusedAdam
    g = torch.load(os.path.join(GOLDEN, 'f4b_pos.pt'), weights_only=False)
    cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise = pos_case()
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, modality_default_shape=(2, 3), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict({**sd, **g['pos_sd']}, strict=True)
    model = model.cuda().train()
    model._noise_override = {0: noise.cuda()}
    loss, bd = model(to_cuda(batch), times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print('[pos] interleaved step')
    close(loss, g['loss'], 'loss'); close(bd.text, g['text_loss'], 'text'); close(bd.flow[0], g['flow_losses'][0], 'flow')
    check_grads(model, g['grad_norms'], g['grad_heads'], 'interleaved')
    assert any(k.startswith('pos_emb_mlp') and float(v) > 1e-6 for k, v in g['grad_norms'].items())
    with torch.no_grad():
        logits = model(to_cuda(batch), times=times, return_loss=False)
    assert logits.shape == g['logits'].shape and rel(logits, g['logits']) <= 1e-2
    for p in model.parameters():
        p.grad = None
    model._noise_override = {0: nm.reshape(-1, 16).cuda()}
    lm = model.forward_modality(xm.cuda(), times=tm)
    lm.backward()
    print('[pos] forward_modality')
    close(lm, g['fm_loss'], 'loss')
    check_grads(model, g['fm_grad_norms'], g['fm_grad_heads'], 'forward_modality')
    with torch.no_grad():
        pm = model.forward_modality(xm.cuda(), times=tm, return_loss=False)
    assert pm.shape == g['fm_pred'].shape and rel(pm, g['fm_pred']) <= 1.5e-2
    # the fused optimizer steps the MLP too (stock Adam under the same global clip coefficient)
    before = {k: v.detach().clone() for k, v in model.pos_emb_mlp.state_dict().items()}
    flat_before = model.store.flat.clone()
    opt = FusedAdam(model, lr=1e-3, max_grad_norm=0.5)
    assert len(opt.ext_params) == len(list(model.pos_emb_mlp.parameters())) > 0
    opt.step(); opt.zero_grad()
    assert any(not torch.equal(before[k], v) for k, v in model.pos_emb_mlp.state_dict().items()) and not torch.equal(flat_before, model.store.flat)
    assert all(p.grad is None for p in model.parameters())


def test_positional_embedding_with_model_output_clean_matches_reference_golden():
    """`add_pos_emb` + `model_output_clean`: the model-space conversion subtracts the PROJECTED noised tokens (MP:786-792); the positional embedding
    joins the stream afterwards (T:3173-3176) and must not be part of the subtrahend (golden f4b_pos_clean.pt from the reference)"""
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'f4b_pos_clean.pt'), weights_only=False)
    cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise = pos_case()
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, modality_default_shape=(2, 3), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        model_output_clean=True, transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict({**sd, **g['pos_sd']}, strict=True)
    model = model.cuda().train()
    model._noise_override = {0: noise.cuda()}
    loss, bd = model(to_cuda(batch), times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print('[pos + model_output_clean] interleaved step')
    # (the conversion divides by 1 - t >= 0.05: the flow loss carries up to 20x the bf16 noise of the embedding)
    close(loss, g['loss'], 'loss', tol=5e-3); close(bd.text, g['text_loss'], 'text'); close(bd.flow[0], g['flow_losses'][0], 'flow', tol=5e-3)
    check_grads(model, g['grad_norms'], g['grad_heads'], 'interleaved (clean)')


def test_sample_one_adds_positional_embedding_like_the_reference():
    """the reference's `sample_one` (= `sample`) goes through forward() and adds the embedding to the block being decoded (T:3179-3180); its
    `sample_many` does not (T:2436-2444).  Greedy text + injected modality noise: the decoded modality must match the reference's `sample_one`."""
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'f4b_pos.pt'), weights_only=False)
    cfg, sd, batch, times, noise, xm, nm, tm, prompt, init_noise = pos_case()
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, modality_default_shape=(2, 3), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict({**sd, **g['pos_sd']}, strict=True)
    model = model.cuda().eval()
    kw = dict(max_length=14, text_temperature=0., init_modality_noise=init_noise, modality_steps=GEN_STEPS, fixed_modality_shape=(2, 3), cfg_scale=1.,
              force_modality_at_start=0)
    ref = g['sample']
    ref_mod = next(p for p in ref if isinstance(p, tuple))
    out_one = model.sample_one(prompt.cuda(), **kw)
    out_many = model.sample_many([prompt.cuda()], **kw)[0]
    through = model._sample_one_through_forward(prompt.cuda(), cache_kv=True, **kw)
    mod = lambda parts: next(p for p in parts if isinstance(p, tuple))
    e_one, e_many, e_thr = rel(mod(out_one)[1], ref_mod[2]), rel(mod(out_many)[1], ref_mod[2]), rel(mod(through)[1], ref_mod[2])
    print(f'decoded modality vs reference sample_one: sample_one {e_one:.2e}, forward()-loop {e_thr:.2e}, sample_many (no embedding in the ODE) {e_many:.2e}')
    assert mod(out_one)[1].shape == ref_mod[2].shape == (2, 3, 16)
    assert e_one <= 3e-2 and e_thr <= 3e-2
    assert e_many > 2 * e_one            # the two reference samplers really differ; each entry point follows its own
    assert out_one[0].tolist() == ref[0].tolist()          # prompt + [meta] shape [som]: forced, identical


def test_unet_encoder_decoder_around_the_transformer_matches_reference_golden():
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.optim import FusedAdam
    g = torch.load(os.path.join(GOLDEN, 'f4b_unet.pt'), weights_only=False)
    cfg, sd, batch, times, noises, xm, nm, tm, g0 = unet_case()
    enc, dec = unet_modules(cfg.dim)
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                        pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    # the reference's key names for the wrapped conv pair load unchanged: Sequential(enc, Rearrange) -> `.0.0.*`, Sequential(Rearrange, dec) -> `.0.1.*`
    assert {'latent_to_model_projs.0.0.weight', 'model_to_latent_projs.0.1.weight'} <= set(g['ext_sd'])
    model.load_state_dict({**sd, **g['ext_sd']}, strict=True)
    model = model.cuda().train()
    model._noise_override = {0: [n.cuda() for n in noises]}
    loss, bd = model(to_cuda(batch), times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print('[unet] interleaved step')
    close(loss, g['loss'], 'loss'); close(bd.text, g['text_loss'], 'text'); close(bd.flow[0], g['flow_losses'][0], 'flow')
    check_grads(model, g['grad_norms'], g['grad_heads'], 'interleaved')
    with torch.no_grad():
        logits = model(to_cuda(batch), times=times, return_loss=False)
    assert logits.shape == g['logits'].shape and rel(logits, g['logits']) <= 1e-2      # packed with the PROJECTED (down-sampled) lengths, MP:738-741
    for p in model.parameters():
        p.grad = None
    model._noise_override = {0: nm.cuda()}
    lm = model.forward_modality(xm.cuda(), times=tm)
    lm.backward()
    print('[unet] forward_modality')
    close(lm, g['fm_loss'], 'loss')
    check_grads(model, g['fm_grad_norms'], g['fm_grad_heads'], 'forward_modality')
    with torch.no_grad():
        pm = model.forward_modality(xm.cuda(), times=tm, return_loss=False)
    assert pm.shape == g['fm_pred'].shape and rel(pm, g['fm_pred']) <= 1.5e-2
    model._gen_noise_override = g0
    gen = model.generate_modality_only(batch_size=2, modality_steps=GEN_STEPS)
    assert gen.shape == g['gen'].shape == (2, 4, 8, 8) and rel(gen, g['gen']) <= 2e-2
    # one optimizer step moves the conv pair, the MLP and the flat buffer; the EMA copy owns its own modules
    opt = FusedAdam(model, lr=1e-3, max_grad_norm=0.5)
    w0 = model.latent_to_model_projs[0][0].weight.detach().clone()
    opt.step(); opt.zero_grad()
    assert not torch.equal(w0, model.latent_to_model_projs[0][0].weight)
    ema = model.create_ema()
    assert ema.ema_model.latent_to_model_projs[0][0].weight.data_ptr() != model.latent_to_model_projs[0][0].weight.data_ptr()
    assert torch.equal(ema.ema_model.latent_to_model_projs[0][0].weight, model.latent_to_model_projs[0][0].weight)


def test_sample_with_unet_encoder_decoder_matches_the_references_uncached_sample_one():
    """`model.sample()` with a stride-2 conv pair around the transformer, as train_mnist_with_unet.py / train_latent_with_text.py call it: the
    un-cached `sample_one` loop through forward() (the only decode path of the reference that handles an encoder which changes the token count,
    and only at cfg_scale = 1 - its guidance branch fails on this model, see oracle/make_golden_f4b.py).  Greedy text, injected noise: the text
    before the modality is forced and identical, the decoded image matches, and the greedy continuation follows the reference's tokens."""
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'f4b_unet.pt'), weights_only=False)
    cfg, sd, *_ = unet_case()
    enc, dec = unet_modules(cfg.dim)
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                        pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict({**sd, **g['ext_sd']}, strict=True)
    model = model.cuda().eval()
    kw = dict(max_length=SAMPLE_MAX_LEN, text_temperature=0., init_modality_noise=SAMPLE_NOISE().cuda(), modality_steps=GEN_STEPS, force_modality_at_start=0)
    ref = g['sample']
    out = model.sample(SAMPLE_PROMPT().cuda(), cfg_scale=1., **kw)
    assert [isinstance(p, tuple) for p in out] == [isinstance(p, tuple) for p in ref] == [False, True, False]
    assert out[0].tolist() == ref[0].tolist()
    assert out[1][0] == ref[1][1] == 0 and out[1][1].shape == ref[1][2].shape == (4, 8, 8)
    e = rel(out[1][1], ref[1][2])
    a, b = out[2].tolist(), ref[2].tolist()
    same = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
    print(f'[unet] sample(): decoded image rel {e:.2e}; greedy continuation {a} vs reference {b} (identical for {same} tokens)')
    assert e <= 3e-2
    assert a[0] == b[0] == model.eom_ids[0] and same >= 2          # [eom] is forced; bf16 may flip a later near-tie, never the first free token
    # batched entry point and classifier-free guidance (where the reference itself stops): same shapes, finite values, guidance changes the image
    many = model.sample_many([SAMPLE_PROMPT().cuda(), None], cfg_scale=3., **{**kw, 'max_length': 66})
    assert len(many) == 2
    for parts in many:
        mods = [p for p in parts if isinstance(p, tuple)]
        assert mods and all(p[1].shape == (4, 8, 8) and torch.isfinite(p[1]).all() for p in mods)
    assert rel(many[0][1][1], out[1][1]) > 1e-3
    # the kv-cached decode contract of forward() needs a length-preserving encoder: with this one the block has 16 tokens for 64 latent positions
    # (the null-text cache of the guidance branch is the first cached modality step of the loop - where the reference fails, too)
    with pytest.raises(AssertionError):
        model._sample_one_through_forward(SAMPLE_PROMPT().cuda(), cache_kv=True, cfg_scale=3., **kw)


@pytest.mark.parametrize('clean', [False, True], ids=['velocity', 'clean_velocity'])
def test_unet_types_with_velocity_consistency_and_model_output_clean_match_reference_golden(clean):
    """SURVEY 8(f) rank 3 x rank 4: the EMA teacher's velocity-consistency term (T:3084-3088, T:3378-3418) and `model_output_clean` (MP:786-792,
    T:2770-2810) on a modality type whose maps are a learnable conv pair (MP:715-745).  The student's and the teacher's decoders run in PyTorch on
    the engine's embedding rows; with `clean` the rows go through (embed - encoder output) / max(1 - t, eps) first.  Goldens from the unmodified
    reference (oracle/make_golden_f4b.py: f4b_unet_velocity.pt, f4b_unet_clean_velocity.pt)."""
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'f4b_unet_clean_velocity.pt' if clean else 'f4b_unet_velocity.pt'), weights_only=False)
    cfg, sd, batch, times, noises, xm, nm, tm, g0 = unet_case()

    def build(sd_, enc, dec, ext_sd):
        m = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                        pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, prob_uncond=0., model_output_clean=clean,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
        m.load_state_dict({**sd_, **ext_sd}, strict=True)
        return m.cuda()

    student = build(sd, *unet_modules(cfg.dim), g['ext_sd']).train()
    teacher = build(*teacher_parts(cfg), g['ext_sd_teacher']).eval()
    student._noise_override = {0: [n.cuda() for n in noises]}
    teacher._noise_override = {0: [n.cuda() for n in teacher_noises()]}
    loss, bd = student(to_cuda(batch), times=times, velocity_consistency_ema_model=teacher, velocity_consistency_delta_time=COMBO_DELTA, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f'[unet x velocity{" x clean" if clean else ""}] interleaved step')
    # (clean: the conversion divides by 1 - t >= 0.05 - the flow terms carry up to 20x the bf16 noise of the embedding rows, as in the pos_clean golden)
    tol = 5e-3 if clean else 2e-3
    close(loss, g['loss'], 'loss', tol=tol); close(bd.text, g['text_loss'], 'text'); close(bd.flow[0], g['flow_losses'][0], 'flow', tol=tol)
    assert len(bd.velocity) == len(g['velocity_losses']) == 1
    close(bd.velocity[0], g['velocity_losses'][0], 'velocity', tol=tol)
    check_grads(student, g['grad_norms'], g['grad_heads'], 'interleaved')
    assert all(p.grad is None for p in teacher.parameters())
    if not clean:
        return
    for p in student.parameters():
        p.grad = None
    student._noise_override = {0: nm.cuda()}
    lm = student.forward_modality(xm.cuda(), times=tm)
    lm.backward()
    print('[unet x clean] forward_modality')
    close(lm, g['fm_loss'], 'loss', tol=tol)
    check_grads(student, g['fm_grad_norms'], g['fm_grad_heads'], 'forward_modality')
    with torch.no_grad():
        pm = student.forward_modality(xm.cuda(), times=tm, return_loss=False)
    assert pm.shape == g['fm_pred'].shape and rel(pm, g['fm_pred']) <= 1.5e-2
    student._gen_noise_override = g0
    gen = student.generate_modality_only(batch_size=2, modality_steps=GEN_STEPS)
    assert gen.shape == g['gen'].shape == (2, 4, 8, 8) and rel(gen, g['gen']) <= 2e-2
    # the un-cached sample_one through the decode contract of forward(): its closures carry the conversion (MP:786-792)
    student.eval()
    ref = g['sample']
    out = student.sample(SAMPLE_PROMPT().cuda(), cfg_scale=1., max_length=SAMPLE_MAX_LEN, text_temperature=0., init_modality_noise=SAMPLE_NOISE().cuda(),
                         modality_steps=GEN_STEPS, force_modality_at_start=0)
    assert [isinstance(p, tuple) for p in out] == [isinstance(p, tuple) for p in ref] == [False, True, False]
    assert out[0].tolist() == ref[0].tolist() and out[1][1].shape == ref[1][2].shape == (4, 8, 8)
    e = rel(out[1][1], ref[1][2])
    print(f'[unet x clean] sample(): decoded image rel {e:.2e}; continuation {out[2].tolist()} vs reference {ref[2].tolist()}')
    assert e <= 3e-2 and out[2].tolist()[0] == ref[2].tolist()[0] == student.eom_ids[0]


def test_reconstruction_loss_matches_reference_golden():
    """`reconstruction_loss_weight > 0` (T:1522-1525): interleaved forward (per type the mean over instances of mse(noised, noise + pred (1 - t)),
    MP:177-200, T:3420-3431 - the native residual kernel adds its gradient to the same flow-prediction seeds), forward_modality without decoder
    (target = the clean latent, gradient) and through a frozen decoder (no gradient, T:2845-2848).  Golden tests/golden/recon1.pt."""
    from oracle.cases import build_case, default_shapes
    from oracle.make_golden_f4 import enc_dec, f4_case
    from oracle.make_golden_recon import WEIGHT, recon_inputs
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'recon1.pt'), weights_only=False)
    cfg, sd, batch, times, noise = build_case('small2')
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents, modality_default_shape=default_shapes(cfg), reconstruction_loss_weight=WEIGHT,
                        prob_uncond=0., transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    loss, bd = model(to_cuda(batch), times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print('[recon] interleaved step')
    close(loss, g['loss'], 'loss'); close(bd.text, g['text_loss'], 'text')
    for i, (a, r) in enumerate(zip(bd.flow, g['flow_losses'])):
        close(a, r, f'flow{i}')
    assert len(bd.recon) == len(g['recon']) == 2
    for i, (a, r) in enumerate(zip(bd.recon, g['recon'])):
        close(a, r, f'recon{i}')
    check_grads(model, g['grad_norms'], g['grad_heads'], 'interleaved')
    for p in model.parameters():
        p.grad = None
    xm, nm, tm = recon_inputs()
    model._noise_override = {0: nm.reshape(-1, cfg.dim_latents[0]).cuda()}
    lm, (fl, vl, rl) = model.forward_modality(xm.cuda(), times=tm, modality_type=0, return_loss_breakdown=True)
    lm.backward()
    print('[recon] forward_modality')
    close(lm, g['fm_loss'], 'loss'); close(fl, g['fm_flow'], 'flow'); close(rl, g['fm_recon'], 'recon')
    check_grads(model, g['fm_grad_norms'], g['fm_grad_heads'], 'forward_modality')
    # through the frozen decoder of the f4 golden: a value without gradient
    cfg4, sd4, _, _, _, xm4, nm4, tm4, _ = f4_case()
    enc, dec = enc_dec()
    m4 = Transfusion(num_text_tokens=cfg4.num_text_tokens, dim_latent=16, channel_first_latent=True, modality_default_shape=(4,), modality_encoder=enc, modality_decoder=dec,
                     reconstruction_loss_weight=WEIGHT, prob_uncond=0., transformer=dict(dim=cfg4.dim, depth=cfg4.depth, dim_head=cfg4.dim_head, heads=cfg4.heads))
    m4.load_state_dict(sd4, strict=False)
    m4 = m4.cuda().train()
    m4._noise_override = {0: nm4.movedim(1, -1).reshape(-1, 16).cuda()}
    l4, (f4, v4, r4) = m4.forward_modality(xm4.cuda(), times=tm4, return_loss_breakdown=True)
    print('[recon] forward_modality through the frozen decoder')
    close(l4, g['dec_loss'], 'loss'); close(f4, g['dec_flow'], 'flow'); close(r4, g['dec_recon'], 'recon')


def test_processing_registry_with_positional_embedding_and_unet_types():
    """`PROCESSING_STRATEGIES[name](..., need_axial_pos_emb=True)` (MP:1050-1058 with MP:1003-1045 evaluated): the embedding rows of every instance
    sit at its slots, zeros elsewhere; for a `pre_post_transformer_enc_dec` type the tokens are the user's encoder output at the PROJECTED positions
    (the reference's `test_unet_encoder_positions_use_projected_lengths`, tests/test_modality_processing.py:482-493)."""
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.modality_processing import PROCESSING_STRATEGIES
    cfg, sd, batch, times, noises, xm, nm, tm, g0 = unet_case()
    g = torch.load(os.path.join(GOLDEN, 'f4b_unet.pt'), weights_only=False)
    enc, dec = unet_modules(cfg.dim)
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=4, modality_default_shape=(8, 8), channel_first_latent=True,
                        pre_post_transformer_enc_dec=(enc, dec), add_pos_emb=True, modality_num_dim=2, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict({**sd, **g['ext_sd']}, strict=True)
    model = model.cuda()
    model._noise_override = {0: [n.cuda() for n in noises]}
    with torch.no_grad():
        out = PROCESSING_STRATEGIES['auto'](to_cuda(batch), times, model, need_axial_pos_emb=True, return_loss=True, return_embed=False)
    # 8 x 8 and 4 x 8 images through a stride-2 conv: 16- and 8-token instances, spelled "4,4" / "2,4" in the meta string
    assert [[(t, L) for t, _, L in s] for s in out.modality_positions] == [[(0, 16)], [(0, 8), (0, 16)]]
    assert out.modality_pos_emb.shape == out.modality_tokens.shape
    (t0, off, L), = out.modality_positions[0]
    want = model.pos_emb_mlp[0]((4, 4), flatten=True)
    assert torch.allclose(out.modality_pos_emb[0, off:off + L], want, atol=1e-6) and float(out.modality_pos_emb[0, :off].abs().max()) == 0.
    tok = model.latent_to_model_projs[0]((batch[0][1][1].cuda() * times[0, 0] + noises[0].cuda() * (1 - times[0, 0]))[None])[0].reshape(16, cfg.dim)
    assert rel(out.modality_tokens[0, off:off + L], tok) <= 8e-3                       # bf16 token buffer
    assert [tuple(f.shape) for f in out.flows[0]] == [(4, 8, 8), (4, 4, 8), (4, 8, 8)]
