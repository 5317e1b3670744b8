"""End-to-end parity of the native (HIP, bf16 compute / fp32 master) training path against the golden vectors
the UNMODIFIED reference produced (tests/golden/*.pt, fp32 CPU) and - for the small cases - the live oracle.

Tolerances (bf16 vs fp32; the reference's own bf16-autocast floor is rel-Frobenius 4.9e-3 on logits):
  loss / text loss / flow losses : |delta| <= 1e-3 * max(1, |ref|) = LOSS_TOL (measured 7e-6 ... 5e-4: the north star's 1e-3 is the gate)
  logits, final embed            : rel-Frobenius <= 1e-2 = LOGIT_TOL           (measured 5.9e-3 at dim512/depth8: bf16 activations cannot reach 1e-3 here -
                                   the reference's own bf16-autocast run is at 4.9e-3)
  greedy token (argmax)          : identical wherever the reference's top-2 margin exceeds 0.05; >= 95% overall (measured 96.9 ... 100%)
                                   (near-ties of random-init logits flip under ANY bf16 rounding: the reference's own
                                   bf16-autocast run agrees with its fp32 run on 99.1%)
  gradients                      : per-parameter rel-Frobenius <= 4e-2 = GRAD_TOL (measured worst 3.3e-2), norm-weighted mean <= 1.2e-2 = GRAD_MEAN_TOL
                                   (measured 7.1e-3); 1024-element HEAD slices of a gradient (big cases) <= 8e-2 = GRAD_HEAD_TOL: the fp32 atomics of the
                                   weight-gradient GEMMs accumulate in a run-dependent order, and a slice of a small-norm gradient (zero-initialised
                                   ada-ln-zero weights deep in the 24-layer case) moves between 4.3e-2 and 6.5e-2 from run to run / build to build
                                   while the median over all 596 slices stays at 1.0e-2
Tolerances are ~1.5x the worst value measured on MI355X, so that a regression which doubles an error fails.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.cases import build_case, with_grad          # noqa: E402
from oracle.transfusion_oracle import forward_train     # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
LOGIT_TOL, GRAD_TOL, GRAD_MEAN_TOL, GRAD_HEAD_TOL = 1e-2, 4e-2, 1.2e-2, 8e-2
LOSS_TOL = 1e-3          # the north star's tolerance, met on every loss (round 3 held them to 2e-3; worst measured 9.8e-4: the text loss of the CFG-dropped batch)


def build_native(cfg, sd):
    from transfusion_pytorch_amd import Transfusion
    dl = cfg.dim_latents if len(cfg.dim_latents) > 1 else cfg.dim_latents[0]
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=dl,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads), prob_uncond=0.)
    model.load_state_dict(sd, strict=True)
    return model.cuda()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def run_native(name):
    cfg, sd, batch, times, noise = build_case(name)
    model = build_native(cfg, sd)
    model.train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    plan = model._live[0]
    b, n, nt = plan.b, plan.n, model._live_n_true               # training lengths are bucketed to multiples of 64: compare the real columns
    logits = plan.logits.view(b, n, -1)[:, :nt, :cfg.vocab].float().cpu()
    embed = plan.embed.view(b, n, -1)[:, :nt].float().cpu()
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    return cfg, model, dict(loss=float(loss), text=float(bd.text), flow=[float(f) for f in bd.flow], logits=logits, embed=embed, grads=grads)


PARITY_LOG = {}          # measured parity figures of this run -> gpurun_out/parity_measured.json (copied to profiles/ and quoted by bench.py `parity_met`)


def _log_parity(name, **kw):
    import json
    PARITY_LOG.setdefault(name, {}).update({k: float(v) for k, v in kw.items()})
    out = os.path.join(os.path.dirname(GOLDEN), '..', 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_measured.json'), 'w') as f:
            json.dump(PARITY_LOG, f, indent=1, sort_keys=True)
    except OSError:
        pass


def compare_losses(out, g, tol=LOSS_TOL):
    for nm, a, r in [('loss', out['loss'], float(g['loss'])), ('text', out['text'], float(g['text_loss']))] + \
                    [(f'flow{i}', a, float(r)) for i, (a, r) in enumerate(zip(out['flow'], g['flow_losses']))]:
        print(f'  {nm}: native {a:.6f} reference {r:.6f} delta {a - r:+.2e}')
        assert abs(a - r) <= tol * max(1., abs(r)), nm


@pytest.mark.parametrize('name', ['tiny1', 'small2', 'mid2', 'head8', 'canon512', 'cfg4_768', 'cfg3_1024'])
def test_training_step_matches_reference_golden(name):
    """`cfg4_768` / `cfg3_1024`: the model dimensions of BASELINE configs 4 and 3 (dim768/depth16 with two modality types, dim1024/depth24:
    AttentionResidual over 25 hiddens, 12 U-Net skip pairs), one canonical 1024-token sample each, golden from the unmodified reference."""
    g = torch.load(os.path.join(GOLDEN, f'{name}.pt'), weights_only=False)
    cfg, model, out = run_native(name)
    print(f'[{name}]')
    compare_losses(out, g)
    rs = g['row_step']
    lg, lr = out['logits'][:, ::rs], g['logits']
    e_log, e_emb = rel(lg, lr), rel(out['embed'][:, ::rs], g['embed'])
    am_n, am_r = lg.argmax(-1), lr.argmax(-1)
    top2 = lr.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    agree = (am_n == am_r).float().mean().item()
    safe = margin > 0.05
    agree_safe = (am_n == am_r)[safe].float().mean().item()
    print(f'  logits rel-fro {e_log:.3e}  embed rel-fro {e_emb:.3e}  argmax agreement {agree:.4f} (margin>0.05: {agree_safe:.4f}, {safe.float().mean():.3f} of positions)')
    _log_parity(name, loss_rel=abs(out['loss'] - float(g['loss'])) / max(1., abs(float(g['loss']))), logits_rel=e_log, embed_rel=e_emb, argmax_unfiltered=agree, argmax_margin_gt_0p05=agree_safe)
    assert e_log <= LOGIT_TOL and e_emb <= LOGIT_TOL
    # every disagreement must sit at a near-tie of the REFERENCE's own logits (top-2 margin <= 0.05); how many near-ties flip is chaotic at depth 24 -
    # 3 or 4 of the 128 sampled positions of cfg3_1024 depending on the build (a 1e-6 change in the soft-cap polynomial moved one) - so the
    # unfiltered share only has a floor
    assert agree_safe == 1.0 and agree >= 0.95
    worst, num, den = (None, 0.), 0., 0.
    for k, gn in g['grad_norms'].items():
        assert k in out['grads'], f'missing gradient for {k}'
        go = out['grads'][k]
        tol = GRAD_TOL
        if 'grads' in g:
            e = rel(go, g['grads'][k])
        else:
            # big cases store norms + the first 256 elements of every gradient: the norm obeys GRAD_TOL, the (noisier) slice GRAD_HEAD_TOL
            head = g['grad_head'][k]
            e_norm = abs(go.double().norm().item() - gn) / (gn + 1e-30)
            e = ((go.reshape(-1)[:head.numel()] - head).norm() / (head.norm() + 1e-30)).item() if head.norm() > 1e-3 * gn else 0.
            assert e_norm <= GRAD_TOL, f'gradient norm {k}: rel err {e_norm:.3e}'
            e, tol = max(e, e_norm), GRAD_HEAD_TOL
        num += e * gn; den += gn
        if e > worst[1]:
            worst = (k, e)
        assert e <= tol, f'gradient {k}: rel err {e:.3e}'
    print(f'  gradients: norm-weighted mean rel err {num / den:.3e}; worst {worst[0]} {worst[1]:.3e}')
    _log_parity(name, grad_mean_rel=num / den, grad_worst_rel=worst[1])
    assert num / den <= GRAD_MEAN_TOL


def test_canon512_at_bench_batch_matches_golden_rows():
    """the bench's own geometry - b = 64 samples x 1024 tokens at dim512 / depth 8, i.e. the GEMM tilings, TN split counts
    and segment grids the timed step runs on - against the REFERENCE golden of `canon512` (b = 2).  The batch is the golden's two samples 32 times
    over (same times, same noise): samples only see themselves, the losses are token means, so loss, every sample's logits and every gradient
    must equal the golden's."""
    g = torch.load(os.path.join(GOLDEN, 'canon512.pt'), weights_only=False)
    cfg, sd, batch, times, noise = build_case('canon512')
    rep = 32
    model = build_native(cfg, sd).train()
    model._noise_override = {t: v.repeat(rep, 1).cuda() for t, v in noise.items()}
    loss, bd = model(batch * rep, times=times.repeat(rep, 1), return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    plan = model._live[0]
    assert (plan.b, plan.n) == (64, 1024)
    out = dict(loss=float(loss), text=float(bd.text), flow=[float(f) for f in bd.flow])
    compare_losses(out, g)
    nt, rs = model._live_n_true, g['row_step']
    logits = plan.logits.view(64, 1024, -1)[:, :nt:rs, :cfg.vocab].float().cpu()
    worst = max(rel(logits[i], g['logits'][i % 2]) for i in range(64))
    agree = (logits.argmax(-1) == g['logits'].argmax(-1).repeat(rep, 1)).float().mean().item()
    spread = max(rel(logits[i], logits[i % 2]) for i in range(2, 64))          # copies of a sample: identical up to nothing (same kernels, same tiles modulo position)
    print(f'  b=64: worst per-sample logits rel-fro {worst:.3e}, argmax agreement {agree:.4f}, copy-to-copy spread {spread:.2e}')
    assert worst <= LOGIT_TOL and agree >= 0.975 and spread <= 1e-3
    num = den = 0.
    for k, gn in g['grad_norms'].items():
        go = model.store.params[k].grad.detach().float().cpu() if k in model.store.params else dict(model.named_parameters())[k].grad.float().cpu()
        e = rel(go, g['grads'][k]) if 'grads' in g else abs(go.double().norm().item() - gn) / (gn + 1e-30)
        assert e <= GRAD_TOL, f'gradient {k}: rel err {e:.3e}'
        num += e * gn; den += gn
    print(f'  b=64 gradients: norm-weighted mean rel err {num / den:.3e}')
    _log_parity('canon512_b64', loss_rel=abs(out['loss'] - float(g['loss'])) / max(1., abs(float(g['loss']))), logits_rel=worst, argmax_unfiltered=agree, grad_mean_rel=num / den)
    assert num / den <= GRAD_MEAN_TOL


def test_default_times_and_cfg_drop_match_reference_golden():
    """The DEFAULT training call - `model(batch)` with no `times=` and the CFG text drop active - is what bench.py times.  Golden
    `cfg1.pt` (oracle/make_golden_cfg.py): the reference's own call with its three uniform draws replaced by fixed vectors.  The native
    path draws the same three vectors (times first: T:193, T:197; then the drop mask, T:3030) and must reproduce the reference's
    `times` EXACTLY (default_modality_length_to_time_fn, T:186-200) and its loss / gradients (dropped rows: every user token, [sos],
    [eos] -> null id, T:3032-3043; null labels ignored, T:3322-3323) within the usual tolerances."""
    from oracle.make_golden_cfg import cfg_case, patched_uniforms
    g = torch.load(os.path.join(GOLDEN, 'cfg1.pt'), weights_only=False)
    cfg, sd, batch, noise, draws = cfg_case()
    from transfusion_pytorch_amd import Transfusion
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads), prob_uncond=g['prob_uncond'])
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    with patched_uniforms([draws['u_k'], draws['u_t'], draws['u_cfg']]) as pu:
        loss, bd, times = model(batch, return_breakdown=True, return_times=True)
        assert not pu.queue, 'the native forward must consume exactly the reference\'s three uniform draws'
    loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(times.cpu(), g['times']), (times, g['times'])
    out = dict(loss=float(loss), text=float(bd.text), flow=[float(f) for f in bd.flow])
    compare_losses(out, g)
    worst, wsum, nsum = 0., 0., 0.
    for k, p in model.named_parameters():
        if k not in g['grad_norms'] or g['grad_norms'][k] < 1e-7:
            continue
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][k])
        gn = float(p.grad.double().norm())
        assert abs(gn - g['grad_norms'][k]) <= GRAD_TOL * g['grad_norms'][k], (k, gn, g['grad_norms'][k])
        worst = max(worst, r); wsum += r * g['grad_norms'][k]; nsum += g['grad_norms'][k]
    print(f'  gradients: worst head rel {worst:.3e}, norm-weighted mean {wsum / nsum:.3e}')
    assert worst <= GRAD_HEAD_TOL and wsum / nsum <= GRAD_MEAN_TOL
    # eval(): no drop (T:3029 is gated on self.training) - the loss changes
    model.eval()
    with patched_uniforms([draws['u_k'], draws['u_t']]):
        loss_eval = model(batch)
    assert abs(float(loss_eval) - float(g['loss'])) > 1e-2


def test_tiny_matches_live_oracle_and_updates():
    """same inputs through the CPU oracle on the GPU box's host cores, then one fused clip+Adam step vs torch Adam."""
    name = 'tiny1'
    cfg, sd, batch, times, noise = build_case(name)
    sdg = with_grad(sd)
    ref = forward_train(sdg, cfg, batch, times, noise, return_all=True)
    ref['loss'].backward()
    cfg, model, out = run_native(name)
    assert abs(out['loss'] - float(ref['loss'])) <= 2e-3 * max(1., abs(float(ref['loss'])))
    for k, p in sdg.items():
        if p.requires_grad:
            assert rel(out['grads'][k], p.grad) <= GRAD_TOL, k
    # fused optimizer step vs torch.optim.Adam + clip_grad_norm_ on the SAME (native) gradients
    from transfusion_pytorch_amd.optim import FusedAdam
    params = [p for p in model.parameters() if p.requires_grad]
    ref_params = [p.detach().clone().requires_grad_(True) for p in params]
    for rp, p in zip(ref_params, params):
        rp.grad = p.grad.detach().clone()
    torch.nn.utils.clip_grad_norm_(ref_params, 0.5)
    topt = torch.optim.Adam(ref_params, lr=3e-4)
    topt.step()
    opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
    opt.step()
    torch.cuda.synchronize()
    for rp, p in zip(ref_params, params):
        assert torch.allclose(rp, p.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('dim,depth,heads,dls,dim_head', [(192, 3, 3, (40,), 64), (320, 5, 2, (24, 72), 64), (576, 2, 1, (8,), 64), (64, 1, 4, (16,), 64),
                                                          (128, 2, 4, (16,), 8), (192, 3, 2, (24, 40), 32), (64, 2, 8, (8,), 16), (128, 1, 3, (16,), 48)])
def test_odd_configurations_match_live_oracle(dim, depth, heads, dls, dim_head):
    """shapes off the golden grid - odd depth (U-Net skip pairing, T:1206-1219), heads * 64 != dim, dim not a multiple of 128 /
    above 512 (two column chunks per lane), tiny latents, dim_head 8 (train_toy.py, the reference's tests) / 16 / 32 (its image
    examples) / 48 run zero-padded to the kernels' 64 columns per head - against the CPU oracle (pinned to the reference) on the same inputs."""
    from oracle import detdata as D
    from oracle.transfusion_oracle import OracleConfig
    cfg = OracleConfig(num_text_tokens=96, dim=dim, depth=depth, dim_latents=dls, heads=heads, dim_head=dim_head)
    tag = f'odd/{dim}/{depth}/{heads}' + (f'/{dim_head}' if dim_head != 64 else '')
    batch = D.ragged_batch(f'{tag}/b', 3, cfg.num_text_tokens, cfg.dim_latents)
    times = D.det_times(f'{tag}/t', batch); noise = D.det_noise(f'{tag}/n', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=tag)
    sdg = with_grad(sd)
    ref = forward_train(sdg, cfg, batch, times, noise, return_all=True)
    ref['loss'].backward()
    model = build_native(cfg, sd)
    model.train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    loss = model(batch, times=times)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  dim {dim} depth {depth} heads {heads} dim_head {dim_head}: loss native {float(loss.detach()):.6f} oracle {float(ref["loss"].detach()):.6f}')
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) <= 2e-3 * max(1., abs(float(ref['loss'].detach())))
    plan = model._live[0]
    assert rel(plan.logits.view(plan.b, plan.n, -1)[:, :model._live_n_true, :cfg.vocab].float().cpu(), ref['logits'].detach()) <= LOGIT_TOL
    wsum = nsum = 0.
    for k, p in model.named_parameters():
        gr = sdg[k].grad if sdg[k].requires_grad else None
        if gr is None or float(gr.norm()) < 1e-7:
            continue
        r = rel(p.grad, gr)
        assert r <= GRAD_HEAD_TOL, (k, r)
        wsum += r * float(gr.norm()); nsum += float(gr.norm())
    assert wsum / nsum <= GRAD_MEAN_TOL


def test_identity_latent_projection_matches_live_oracle():
    """dim_latent == dim: the reference uses nn.Identity for latent_to_model (T:1478) - noised latents enter the stream unprojected.
    Checked against the CPU oracle (pinned to the reference) on the same inputs."""
    from oracle import detdata as D
    from oracle.transfusion_oracle import OracleConfig
    cfg = OracleConfig(num_text_tokens=256, dim=64, depth=2, dim_latents=(64,), heads=2, dim_head=64)
    batch = D.ragged_batch('ident/b', 3, cfg.num_text_tokens, cfg.dim_latents)
    times = D.det_times('ident/t', batch); noise = D.det_noise('ident/n', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='ident')
    assert 'latent_to_model_projs.0.weight' not in sd
    sdg = with_grad(sd)
    ref = forward_train(sdg, cfg, batch, times, noise, return_all=True)
    ref['loss'].backward()
    model = build_native(cfg, sd)
    model.train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    loss = model(batch, times=times)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach()) - float(ref['loss'])) <= 2e-3 * max(1., abs(float(ref['loss'])))
    for k, p in model.named_parameters():
        if sdg[k].requires_grad and sdg[k].grad is not None and float(sdg[k].grad.norm()) > 1e-7:
            assert rel(p.grad, sdg[k].grad) <= GRAD_TOL, k


def test_forward_text_matches_reference_golden():
    """`forward_text` (T:2586-2664) against the reference's golden loss / logits / gradients
    (tests/golden/text1.pt, oracle/make_golden_text.py).  Same tolerances as the interleaved path."""
    from oracle.cases import build_text_case
    cfg, sd, text = build_text_case('text1')
    g = torch.load(os.path.join(GOLDEN, 'text1.pt'))
    model = build_native(cfg, sd)
    model.train()
    loss = model.forward_text(text)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  loss native {float(loss):.6f} reference {float(g["loss"]):.6f}')
    assert abs(float(loss) - float(g['loss'])) <= 2e-3 * max(1., abs(float(g['loss'])))
    plan = model._live[0]
    embed = plan.embed.view(plan.b, plan.n, -1).float().cpu()
    assert rel(embed, g['embed']) <= LOGIT_TOL
    worst, wsum, nsum = 0., 0., 0.
    for k, p in model.named_parameters():
        if k not in g['grad_norms'] or g['grad_norms'][k] < 1e-7:
            continue
        assert p.grad is not None, k
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][k])
        gn = float(p.grad.double().norm())
        assert abs(gn - g['grad_norms'][k]) <= GRAD_TOL * g['grad_norms'][k], (k, gn, g['grad_norms'][k])
        worst = max(worst, r); wsum += r * g['grad_norms'][k]; nsum += g['grad_norms'][k]
    print(f'  gradients: worst head rel {worst:.3e}, norm-weighted mean {wsum / nsum:.3e}')
    assert worst <= GRAD_HEAD_TOL and wsum / nsum <= GRAD_MEAN_TOL
    with torch.no_grad():
        logits = model(text[:, :-1].cuda(), return_loss=False)             # tensor input routes to forward_text (T:2967)
    assert rel(logits.float().cpu(), g['logits']) <= LOGIT_TOL


def test_forward_modality_matches_reference_golden():
    """`forward_modality` (T:2710-2869) and `generate_modality_only` (T:2871-2923) against the reference's
    golden (tests/golden/flow1.pt, oracle/make_golden_modality.py): two modality types, axial shape (5, 7), type 1 trained."""
    from oracle.cases import MODALITY_CASES, build_modality_case
    from oracle.make_golden_modality import gen_noise
    cfg, sd, x, times, noise, ty = build_modality_case('flow1')
    shape = MODALITY_CASES['flow1'][2]
    g = torch.load(os.path.join(GOLDEN, 'flow1.pt'))
    model = build_native(cfg, sd)
    model.train()
    model._noise_override = {ty: noise.cuda()}
    loss, (flow_loss, _, _) = model.forward_modality(x, times=times, modality_type=ty, return_loss_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  loss native {float(loss.detach()):.6f} reference {float(g["loss"]):.6f}')
    assert abs(float(loss.detach()) - float(g['loss'])) <= 2e-3 * max(1., abs(float(g['loss'])))
    worst, wsum, nsum = 0., 0., 0.
    for k, p in model.named_parameters():
        if k not in g['grad_norms'] or g['grad_norms'][k] < 1e-7:
            continue
        assert p.grad is not None, k
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][k])
        gn = float(p.grad.double().norm())
        assert abs(gn - g['grad_norms'][k]) <= GRAD_TOL * g['grad_norms'][k], (k, gn, g['grad_norms'][k])
        worst = max(worst, r); wsum += r * g['grad_norms'][k]; nsum += g['grad_norms'][k]
    print(f'  gradients: worst head rel {worst:.3e}, norm-weighted mean {wsum / nsum:.3e}')
    assert worst <= GRAD_HEAD_TOL and wsum / nsum <= GRAD_MEAN_TOL
    with torch.no_grad():
        pred = model(x.cuda(), times=times, modality_type=ty, return_loss=False)          # float tensor routes to forward_modality (T:2989)
    assert pred.shape == x.shape and rel(pred.cpu(), g['pred_noloss']) <= LOGIT_TOL
    # with an EMA teacher (T:2716-2859): student noised at t (1 - delta), term = mse(flow target, teacher flow at t + delta)
    from oracle import detdata as D
    teacher = build_native(cfg, D.det_state_dict(cfg.state_dict_shapes(), tag='flow1/teacher')).eval()
    model.zero_grad(set_to_none=True)
    vloss, (vflow, vvel, _) = model.forward_modality(x, times=times, modality_type=ty, velocity_consistency_ema_model=teacher,
                                                     velocity_consistency_delta_time=g['vc_delta'], return_loss_breakdown=True)
    vloss.backward()
    print(f'  with teacher: loss {float(vloss.detach()):.6f} / {float(g["vc_loss"]):.6f}, flow {float(vflow):.6f} / {float(g["vc_flow"]):.6f}, '
          f'velocity {float(vvel):.6f} / {float(g["vc_velocity"]):.6f}')
    for a, r in ((vloss.detach(), g['vc_loss']), (vflow, g['vc_flow']), (vvel, g['vc_velocity'])):
        assert abs(float(a) - float(r)) <= 3e-3 * max(1., abs(float(r)))
    with pytest.raises(ValueError):
        model.forward_modality(x, times=times, modality_type=ty, velocity_consistency_ema_model=model)
    model._gen_noise_override = gen_noise('flow1', 2, shape, cfg.dim_latents[ty])
    gen = model.generate_modality_only(batch_size=2, modality_type=ty, fixed_modality_shape=tuple(shape), modality_steps=g['gen_steps'])
    r = rel(gen.cpu(), g['gen'])
    print(f'  generate_modality_only ({g["gen_steps"]} grid points): rel {r:.3e}')
    assert r <= 3e-2


def test_velocity_consistency_matches_reference_golden():
    """`forward(..., velocity_consistency_ema_model=teacher)` (T:3084-3088, T:3378-3418) against the reference's
    golden (tests/golden/velocity1.pt): student and teacher with different weights and different injected noise; plus the EMA
    wrapper (`create_ema`, fused tfx_ema_update) against a torch lerp."""
    from oracle.make_golden_velocity import DELTA, velocity_case
    cfg, sd, sd_t, batch, times, noise, noise_t = velocity_case()
    g = torch.load(os.path.join(GOLDEN, 'velocity1.pt'))
    student, teacher = build_native(cfg, sd), build_native(cfg, sd_t)
    student.train(); teacher.eval()
    student._noise_override = {t: v.cuda() for t, v in noise.items()}
    teacher._noise_override = {t: v.cuda() for t, v in noise_t.items()}
    loss, bd = student(batch, times=times, velocity_consistency_ema_model=teacher, velocity_consistency_delta_time=DELTA, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  loss native {float(loss.detach()):.6f} reference {float(g["loss"]):.6f}; velocity {[round(float(v), 5) for v in bd.velocity]} vs '
          f'{[round(float(v), 5) for v in g["velocity_losses"]]}')
    assert abs(float(loss.detach()) - float(g['loss'])) <= 2e-3 * max(1., abs(float(g['loss'])))
    for a, r in zip(bd.velocity, g['velocity_losses']):
        assert abs(float(a) - float(r)) <= 3e-3 * max(1., abs(float(r)))
    worst, wsum, nsum = 0., 0., 0.
    for k, p in student.named_parameters():
        if k not in g['grad_norms'] or g['grad_norms'][k] < 1e-7:
            continue
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][k])
        gn = float(p.grad.double().norm())
        assert abs(gn - g['grad_norms'][k]) <= GRAD_TOL * g['grad_norms'][k], (k, gn, g['grad_norms'][k])
        worst = max(worst, r); wsum += r * g['grad_norms'][k]; nsum += g['grad_norms'][k]
    print(f'  gradients: worst head rel {worst:.3e}, norm-weighted mean {wsum / nsum:.3e}')
    assert worst <= GRAD_HEAD_TOL and wsum / nsum <= GRAD_MEAN_TOL
    # EMA wrapper: after the warm-up copy, update() is ema = d*ema + (1-d)*online in one fused launch
    ema = student.create_ema()
    ema.update_after_step, ema.update_every = 0, 1
    ema.update(); ema.update()                                       # step 0: copy, step 1: copy + initted
    before = ema.ema_model.store.flat.clone()
    with torch.no_grad():
        student.store.flat.add_(0.01)
    ema.update()
    d = ema.get_current_decay()
    torch.cuda.synchronize()
    expect = before * d + student.store.flat * (1 - d)
    assert 0. < d < 1. and torch.allclose(ema.ema_model.store.flat, expect, rtol=1e-6, atol=1e-7)
    out = ema(batch, times=times, return_loss=False)                 # forward goes to the averaged copy
    assert out.shape[0] == len(batch)


def test_model_output_clean_forward_modality_matches_reference_golden():
    """`model_output_clean=True` on the pure flow path (T:2772-2810): pred flow = (output - x_t) / max(1 - t, eps) in fp32, with the
    eps floor active for one sample (t = 0.995); golden from the reference (tests/golden/clean1.pt).  Then the INTERLEAVED step of the same
    model (model-space conversion, MP:786-792) against the reference's interleaved golden of the same fixture."""
    from oracle.make_golden_clean import clean_case
    from transfusion_pytorch_amd import Transfusion
    cfg, sd, batch, times, noise, xm, tm, nm = clean_case()
    g = torch.load(os.path.join(GOLDEN, 'clean1.pt'))
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents, model_output_clean=True, eps=cfg.eps,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads), prob_uncond=0.)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model._noise_override = {1: nm.cuda()}
    loss = model.forward_modality(xm, times=tm, modality_type=1)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  loss native {float(loss.detach()):.4f} reference {float(g["mod_loss"]):.4f}')
    assert abs(float(loss.detach()) - float(g['mod_loss'])) <= 5e-3 * abs(float(g['mod_loss']))      # 1/(1-t) = 100 amplifies the bf16 floor
    wsum = nsum = 0.
    for k, p in model.named_parameters():
        if k not in g['mod_grad_norms'] or g['mod_grad_norms'][k] < 1e-6:
            continue
        r = rel(p.grad.float().reshape(-1)[:1024], g['mod_grad_head'][k])
        wsum += r * g['mod_grad_norms'][k]; nsum += g['mod_grad_norms'][k]
    print(f'  gradients: norm-weighted mean rel {wsum / nsum:.3e}')
    assert wsum / nsum <= 3e-2
    with torch.no_grad():
        pred = model.forward_modality(xm, times=tm, modality_type=1, return_loss=False)
    assert rel(pred.cpu(), g['mod_pred_noloss']) <= 2e-2
    # the interleaved step converts in MODEL space against the projected noised tokens (MP:786-792): (W embed - W proj) / max(1 - t, eps), the
    # subtraction between two fp32 GEMM results; golden from the reference's own interleaved call (same fixture)
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    model.zero_grad(set_to_none=True)
    loss, bd = model(batch, times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f'  interleaved: loss native {float(loss.detach()):.4f} reference {float(g["loss"]):.4f}; flows {[round(float(f), 3) for f in bd.flow]} vs '
          f'{[round(float(f), 3) for f in g["flow_losses"]]}')
    assert abs(float(loss.detach()) - float(g['loss'])) <= 5e-3 * abs(float(g['loss']))                # 1 / (1 - t) up to 100 amplifies the bf16 floor
    for a, r in zip(bd.flow, g['flow_losses']):
        assert abs(float(a) - float(r)) <= 8e-3 * abs(float(r))
    assert abs(float(bd.text) - float(g['text_loss'])) <= 2e-3 * abs(float(g['text_loss']))
    wsum = nsum = worst = 0.
    for k, p in model.named_parameters():
        if k not in g['grad_norms'] or g['grad_norms'][k] < 1e-6:
            continue
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][k])
        gn = float(p.grad.double().norm())
        assert abs(gn - g['grad_norms'][k]) <= 6e-2 * g['grad_norms'][k], (k, gn, g['grad_norms'][k])
        worst = max(worst, r); wsum += r * g['grad_norms'][k]; nsum += g['grad_norms'][k]
    print(f'  interleaved gradients: worst head rel {worst:.3e}, norm-weighted mean rel {wsum / nsum:.3e}')
    assert wsum / nsum <= 3e-2


def test_native_list_replay_equals_per_launch_calls():
    """tfx_run_list (one call per list, the product path) against one C-ABI call per launch on the same plan: same launches in the
    same order, so loss and gradients agree to fp32-atomic ordering noise."""
    from transfusion_pytorch_amd.engine import LaunchList, Plan
    outs = []
    orig = Plan.run
    for per_launch in (False, True):
        if per_launch:
            Plan.run = staticmethod(lambda launches, stream, lo=0, hi=None, graph=False: orig(list(launches), stream, lo, hi))
        try:
            cfg, model, out = run_native('small2')
        finally:
            Plan.run = orig
        assert isinstance(model._live[0].fwd, LaunchList) and model._live[0].fwd._image is not None or per_launch
        outs.append(out)
    a, b = outs
    assert abs(a['loss'] - b['loss']) <= 1e-5 * max(1., abs(b['loss']))
    assert rel(a['logits'], b['logits']) <= 1e-6
    for k in a['grads']:
        assert rel(a['grads'][k], b['grads'][k]) <= 1e-3, k


def test_side_stream_weight_gradients_equal_single_stream():
    """the weight-gradient GEMMs run on the library's side stream, one layer behind the data-gradient chain (fork / join events,
    per-wrapper + parity-double-buffered operands).  Same plan replayed on ONE stream (tfx_set_single_stream): identical launches,
    so loss and every gradient agree to fp32-atomic ordering noise - at the canonical size, where the two streams really overlap."""
    from transfusion_pytorch_amd import capi
    outs = []
    for single in (1, 0, 0):
        capi.lib().tfx_set_single_stream(single)
        try:
            cfg, model, out = run_native('canon512')
        finally:
            capi.lib().tfx_set_single_stream(0)
        outs.append(out)
    ref = outs[0]
    for o in outs[1:]:
        assert abs(o['loss'] - ref['loss']) <= 1e-6 * max(1., abs(ref['loss']))
        for k in ref['grads']:
            assert rel(o['grads'][k], ref['grads'][k]) <= 2e-3, k


@pytest.mark.parametrize('name', ['small2', 'canon512'])
def test_fused_qk_norm_rope_backward_equals_the_separate_launch(name, monkeypatch):
    """round 5: with TFX_ATTN_QKNR=1 (default) a layer's backward runs QK-RMSNorm + RoPE backwards inside the attention-backward epilogues; TFX_ATTN_QKNR=0
    is the round-4 list (attention backward writes d q~ | d k~, tfx_qk_norm_rope_bwd reads them back).  Same loss, every gradient equal to the rounding of
    one bf16 intermediate (`small2`: ragged lengths - wave blocks past a sample's end; `canon512`: the bench's model)."""
    outs = []
    for mode in ('0', '1'):
        monkeypatch.setenv('TFX_ATTN_QKNR', mode)
        cfg, model, out = run_native(name)
        n_sep = sum(1 for it in model._live[0].bwd if it[0] == 'tfx_qk_norm_rope_bwd')
        assert n_sep == (cfg.depth if mode == '0' else 0), 'the switch must change the backward list'
        outs.append(out)
    ref, fus = outs
    assert abs(fus['loss'] - ref['loss']) <= 1e-6 * max(1., abs(ref['loss']))
    # At kernel level the two forms agree to 0 ... 2.4e-5 on d q | d k (tests/test_kernels_gpu.py: a handful of one-ulp bf16 flips).  What this test sees is what
    # those flips and a different launch timing (fp32 atomics of the weight-gradient GEMMs and column sums arrive in another order; the side-stream test allows the
    # same 2e-3 for IDENTICAL launches) do to small-norm gradients after 8 layers: worst 2.5e-3 (`small2`) / 5.5e-3 (`canon512`: a layer-0 LayerNorm gain gradient
    # of norm 2e-3), norm-weighted mean 4.6e-4 / 1.2e-3
    errs = {k: rel(fus['grads'][k], ref['grads'][k]) for k in ref['grads']}
    norms = {k: float(ref['grads'][k].double().norm()) for k in ref['grads']}
    worst = max(errs.values())
    mean = sum(errs[k] * norms[k] for k in errs) / sum(norms.values())
    print(f'  fused vs separate QK-norm / RoPE backward: worst gradient rel difference {worst:.2e}, norm-weighted mean {mean:.2e}')
    assert mean <= 3e-3
    for k in errs:
        assert errs[k] <= 1e-2, k


def test_training_lists_replay_as_graphs_while_their_fingerprint_holds(monkeypatch):
    """TFX_TRAIN_GRAPH=1: the forward / backward launch lists of a fixed-shape training loop run as ONE hipGraph launch each from the third step
    on (LaunchList.replay_auto: captured once `tfx_list_fingerprint` - every byte a capture freezes - has repeated), the side-stream
    weight-gradient GEMMs in list order on the one chain.  Same launches, same arguments: loss and gradients equal the list replay's to
    fp32-atomic ordering noise; a step whose scalars differ (another loss weight -> other seeds in the args structs) drops the graph, runs
    through the list, and is captured again when it repeats."""
    from transfusion_pytorch_amd import engine
    cfg, sd, batch, times, noise = build_case('canon512')

    def steps(model, n):
        outs = []
        for _ in range(n):
            for p in model.parameters():
                p.grad = None
            loss = model(batch, times=times)
            loss.backward()
            torch.cuda.synchronize()
            outs.append((float(loss), {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}))
        return outs

    def fresh():
        m = build_native(cfg, sd).train()
        m._noise_override = {t: v.cuda() for t, v in noise.items()}
        return m

    ref_loss, ref_grads = steps(fresh(), 1)[0]
    monkeypatch.setattr(engine, 'TRAIN_GRAPH', True)
    model = fresh()
    outs = steps(model, 5)
    plan = model._live[0]
    st_f, st_b = plan.fwd._auto[(0, len(plan.fwd))], plan.bwd._auto[(0, len(plan.bwd))]
    assert st_f[1] and st_b[1] and st_f[2] >= 4, 'the fixed-shape loop must end up on graph replays'
    for i, (l, g) in enumerate(outs):
        assert abs(l - ref_loss) <= 1e-6 * max(1., abs(ref_loss)), i
        for k in ref_grads:
            assert rel(g[k], ref_grads[k]) <= 2e-3, (i, k)
    # a changed scalar: other loss seeds -> another fingerprint -> list replay, then a new capture
    model.flow_loss_weight = 2.0 * model.flow_loss_weight
    l2 = steps(model, 1)[0][0]
    assert plan.fwd._auto[(0, len(plan.fwd))][1] is None and abs(l2 - ref_loss) > 1e-4
    l3 = [o[0] for o in steps(model, 3)]
    assert plan.fwd._auto[(0, len(plan.fwd))][1] and all(abs(x - l2) <= 1e-6 * max(1., abs(l2)) for x in l3)


def test_upstream_gradient_scales_seeds_once_and_second_backward_raises():
    """`(loss / 3).backward()`: the upstream gradient reaches the loss seeds as an fp32 device scalar (tfx_scale_bf16_dev) - every gradient is
    exactly the plain one / 3 up to bf16 rounding of the seeds; a second backward of the same forward raises instead of double-scaling."""
    cfg, sd, batch, times, noise = build_case('small2')
    grads = []
    for div in (1., 3.):
        model = build_native(cfg, sd).train()
        model._noise_override = {t: v.cuda() for t, v in noise.items()}
        loss = model(batch, times=times)
        (loss / div).backward(retain_graph=True)
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None})
        with pytest.raises(RuntimeError, match='second time'):
            loss.backward()
    for k in grads[0]:
        if float(grads[0][k].norm()) > 1e-7:
            assert rel(grads[1][k] * 3., grads[0][k]) <= 2e-2, k          # the seeds are re-rounded to bf16 after the scale: noise, not a factor


def test_overlapped_gradient_exchange_equals_plain_step_rccl_world1():
    """data-parallel step with the gradient all-reduce cut into layer groups and issued DURING the backward (optim.GradReducer, engine
    `bwd_cuts`) on a real RCCL communicator (world size 1 on this box; world 2 runs on gloo in tests/test_dp_gloo.py): loss, every gradient and
    the post-step parameters equal the plain path's (per-layer AdaLN weight-gradient GEMMs instead of one: fp32 atomic ordering noise only)."""
    import torch.distributed as dist
    from transfusion_pytorch_amd.optim import FusedAdam
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        cfg, sd, batch, times, noise = build_case('small2')
        res = []
        for overlap in (False, True):
            model = build_native(cfg, sd).train()
            model._noise_override = {t: v.cuda() for t, v in noise.items()}
            opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
            opt.always_sync = True
            if overlap:
                opt.overlap_grad_sync(groups=3)
            loss = model(batch, times=times)
            loss.backward()
            plan = model._live[0]
            assert bool(plan.bwd_cuts) == overlap and (not overlap or len(plan.bwd_cuts) == 2)        # depth 4, groups of 2 layers
            grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
            opt.step()
            torch.cuda.synchronize()
            res.append((float(loss), grads, model.store.flat.clone()))
        (l0, g0, p0), (l1, g1, p1) = res
        assert abs(l0 - l1) <= 1e-6 * max(1., abs(l0))
        for k in g0:
            assert rel(g1[k], g0[k]) <= 2e-3, k
        assert rel(p1, p0) <= 1e-5
    finally:
        if own:
            dist.destroy_process_group()


def test_training_graphs_under_the_overlapped_exchange_rccl_world1(monkeypatch):
    """TFX_TRAIN_GRAPH=1 together with the overlapped gradient exchange: the backward list is replayed in segments (one per exchange group), and
    a segment that ends between a side-stream fork and its join cannot be captured on its own - the capture fails cleanly, that range stays on
    `tfx_run_list` for good, the others become graphs.  Five identical steps: losses and post-step parameters equal the list replay's."""
    import torch.distributed as dist
    from transfusion_pytorch_amd import engine
    from transfusion_pytorch_amd.optim import FusedAdam
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29543')
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        cfg, sd, batch, times, noise = build_case('canon512')
        res = []
        for graphs in (False, True):
            monkeypatch.setattr(engine, 'TRAIN_GRAPH', graphs)
            model = build_native(cfg, sd).train()
            model._noise_override = {t: v.cuda() for t, v in noise.items()}
            opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
            opt.always_sync = True
            opt.overlap_grad_sync(groups=4)
            losses = []
            for _ in range(5):
                loss = model(batch, times=times)
                loss.backward()
                opt.step(); opt.zero_grad()
                losses.append(float(loss))
            torch.cuda.synchronize()
            plan = model._live[0]
            if graphs:
                states = {k: (bool(v[1]), v[2]) for k, v in plan.bwd._auto.items()}
                print('backward ranges (captured, replays):', states)
                assert len(states) == len(plan.bwd_cuts) + 1 and plan.fwd._auto[(0, len(plan.fwd))][1]
            res.append((losses, model.store.flat.clone()))
        (l0, p0), (l1, p1) = res
        # (the first step is the same forward: identical; afterwards the two runs drift apart like any two runs of this step do - Adam's first updates
        #  are +-lr whatever a gradient's size, so the fp32-atomic ordering noise of near-zero gradients moves single weights by 2 lr)
        assert abs(l0[0] - l1[0]) <= 1e-6 * max(1., abs(l0[0]))
        for a, b in zip(l0, l1):
            assert abs(a - b) <= 5e-4 * max(1., abs(a)), (l0, l1)
        assert rel(p1, p0) <= 1e-3
    finally:
        if own:
            dist.destroy_process_group()


def test_overlapped_exchange_world1_default_gates_and_accumulation_rccl():
    """ `torchrun --nproc-per-node 1`: torch.distributed is initialised at world size 1 and `always_sync` keeps its
    default - the backward still takes the overlapped path, so `step()` must wait for its handles and re-arm the reducer: two consecutive steps
    run.  Then gradient accumulation: micro-batches under `opt.no_sync()` only accumulate, the last backward exchanges - the step equals the
    plain exchange's two-backward step; without `no_sync` the second backward is refused.  Also the coalesced exchange (TFX_DP_COALESCE=1)."""
    import torch.distributed as dist
    from transfusion_pytorch_amd.optim import FusedAdam
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29542')
    own = not dist.is_initialized()
    if own:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        cfg, sd, batch, times, noise = build_case('small2')
        def fresh(overlap):
            model = build_native(cfg, sd).train()
            model._noise_override = {t: v.cuda() for t, v in noise.items()}
            opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
            if overlap:
                opt.overlap_grad_sync(groups=3)
            return model, opt
        # two steps with the default gates (the second one used to raise "second backward() before optimizer.step()")
        model, opt = fresh(True)
        for _ in range(2):
            loss = model(batch, times=times); loss.backward(); opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        assert not opt.reducer.exchanged and not opt.reducer.handles
        # accumulation of two micro-batches
        res = []
        for overlap in (False, True):
            model, opt = fresh(overlap)
            with opt.no_sync():
                model(batch, times=times).backward()
            loss = model(batch, times=times); loss.backward()
            if overlap:
                assert opt.reducer.exchanged and opt.reducer.launches == 4                  # depth 4 in groups of 2 layers, two ranges per group on the (default) public-API exchange: only the LAST backward sent them
            g = model.store.grad.clone()
            opt.step(); torch.cuda.synchronize()
            res.append((g, model.store.flat.clone()))
        assert rel(res[1][0], res[0][0]) <= 2e-3 and rel(res[1][1], res[0][1]) <= 1e-5
        model, opt = fresh(True)
        model(batch, times=times).backward()
        with pytest.raises(RuntimeError, match='no_sync'):
            model(batch, times=times).backward()
        # the coalesced exchange (torch's private `_coalescing_manager`, TFX_DP_COALESCE=1; the default is the public API since round 5): one launch per
        # group, same sums
        os.environ['TFX_DP_COALESCE'] = '1'
        try:
            model, opt = fresh(True)
            loss = model(batch, times=times); loss.backward()
            assert opt.reducer.launches == 2
            g = model.store.grad.clone(); opt.step(); torch.cuda.synchronize()
        finally:
            del os.environ['TFX_DP_COALESCE']
        model, opt = fresh(False)
        loss = model(batch, times=times); loss.backward()
        assert rel(g, model.store.grad) <= 2e-3
    finally:
        if own:
            dist.destroy_process_group()


def test_no_fallback_on_cpu():
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.capi import TfxError
    m = Transfusion(num_text_tokens=16, dim_latent=32, transformer=dict(dim=64, depth=1, heads=1))
    with pytest.raises(TfxError):
        m([[torch.randint(0, 16, (4,)), torch.randn(2, 32)]])


def test_channel_first_latents_with_frozen_encoder_decoder_match_reference_golden():
    """SURVEY 8(f) rank 4, first half: `channel_first_latent=True` + frozen `modality_encoder` / `modality_decoder` (T:1352, T:1405-1418,
    T:1481-1489, T:3094-3101) - host-side layout and pre / post-processing around the native path.  Golden tests/golden/f4_chfirst.pt
    (oracle/make_golden_f4.py): interleaved step, `return_only_pred_flows` shapes, forward_modality, generate_modality_only through the decoder;
    the reference checkpoint's `Sequential` key names (`latent_to_model_projs.0.1.*`, `model_to_latent_projs.0.0.*`) load unchanged."""
    from oracle.make_golden_f4 import enc_dec, f4_case
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(GOLDEN, 'f4_chfirst.pt'), weights_only=False)
    cfg, sd, batch, times, noise, xm, nm, tm, g0 = f4_case()
    enc, dec = enc_dec()
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=16, channel_first_latent=True, modality_default_shape=(4,),
                        modality_encoder=enc, modality_decoder=dec, prob_uncond=0.,
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    ref_keys = {k.replace('latent_to_model_projs.0.', 'latent_to_model_projs.0.1.').replace('model_to_latent_projs.0.', 'model_to_latent_projs.0.0.'): v
                for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(ref_keys, strict=False)
    assert not unexpected and all(k.startswith(('modality_encoder', 'modality_decoder')) for k in missing), (missing, unexpected)
    assert 'latent_to_model_projs.0.1.weight' in model.state_dict() and 'model_to_latent_projs.0.0.weight' in model.state_dict()
    model = model.cuda().train()
    model._noise_override = {0: noise[0].cuda()}
    loss, bd = model([[p.cuda() for p in s] for s in batch], times=times, return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    out = dict(loss=float(loss), text=float(bd.text), flow=[float(f) for f in bd.flow])
    compare_losses(out, g)
    wsum = nsum = 0.
    for k, p in model.named_parameters():
        kr = k.replace('latent_to_model_projs.0.', 'latent_to_model_projs.0.1.').replace('model_to_latent_projs.0.', 'model_to_latent_projs.0.0.')
        if kr not in g['grad_norms'] or g['grad_norms'][kr] < 1e-7:
            continue
        r = rel(p.grad.float().reshape(-1)[:1024], g['grad_head'][kr])
        assert abs(float(p.grad.double().norm()) - g['grad_norms'][kr]) <= GRAD_TOL * g['grad_norms'][kr], k
        wsum += r * g['grad_norms'][kr]; nsum += g['grad_norms'][kr]
    assert wsum / nsum <= GRAD_MEAN_TOL
    with torch.no_grad():
        flows = model([[p.cuda() for p in s] for s in batch], times=times, return_only_pred_flows=True)
    assert [tuple(f.shape) for f in flows[0]] == g['pred_flow_shapes']                  # channel-first (16, L)
    assert rel(flows[0][0].cpu(), g['pred_flow0']) <= 1.5e-2
    model._noise_override = {0: nm.movedim(1, -1).reshape(-1, 16).cuda()}               # the reference's (b, d, L) noise as channel-last rows
    lm = model.forward_modality(xm.cuda(), times=tm)
    assert abs(float(lm.detach()) - float(g['mod_loss'])) <= 2e-3 * max(1., abs(float(g['mod_loss'])))
    with torch.no_grad():
        pm = model.forward_modality(xm.cuda(), times=tm, return_loss=False)
    assert pm.shape == g['mod_pred_noloss'].shape and rel(pm.cpu(), g['mod_pred_noloss']) <= 1.5e-2
    model._gen_noise_override = g0
    gen = model.generate_modality_only(batch_size=2, fixed_modality_shape=(4,), modality_steps=g['gen_steps'])
    assert gen.shape == g['gen'].shape and rel(gen.cpu(), g['gen']) <= 2e-2               # decoded: (2, 3, 4)
    # sample_many: prompted raw modality is encoded, the decoded one comes back through the decoder, raw channel-first
    model.eval()
    res = model.sample_many([[torch.randint(0, 256, (5,)).cuda(), (0, torch.randn(3, 4).cuda())]], max_length=8, text_temperature=0., modality_steps=3,
                            fixed_modality_shape=(4,), force_modality_at_start=0, cfg_scale=1.)
    mods = [p for p in res[0] if isinstance(p, tuple)]
    assert len(mods) >= 2 and all(m[1].shape == (3, 4) for m in mods)


def test_plan_cache_is_bounded_by_bytes_and_inference_plans_share_layer_buffers(monkeypatch):
    """a plan owns every activation of its step: the cache evicts least-recently-used plans by total BYTES as well as by count, and an inference
    plan keeps one set of per-layer scratch buffers for all layers (a training plan keeps them per layer for the backward)"""
    from transfusion_pytorch_amd import Transfusion
    m = Transfusion(num_text_tokens=32, dim_latent=16, modality_default_shape=(4,), transformer=dict(dim=64, depth=4, dim_head=8, heads=2)).cuda()
    batch = lambda n: [[torch.randint(0, 32, (n,)).cuda(), torch.randn(4, 16).cuda()]]
    loss = m(batch(40)); loss.backward()
    train_plan = m._live[0]
    with torch.no_grad():
        m(batch(40), return_loss=False)
    infer_plan = next(p for p in m._plans.values() if p is not train_plan)
    assert train_plan.ua.shape[0] == 4 and infer_plan.ua.shape[0] == 1 and infer_plan.qkr.shape[0] == 4 and infer_plan.nbytes < 0.6 * train_plan.nbytes
    monkeypatch.setenv('TFX_PLAN_BUDGET_GB', str(2.5 * train_plan.nbytes / 2 ** 30))
    for n in (100, 170, 230, 300, 360):                       # five more bucketed lengths: only ~2 plans fit the budget
        m(batch(n)).backward()
        assert sum(p.nbytes for p in m._plans.values()) <= 2.5 * train_plan.nbytes * 4 and m._live[0] in m._plans.values()
    assert len(m._plans) <= 3


def test_large_vocabulary_multi_axial_latents_long_ragged_rows_match_live_oracle():
    """off-grid in other directions: a 5000-token vocabulary (logits / CE over 5134 columns, the one-hot embedding-gradient GEMM), 2-d and 1-d
    modality shapes in one batch (meta strings "2,3" / "5"), rows of 300+ tokens beside a 20-token row (several 128-query attention tiles with
    ragged ends, heavy padding) - against the CPU oracle on the same inputs."""
    from oracle import detdata as D
    from oracle.transfusion_oracle import OracleConfig
    cfg = OracleConfig(num_text_tokens=5000, dim=128, depth=2, dim_latents=(16, 24), heads=2, dim_head=64)
    tx = lambda key, n: D.det_randint(f'big/{key}', (n,), 0, 5000)
    batch = [[tx('a', 150), (0, D.det_normalish('big/m0', (2, 3, 16))), tx('b', 120), (1, D.det_normalish('big/m1', (5, 24))), tx('c', 40)],
             [tx('d', 20)],
             [(1, D.det_normalish('big/m2', (7, 24))), tx('e', 200), (0, D.det_normalish('big/m3', (3, 2, 16)))]]
    times = D.det_uniform('big/t', (3, 2), 0.05, 0.95)
    noise = {0: D.det_normalish('big/n0', (12, 16)), 1: D.det_normalish('big/n1', (12, 24))}
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='big')
    ref = forward_train(with_grad(sd), cfg, batch, times, noise, return_all=True)
    sdg = None
    model = build_native(cfg, sd).train()
    model._noise_override = {t: v.cuda() for t, v in noise.items()}
    loss = model(batch, times=times)
    print(f'  loss native {float(loss.detach()):.6f} oracle {float(ref["loss"].detach()):.6f}')
    assert abs(float(loss.detach()) - float(ref['loss'].detach())) <= 2e-3 * max(1., abs(float(ref['loss'].detach())))
    plan = model._live[0]
    assert rel(plan.logits.view(plan.b, plan.n, -1)[:, :model._live_n_true, :cfg.vocab].float().cpu(), ref['logits'].detach()) <= LOGIT_TOL


def test_ragged_batches_share_one_bucketed_training_plan(monkeypatch):
    """Ragged corpora change the instance and latent-row counts with every batch: the training plan is built for counts rounded up (instances to 64,
    rows to 256) and shared.  (1) two batches of different structure run through ONE plan; (2) each gives the loss and the gradients of its own
    exact-count plan (TFX_PLAN_BUCKETS=0) - padding rows / instances contribute nothing."""
    from transfusion_pytorch_amd import Transfusion
    torch.manual_seed(0)
    model = Transfusion(num_text_tokens=64, dim_latent=(24, 16), modality_default_shape=((3,), (2,)), transformer=dict(dim=128, depth=2, dim_head=64, heads=2),
                        prob_uncond=0.).cuda().train()
    g = torch.Generator(device='cuda').manual_seed(3)
    T_ = lambda n: torch.randint(0, 64, (n,), device='cuda', generator=g)
    Lt = lambda t, n: (t, torch.randn(n, (24, 16)[t], device='cuda', generator=g))
    batches = [[[T_(5), Lt(0, 3), T_(7), Lt(1, 2), T_(3)], [T_(9), Lt(0, 4)]],
               [[Lt(1, 5), T_(11)], [T_(4), Lt(0, 2), T_(6), Lt(0, 6), T_(2), Lt(1, 1)], ]]
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('TFX_PLAN_BUCKETS', mode)
        model._plans, model._struct_cache = {}, {}
        plans = []
        for k, batch in enumerate(batches):
            times = torch.full((2, 3), 0.4, device='cuda')
            noise = {t: torch.randn(sum(p[1].shape[0] for s in batch for p in s if isinstance(p, tuple) and p[0] == t), (24, 16)[t],
                                    device='cuda', generator=torch.Generator(device='cuda').manual_seed(10 * k + t)) for t in (0, 1)}
            model._noise_override = noise
            for p in model.parameters():
                p.grad = None
            loss = model(batch, times=times)
            loss.backward()
            torch.cuda.synchronize()
            plans.append(model._live[0])
            out[(mode, k)] = (float(loss), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
        if mode == '1':
            assert plans[0] is plans[1] and plans[0].R == {0: 256, 1: 256} and plans[0].I == 64       # one plan for both structures
        else:
            assert plans[0] is not plans[1]
    model._noise_override = None
    for k in range(2):
        lb, gb = out[('1', k)]; le, ge = out[('0', k)]
        assert abs(lb - le) <= 1e-5 * max(1., abs(le)), (k, lb, le)
        worst = max(rel(gb[n], ge[n]) for n in ge if float(ge[n].norm()) > 1e-8)
        print(f'batch {k}: loss {lb:.6f} / {le:.6f}, worst gradient rel diff bucketed vs exact {worst:.2e}')
        assert worst <= 2e-3
