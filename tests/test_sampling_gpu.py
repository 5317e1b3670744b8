"""Decode-path parity: native KV-cached `sample_many` / `sample_one` vs the golden sequences the UNMODIFIED reference
produced (oracle/make_golden_sampling.py: deterministic weights, prompts and initial noise, greedy text).

bf16 vs fp32 cannot promise identical greedy tokens at near-ties on random-like weights, so the test pins:
  * the sampled token sequence up to the first reference near-tie (in practice: the whole sequence, see printed report),
  * every decoded modality (midpoint ODE with / without CFG) while the two histories still agree: rel-Frobenius <= 5e-2,
  * sample_one == sample_many for one prompt (the reference's own equivalence test, tests/test_transfusion.py:758-808),
  * self-consistency of the KV cache: teacher-forced full-sequence logits reproduce every cached greedy decision.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.make_golden_sampling import sampling_case      # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'sampling.pt')


def native_model():
    from transfusion_pytorch_amd import Transfusion
    cfg, sd, prompts, noise = sampling_case()
    m = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents[0], modality_default_shape=(4,),
                    transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    m.load_state_dict(sd)
    return m.cuda().eval(), prompts, noise


def plain(sample):
    return [('mod', int(p[0]), p[1].float().cpu()) if isinstance(p, tuple) else ('text', p.cpu().long()) for p in sample]


def compare(native, ref):
    """returns (tokens compared, tokens equal before first divergence, list of modality rel errors while histories agree, diverged?)"""
    n_tok = n_eq = 0
    mod_errs = []
    for pn, pr in zip(native, ref):
        if pn[0] != pr[0]:
            return n_tok, n_eq, mod_errs, True
        if pn[0] == 'mod':
            if pn[2].shape != pr[2].shape:
                return n_tok, n_eq, mod_errs, True
            mod_errs.append(((pn[2] - pr[2]).norm() / pr[2].norm()).item())
            continue
        a, b = pn[1].tolist(), pr[1].tolist()
        for x, y in zip(a, b):
            n_tok += 1
            if x != y:
                return n_tok, n_eq, mod_errs, True
            n_eq += 1
        if len(a) != len(b):
            return n_tok, n_eq, mod_errs, True
    return n_tok, n_eq, mod_errs, len(native) != len(ref)


@pytest.mark.parametrize('run,kw', [('free', {}), ('forced', dict(force_modality_at_start=0)), ('forced_nocfg', dict(force_modality_at_start=0, cfg_scale=1.))])
def test_sample_many_matches_reference_golden(run, kw):
    g = torch.load(GOLDEN, weights_only=False)
    m, prompts, noise = native_model()
    kwargs = dict(max_length=12, text_temperature=0., init_modality_noise=noise, modality_steps=4, fixed_modality_shape=(4,), cfg_scale=3.)
    kwargs.update(kw)
    outs = m.sample_many([p if not isinstance(p, list) else list(p) for p in prompts], **kwargs)
    total = eq = 0
    n_div = 0
    for i, (o, r) in enumerate(zip(outs, g['runs'][run])):
        n_tok, n_eq, errs, div = compare(plain(o), r)
        print(f'[{run}] sample {i}: {n_eq}/{n_tok} tokens identical before first divergence; modality rel errs {["%.2e" % e for e in errs]}; diverged={div}')
        total += n_tok; eq += n_eq; n_div += int(div)
        for e in errs:
            assert e <= 5e-2
        if run != 'free':
            assert len(errs) >= 1, 'the forced modality must have been decoded and compared'
    # random-like weights give near-uniform logits: allow at most one sample of the four to leave the reference's greedy path
    assert n_div <= 1, f'{n_div} of 4 samples diverged from the reference greedy path'


def test_sample_one_equals_sample_many_and_cache_is_consistent():
    m, prompts, noise = native_model()
    kwargs = dict(max_length=10, text_temperature=0., init_modality_noise=noise, modality_steps=4, fixed_modality_shape=(4,), cfg_scale=3.,
                  force_modality_at_start=0)
    many = m.sample_many([prompts[0], prompts[1]], **kwargs)
    one = m.sample_one(prompts[0], **kwargs)
    for a, b in zip(plain(many[0]), plain(one)):
        assert a[0] == b[0]
        if a[0] == 'text':
            assert a[1].tolist() == b[1].tolist()
        else:
            assert torch.allclose(a[2], b[2], atol=2e-2, rtol=2e-2)
    # teacher forcing: a text-only continuation decoded with the cache must be reproduced by one full forward
    out = m.sample_many([prompts[0]], max_length=10, text_temperature=0.)[0]
    seq = torch.cat([p for p in out if not isinstance(p, tuple)])
    if all(not isinstance(p, tuple) for p in out):
        plan, S = m._forward_plain([[seq]], torch.ones(1, 1, device='cuda'), add_meta=False)
        logits = plan.logits.view(1, S['n'], -1)[0, :seq.numel(), :m.md.vocab]
        n_prompt = 1 + prompts[0].numel()
        for i in range(n_prompt - 1, seq.numel() - 1):
            top = logits[i].max().item()
            assert logits[i, seq[i + 1]].item() >= top - 0.05, f'cached decision at position {i} is not the full-forward argmax'


def test_generate_text_only_matches_reference_greedy():
    """KV-cached `generate_text_only` (T:2666-2707) at temperature 0 against the reference's greedy tokens
    (tests/golden/text1.pt).  bf16 may flip a near-tie, after which a greedy sequence legitimately diverges: every token must
    match up to (not including) the first step whose REFERENCE top-2 margin is below 0.05."""
    import os
    import torch
    from oracle.cases import build_text_case
    from transfusion_pytorch_amd import Transfusion
    cfg, sd, _ = build_text_case('text1')
    g = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'text1.pt'))
    model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents[0],
                        transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads))
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    prompt = g['gen_prompt']
    gen = model.generate_text_only(prompt, prompt.shape[1] + g['gen_tokens'].shape[1], temperature=0.).cpu()
    assert gen.shape == g['gen_tokens'].shape
    checked = 0
    for b in range(gen.shape[0]):
        weak = (g['gen_margin'][b] < 0.05).nonzero()
        upto = int(weak[0]) if len(weak) else gen.shape[1]
        assert torch.equal(gen[b, :upto], g['gen_tokens'][b, :upto]), (b, gen[b], g['gen_tokens'][b])
        checked += upto
    print(f'  greedy tokens identical on {checked} decisive steps; overall agreement {(gen == g["gen_tokens"]).float().mean():.3f}')
    assert checked >= 8
    # the cached decode must agree with an uncached teacher-forced forward of the same model
    full = torch.cat((prompt, gen), dim=-1).cuda()
    lg = model.forward_text(full[:, :-1], return_loss=False)[:, prompt.shape[1] - 1:]
    agree = (lg.argmax(-1).cpu() == gen).float().mean()
    assert agree >= 0.95, agree


def test_small_heads_cached_decode_is_consistent_with_the_full_forward():
    """dim_head 8 (train_toy.py / the reference's tests) runs zero-padded to 64 columns per head: the KV cache, the rotary tables
    and the decode plans must agree with the full forward - every greedy token decoded with the cache is the full forward's argmax
    (0.05 logit slack for bf16 near-ties)."""
    import torch
    from transfusion_pytorch_amd import Transfusion
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=64, dim_latent=16, transformer=dict(dim=128, depth=2, dim_head=8, heads=4)).cuda().eval()
    prompt = torch.randint(0, 64, (3, 9), device='cuda')
    gen = m.generate_text_only(prompt, 9 + 14, temperature=0.)
    seq = torch.cat([prompt, gen], dim=1)
    logits = m.forward_text(seq, return_loss=False)
    assert logits.shape[:2] == seq.shape
    for b in range(seq.shape[0]):
        for i in range(prompt.shape[1] - 1, seq.shape[1] - 1):
            assert logits[b, i, seq[b, i + 1]].item() >= logits[b, i].max().item() - 0.05, (b, i)
