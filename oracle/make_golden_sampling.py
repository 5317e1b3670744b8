"""Golden vectors for the decode path: the UNMODIFIED reference's `sample_many` (and `sample_one`) on deterministic
weights / prompts / initial noise, greedy text (temperature 0).  Build container only.

    python -m oracle.make_golden_sampling

Two cases:
  * `sampling.pt`       dim128/depth4, max_length 12, 4 ODE grid points (fast; used by the sample_one == sample_many test)
  * `sampling_deep.pt`  dim256/depth8, max_length 64, 16 ODE grid points (the reference default), and for EVERY greedy decision the
                        reference's top-2 logit margin: the GPU test demands identity on every decisive step and accepts a
                        divergence only AT a recorded near-tie.

The margins are recorded without touching the reference source: `sample_text_token` (T:597-605) is wrapped to note the top-2 margin of
each row it decides, and a `_SamplingState` (T:1270-1287) subclass pairs every `last_token` assignment (T:2238, T:2331) with the row that
produced it - the k-th assignment after a call is row k of that call (T:2321-2331 walks the group in order).
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .ref_runner import build_reference_model, import_reference
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

CFG = dict(num_text_tokens=256, dim=128, depth=4, dim_latents=(32,), heads=2, dim_head=64)
CFG_DEEP = dict(num_text_tokens=256, dim=256, depth=8, dim_latents=(32,), heads=4, dim_head=64)
DEEP_KW = dict(max_length=64, text_temperature=0., modality_steps=16, fixed_modality_shape=(4,), cfg_scale=3.)
# the model of SURVEY 8(d) config 5 (dim 1024 / depth 24 / dim_latent 384: 885 M parameters): a short run of the same sampler, forced modality first
CFG_BIG = dict(num_text_tokens=256, dim=1024, depth=24, dim_latents=(384,), heads=8, dim_head=64)
# (round 3: 8 prompts - the four README prompt kinds twice -, max_length 64, 16 ODE grid points, as the timed workload uses them; round 2 pinned
#  2 prompts / max_length 20 / 4 grid points)
BIG_KW = dict(max_length=64, text_temperature=0., modality_steps=16, fixed_modality_shape=(4,), cfg_scale=3., force_modality_at_start=0)


def big_case():
    cfg = OracleConfig(**CFG_BIG)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='sampling_big')
    prompts = []
    for k in range(2):
        prompts += [D.det_randint(f'spb/p0/{k}', (16,), 0, 256), (0, D.det_normalish(f'spb/p1/{k}', (4, 384))), None,
                    [D.det_randint(f'spb/p3/{k}', (8,), 0, 256), (0, D.det_normalish(f'spb/p3m/{k}', (6, 384)))]]
    noise = D.det_normalish('spb/noise', (8, 384))
    return cfg, sd, prompts, noise


def run_big():
    cfg, sd, prompts, noise = big_case()
    model = build_reference_model(cfg, sd, modality_default_shape=(4,))
    model.eval()
    rec = MarginRecorder().install()
    try:
        outs = model.sample_many([p if not isinstance(p, list) else list(p) for p in prompts], init_modality_noise=noise, **BIG_KW)
        margins = rec.take()
    finally:
        rec.remove()
    g = dict(cfg=CFG_BIG, runs={'forced': [to_plain(o) for o in outs]}, margins={'forced': margins})
    for i, o in enumerate(outs):
        mg = margins[i]
        print('big', i, [('mod', tuple(p[1].shape)) if isinstance(p, tuple) else p.tolist() for p in o],
              f'| {len(mg)} decisions, min margin {min((m for _, _, m in mg), default=float("nan")):.4f}, {sum(m < 0.05 for _, _, m in mg)} below 0.05')
    path = os.path.join(OUT, 'sampling_big.pt')
    torch.save(g, path)
    print('saved', path, os.path.getsize(path), 'bytes')


def sampling_case(deep: bool = False, clean: bool = False):
    """`clean`: the same small case with `model_output_clean=True` (T:1297): the sampler converts the model's output to a flow in model
    space against the projected state (T:2446-2456)"""
    cfg = OracleConfig(**(CFG_DEEP if deep else CFG), **(dict(model_output_clean=True, eps=1e-2) if clean else {}))
    tag = 'sampling_deep' if deep else 'sampling'
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag=tag)
    pk = 'spd' if deep else 'sp'
    prompts = [
        D.det_randint(f'{pk}/p0', (12,), 0, 256),                                   # text prompt
        (0, D.det_normalish(f'{pk}/p1', (4, 32))),                                  # raw modality prompt
        None,                                                                      # empty prompt
        [D.det_randint(f'{pk}/p3', (6,), 0, 256), (0, D.det_normalish(f'{pk}/p3m', (3, 32)))],   # list prompt ending in a modality
    ]
    noise = D.det_normalish(f'{pk}/noise', (8, 32))
    return cfg, sd, prompts, noise


def to_plain(sample):
    out = []
    for p in sample:
        if isinstance(p, tuple):
            out.append(('mod', int(p[0]), p[1].detach().clone()))
        else:
            out.append(('text', p.detach().clone().long()))
    return out


class MarginRecorder:
    """records (part index, position in part, top-2 margin) of every greedily decided token, per sampling state, in creation order."""

    def __init__(self):
        self.pending = None
        self.states = []

    def install(self):
        import_reference()
        import transfusion_pytorch.transfusion as T
        rec = self
        orig = T.sample_text_token
        base = T._SamplingState

        def wrapped(logits, temperature=1.0, min_p=0.1):
            out = orig(logits, temperature, min_p)
            lg = logits.detach().float().reshape(-1, logits.shape[-1])
            top2 = lg.topk(2, dim=-1).values
            rec.pending = dict(margin=(top2[:, 0] - top2[:, 1]).tolist(), tok=out.reshape(-1).tolist(), k=0)
            return out

        class Recording(base):
            def __init__(s, *a, **k):
                object.__setattr__(s, '_rec', [])
                rec.states.append(s)
                super().__init__(*a, **k)

            def __setattr__(s, name, value):
                object.__setattr__(s, name, value)
                p = rec.pending
                if name == 'last_token' and value is not None and p is not None and p['k'] < len(p['tok']):
                    k = p['k']
                    if int(value.reshape(-1)[-1]) == p['tok'][k] and int(s.curr_seq[-1]) == p['tok'][k]:
                        s._rec.append((len(s.sample) - 1, int(s.curr_seq.numel()) - 1, float(p['margin'][k])))
                        p['k'] = k + 1

        T.sample_text_token = wrapped
        T._SamplingState = Recording
        self._undo = (T, orig, base)
        return self

    def remove(self):
        T, orig, base = self._undo
        T.sample_text_token = orig
        T._SamplingState = base

    def take(self):
        out = [list(s._rec) for s in self.states]
        self.states, self.pending = [], None
        return out


def run_case(deep: bool, clean: bool = False):
    cfg, sd, prompts, noise = sampling_case(deep, clean)
    model = build_reference_model(cfg, sd, modality_default_shape=(4,))
    model.eval()
    g = dict(cfg=CFG_DEEP if deep else CFG, runs={}, margins={}, model_output_clean=clean)
    rec = MarginRecorder().install()
    try:
        for name, kw in [('free', dict()), ('forced', dict(force_modality_at_start=0)), ('forced_nocfg', dict(force_modality_at_start=0, cfg_scale=1.))]:
            kwargs = dict(DEEP_KW, init_modality_noise=noise) if deep else \
                dict(max_length=12, text_temperature=0., init_modality_noise=noise, modality_steps=4, fixed_modality_shape=(4,), cfg_scale=3.)
            kwargs.update(kw)
            outs = model.sample_many([p if not isinstance(p, list) else list(p) for p in prompts], **kwargs)
            g['runs'][name] = [to_plain(o) for o in outs]
            g['margins'][name] = rec.take()
            for i, o in enumerate(outs):
                desc = [('mod', tuple(p[1].shape)) if isinstance(p, tuple) else p.tolist() for p in o]
                mg = g['margins'][name][i]
                print(name, i, desc, f'| {len(mg)} decisions, min margin {min((m for _, _, m in mg), default=float("nan")):.4f}, '
                      f'{sum(m < 0.05 for _, _, m in mg)} below 0.05')
    finally:
        rec.remove()
    path = os.path.join(OUT, 'sampling_clean.pt' if clean else ('sampling_deep.pt' if deep else 'sampling.pt'))
    torch.save(g, path)
    print('saved', path, os.path.getsize(path), 'bytes')


def main():
    import sys
    if 'big' in sys.argv[1:]:
        return run_big()
    if 'clean' in sys.argv[1:]:
        return run_case(False, clean=True)
    run_case(False)
    run_case(True)
    run_case(False, clean=True)


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    main()
