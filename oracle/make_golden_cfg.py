"""Generate tests/golden/cfg1.pt from the UNMODIFIED reference: the DEFAULT training call `model(batch)` - no `times=`, CFG text
drop active (`prob_uncond` > 0 in `.training`) - i.e. the two pieces of `Transfusion.forward` every other golden bypasses:

  * `default_modality_length_to_time_fn` (T:186-200): k = floor(U * m_b); instances < k get t = 0.5, the rest all get the same U',
  * the classifier-free-guidance drop (T:3027-3043): with probability `prob_uncond` EVERY int tensor of a sample (sos / eos included)
    becomes `null_text_id`, and those labels are then ignored by the cross entropy (T:3322-3323).

Build container only:   python -m oracle.make_golden_cfg

TEST INFRASTRUCTURE ONLY.  The reference's three uniform draws - `torch.rand(b) < prob_uncond` (T:3030), `torch.rand_like(num_modalities.float())`
twice (T:193, T:197) - are replaced, in call order, by deterministic vectors stored in the fixture; the native test feeds the SAME
vectors to its own `torch.rand` calls and must reproduce the reference's `times` exactly and its loss / gradients to the usual tolerance.
"""
from __future__ import annotations

import os

import torch

from . import detdata as D
from .cases import CASES, default_shapes
from .ref_runner import build_reference_model, inject_noise
from .transfusion_oracle import OracleConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
PROB_UNCOND = 0.4


def cfg_case():
    kw, _, _ = CASES['small2']
    cfg = OracleConfig(**kw)
    b = 8
    batch = D.ragged_batch('cfg1/b', b, cfg.num_text_tokens, cfg.dim_latents)
    noise = D.det_noise('cfg1/n', batch, cfg.num_modalities)
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='cfg1')
    draws = dict(u_cfg=torch.tensor([0.1, 0.9, 0.3, 0.8, 0.7, 0.2, 0.95, 0.6]),   # rows with u_cfg < PROB_UNCOND lose their text: 0 and 2 (with modalities), 5 (text only)
                 u_k=D.det_uniform('cfg1/u_k', (b,), 0., 1.),          # -> k = floor(u_k * m_b)
                 u_t=D.det_uniform('cfg1/u_t', (b,), 0., 1.))          # -> the shared time of instances >= k
    return cfg, sd, batch, noise, draws


class patched_uniforms:
    """torch.rand / torch.rand_like -> the queued vectors, in call order (shape-checked)."""

    def __init__(self, queue):
        self.queue = list(queue)

    def __enter__(self):
        self._rand, self._rand_like = torch.rand, torch.rand_like

        def take(shape, device=None):
            v = self.queue.pop(0)
            assert tuple(v.shape) == tuple(shape), (tuple(v.shape), tuple(shape))
            return v.clone().to(device) if device is not None else v.clone()

        def rand(*size, device=None, **kw):
            shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            return take(shape, device)

        torch.rand = rand
        torch.rand_like = lambda t, **kw: take(t.shape, t.device)
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like = self._rand, self._rand_like
        return False


def make():
    cfg, sd, batch, noise, draws = cfg_case()
    model = build_reference_model(cfg, sd, default_shapes(cfg))
    model.prob_uncond = PROB_UNCOND
    model.train()
    with inject_noise(noise), patched_uniforms([draws['u_cfg'], draws['u_k'], draws['u_t']]) as pu:
        loss, bd, times = model(batch, return_breakdown=True, return_times=True)
        assert not pu.queue, 'the reference must have consumed exactly three uniform draws'
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    dropped = (draws['u_cfg'] < PROB_UNCOND)
    g = dict(reference='lucidrains/transfusion-pytorch v0.19.4 model(batch) in .train(), prob_uncond=0.4, default times fn, flat packing, fp32, CPU',
             prob_uncond=PROB_UNCOND, dropped_rows=dropped, times=times.detach().clone(),
             loss=loss.detach().double(), text_loss=bd.text.detach().double(), flow_losses=[f.detach().double() for f in bd.flow],
             grad_norms={k: float(v.double().norm()) for k, v in grads.items()},
             grad_head={k: v.reshape(-1)[:1024].clone() for k, v in grads.items()})
    path = os.path.join(OUT, 'cfg1.pt')
    torch.save(g, path)
    print(f'cfg1: dropped rows {dropped.nonzero().flatten().tolist()} times[0] {times[0].tolist()} loss {float(loss):.6f} text {float(bd.text):.6f} '
          f'flow {[float(f) for f in bd.flow]} ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    make()
