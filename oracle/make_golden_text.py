"""Generate tests/golden/text*.pt from the UNMODIFIED reference's `forward_text` (build container only).

    python -m oracle.make_golden_text

TEST INFRASTRUCTURE ONLY.  Stores loss, logits (return_loss=False on the shifted input), final embed and every
parameter gradient of the pure-text path (T:2586-2664) on the deterministic inputs of oracle/cases.py.
"""
from __future__ import annotations

import os
import sys

import torch

from .cases import TEXT_CASES, build_text_case, default_shapes
from .ref_runner import build_reference_model

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def make(name: str):
    cfg, sd, text = build_text_case(name)
    model = build_reference_model(cfg, sd, default_shapes(cfg))
    model.train()
    captured = {}
    h = model.transformer.norm.register_forward_hook(lambda m, i, o: captured.__setitem__('embed', o.detach()))
    loss = model.forward_text(text)
    h.remove()
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad():
        logits = model.forward_text(text[:, :-1], return_loss=False)
    g = dict(case=name, reference='lucidrains/transfusion-pytorch v0.19.4 forward_text, fp32, CPU',
             input_checksum=float(text.double().abs().sum() + sum(float(v.double().abs().sum()) for v in sd.values())),
             loss=loss.detach().double(), logits=logits.detach().clone(), embed=captured['embed'].clone(),
             grad_norms={k: float(v.double().norm()) for k, v in grads.items()},
             grad_head={k: v.reshape(-1)[:1024].clone() for k, v in grads.items()})
    # greedy KV-cached generation (generate_text_only, T:2666-2707): tokens + the reference's top-2 margin at every step
    model.eval()
    prompt = text[:, :16].clone()
    with torch.no_grad():
        gen = model.generate_text_only(prompt, 16 + 24, temperature=0.)
        full = torch.cat((prompt, gen), dim=-1)
        lg = model.forward_text(full[:, :-1], return_loss=False)[:, 15:]                 # logits that produced each generated token
    top2 = lg.topk(2, dim=-1).values
    assert torch.equal(lg.argmax(-1), gen), 'cached generation must equal the teacher-forced argmax in the fp32 reference'
    g.update(gen_prompt=prompt, gen_tokens=gen.clone(), gen_margin=(top2[..., 0] - top2[..., 1]).clone())
    path = os.path.join(OUT, f'{name}.pt')
    torch.save(g, path)
    print(f'{name}: loss {float(g["loss"]):.6f}  {len(grads)} gradients  ({os.path.getsize(path) / 1e6:.2f} MB)')


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    for n in (sys.argv[1:] or list(TEXT_CASES)):
        make(n)
