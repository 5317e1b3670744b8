"""Where does the bf16 error of the native forward enter?  Per-layer relative error (Frobenius) of the hidden stream hid[l] - the output of layer
l - 1's feed-forward wrapper, the rows AttentionResidual mixes - against the fp32 oracle restatement on a golden case (default cfg3_1024:
dim 1024 / depth 24, one canonical sample), plus the final embedding / logits and the greedy agreement.
    python tools/layer_error.py [case]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.transfusion_oracle as O                      # noqa: E402
from oracle.cases import build_case                        # noqa: E402
from transfusion_pytorch_amd import Transfusion            # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3_1024'
cfg, sd, batch, times, noise = build_case(name)
dl = cfg.dim_latents if len(cfg.dim_latents) > 1 else cfg.dim_latents[0]
model = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=dl, transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads), prob_uncond=0.)
model.load_state_dict(sd, strict=True)
model = model.cuda().train()
model._noise_override = {t: v.cuda() for t, v in noise.items()}
loss = model(batch, times=times)
torch.cuda.synchronize()
plan = model._live[0]
b, n, nt, d = plan.b, plan.n, model._live_n_true, cfg.dim
cap = {}
orig = O.transformer_forward
def wrap(*a, **k):
    out, hid = orig(*a, **{**k, 'return_hiddens': True}); cap['hid'] = hid; return out
O.transformer_forward = wrap
torch.set_num_threads(min(32, os.cpu_count()))
with torch.no_grad():
    ref = O.forward_train(sd, cfg, batch, times, noise, return_all=True)
rel = lambda a, r: float((a.double() - r.double()).norm() / (r.double().norm() + 1e-30))
print(f'{name}: loss native {float(loss):.6f} oracle {float(ref["loss"]):.6f}')
for l, h in enumerate(cap['hid']):
    mine = plan.hid[l].view(b, n, d)[:, :nt].float().cpu()
    xr = plan.xres[l].view(b, n, d)[:, :nt].float().cpu()
    print(f'  hidden {l:2d}: rel err {rel(mine, h):.3e}   |h| {float(h.norm()):9.2f}' + (f'   (layer input xres[{l}] |x| {float(xr.norm()):9.2f})' if l else ''))
emb = plan.embed.view(b, n, d)[:, :nt].float().cpu(); lg = plan.logits.view(b, n, -1)[:, :nt, :cfg.vocab].float().cpu()
print(f'  final embedding rel err {rel(emb, ref["embed"]):.3e}   logits {rel(lg, ref["logits"]):.3e}   greedy agreement {(lg.argmax(-1) == ref["logits"].argmax(-1)).float().mean():.4f}')
