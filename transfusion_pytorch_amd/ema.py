"""Exponential moving average of a native Transfusion - what `Transfusion.create_ema` returns (T:1681-1699).

The reference builds on the un-vendored `ema_pytorch.EMA`; its public behaviour is restated here (parity UNPINNED: no
reference test pins the schedule): a frozen copy `ema_model`, `update()` every `update_every` steps - a plain copy until
`update_after_step`, then  ema = d * ema + (1 - d) * online  with the warm-up decay
d = clamp(1 - (1 + max(step - update_after_step - 1, 0) / inv_gamma) ** -power, min_value, beta) - `forward` and the listed
method names forwarded to the copy.  The update is ONE HIP launch over the flat fp32 parameter buffer (tfx_ema_update).
"""
from __future__ import annotations

import torch
from torch import nn

from . import capi


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3, min_value=0.0,
                 forward_method_names=()):
        super().__init__()
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.online_model = [model]                                  # a list keeps it out of the module tree (as ema_pytorch does)
        self.ema_model = model._clone_architecture()
        self.ema_model.requires_grad_(False)
        self.copy_params_from_model_to_ema()
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))
        for name in forward_method_names:
            setattr(self, name, getattr(self.ema_model, name))

    @property
    def model(self):
        return self.online_model[0]

    def copy_params_from_model_to_ema(self):
        with torch.no_grad():
            self.ema_model.store.flat.copy_(self.model.store.flat)
            self.ema_model.store.fourier_w.copy_(self.model.store.fourier_w)
            self.ema_model.store.rot_param.copy_(self.model.store.rot_param)
            self.ema_model.store.mark_dirty()
            for dst, src in self._external_pairs():
                dst.copy_(src)

    def _external_pairs(self):
        """(ema, online) pairs of the parameters that live outside the flat buffer (positional-embedding MLPs, user encoder / decoder modules)"""
        ext = lambda m: [p for mod in ([x for x in m.pos_emb_mlp if x is not None] +
                                       [x for t in sorted(m._ext) for x in (m.latent_to_model_projs[t], m.model_to_latent_projs[t])]) for p in mod.parameters()]
        return list(zip(ext(self.ema_model), ext(self.model)))

    def get_current_decay(self) -> float:
        epoch = max(int(self.step) - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    @torch.no_grad()
    def update(self):
        step = int(self.step)
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step or not bool(self.initted):
            self.copy_params_from_model_to_ema()
            if step > self.update_after_step:
                self.initted.fill_(True)
            return
        decay = self.get_current_decay()
        src, dst = self.model.store.flat, self.ema_model.store.flat
        capi.check(capi.lib().tfx_ema_update(dst.data_ptr(), src.data_ptr(), dst.numel(), float(decay), self.model._stream()), 'tfx_ema_update')
        self.ema_model.store.mark_dirty()
        for dst, src in self._external_pairs():                      # few, small: plain lerp
            dst.lerp_(src.to(dst.dtype), 1. - float(decay))

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
