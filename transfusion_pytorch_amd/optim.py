"""Fused optimizer + data-parallel gradient exchange for the flat parameter buffer.

`train_toy.py:52-57` does loss.backward(); clip_grad_norm_(0.5); Adam.step(); zero_grad().  Here the whole
model is ONE flat fp32 buffer, so the step is: [one RCCL all-reduce over xGMI] -> one sum-of-squares
reduction -> one fused clip+Adam kernel.  No host sync: the clip coefficient is computed on the device.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import capi


class FusedAdam:
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0., max_grad_norm=None, process_group=None,
                 average_grads=True):
        self.model, self.lr, self.betas, self.eps, self.weight_decay = model, lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.group = process_group
        self.average = average_grads
        self.step_count = 0
        self.always_sync = False        # run the collective even at world size 1 (exercises the RCCL path on a 1-GPU box)
        self.m = self.v = self.sumsq = None

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def sync_grads(self):
        """the ONE collective of a data-parallel step: all-reduce(sum) of the flat gradient buffer (RCCL over xGMI)."""
        world = self._world()
        if world > 1 or (self.always_sync and dist.is_initialized()):
            dist.all_reduce(self.model.store.grad, op=dist.ReduceOp.SUM, group=self.group)
        return world

    def step(self):
        ps = self.model.store
        if ps.grad is None:
            raise capi.TfxError('FusedAdam needs the model on an MI355X (model.cuda())')
        world = self.sync_grads()
        if self.m is None or self.m.device != ps.flat.device or self.m.numel() != ps.numel:
            self.m = torch.zeros_like(ps.flat); self.v = torch.zeros_like(ps.flat)
            self.sumsq = torch.zeros(1, device=ps.flat.device)
        stream = torch.cuda.current_stream(ps.flat.device).cuda_stream
        self.step_count += 1
        max_norm = float(self.max_grad_norm) if self.max_grad_norm else 0.
        if max_norm > 0:
            self.sumsq.zero_()
            capi.check(capi.lib().tfx_sumsq(ps.grad.data_ptr(), ps.numel, self.sumsq.data_ptr(), stream), 'tfx_sumsq')
        a = capi.make_args('tfx_adam_args', p=ps.flat, g=ps.grad, m=self.m, v=self.v, n=ps.numel, lr=self.lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay, max_norm=max_norm,
                           grad_scale=(1.0 / world) if (self.average and world > 1) else 1.0, step=self.step_count, sumsq=self.sumsq)
        capi.call('tfx_adam_step', a, stream)
        # the master changed behind autograd's back: bump the version counters so the bf16 shadows are rebuilt
        ps._shadow_version = None

    def zero_grad(self, set_to_none: bool = True):
        ps = self.model.store
        if set_to_none:
            for p in ps.params.values():
                p.grad = None
        elif ps.grad is not None:
            ps.grad.zero_()
