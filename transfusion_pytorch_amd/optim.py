"""Fused optimizer + data-parallel gradient exchange for the flat parameter buffer.

`train_toy.py:52-57` does loss.backward(); clip_grad_norm_(0.5); Adam.step(); zero_grad().  Here the whole
model is ONE flat fp32 buffer, so the step is: [one RCCL all-reduce over xGMI] -> one sum-of-squares
reduction -> one fused clip+Adam kernel.  No host sync: the clip coefficient is computed on the device.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist

from . import capi


class GradReducer:
    """Overlap of the data-parallel gradient exchange with the backward (reference practice: DDP's bucketed all-reduce under `accelerate`,
    train_mnist.py:114-126).  The flat gradient buffer completes BACK TO FRONT during the hand-written backward - the heads first, then the
    layers from the last to the first (their AdaLN conditioning weights included: engine.Plan forms those per layer), the embeddings last -
    so the buffer is exchanged in `groups` layer groups: as soon as a group's gradients are final (`Plan.bwd_cuts`), its two contiguous
    ranges (AdaLN weights, the layers' own parameters) go out as ONE asynchronous collective launch on the collective's stream while the
    backward of the earlier layers runs; what is left (AdaLN biases, time conditioning, the AttentionResidual parameters, embeddings, input /
    output projections) follows the backward as one more launch: groups + 1 collectives per step (5 by default).
    Every element is reduced exactly once (tests/test_dp_gloo.py).  One process per GPU, RCCL over xGMI via torch.distributed."""

    def __init__(self, model, process_group=None, groups: int = 4, exchange_dtype=None):
        self.model, self.group = model, process_group
        self.exchange_dtype = exchange_dtype         # torch.bfloat16: the ranges travel as bf16 (half the xGMI bytes), summed in bf16, written back to the fp32 buffer
        ps, md = model.store, model.md
        self.groups = max(1, min(groups, md.depth))
        self.per = -(-md.depth // self.groups)
        self.handles, self.done, self.staged = [], [], []
        self._step = -1
        d, D = md.dim, md.depth
        off = lambda name: ps.offsets[name][0]
        end = lambda name: ps.offsets[name][0] + int(torch.Size(ps.offsets[name][1]).numel())
        self._w0, self._b0 = off('transformer.layers.0.1.to_film.weight'), off('transformer.layers.0.1.to_film.bias')
        self._wl, self._bl = 2 * 3 * d * 4 * d, 2 * 3 * d                     # AdaLN weight / bias elements per layer (contiguous, layer-major)
        first = lambda i: off(f'transformer.layers.{i}.0.weight') if md.has_skip(i) else off(f'transformer.layers.{i}.1.fn.to_qk.0.weight')
        self._core = [first(i) for i in range(D)] + [off('transformer.norm.gamma')]
        self._head = (off('transformer.norm.gamma'), ps.numel)               # final norm, in / out projections, embeddings, logits
        self._mid = (end(f'transformer.layers.{D - 1}.2.to_ada_ln_zero.bias'), self._core[0])     # time conditioning MLP

    def ranges(self, lo: int, hi: int):
        """flat [start, end) ranges of the gradients of layers lo..hi (inclusive) that go out with the group: the layers' AdaLN conditioning
        weights and their own parameters.  (Their AdaLN biases - 6 d floats per layer, adjacent to the time-conditioning MLP - travel with the tail.)"""
        return [(self._w0 + lo * self._wl, self._w0 + (hi + 1) * self._wl), (self._core[lo], self._core[hi + 1])]

    def _reduce(self, ranges, async_op):
        """ONE collective launch for all `ranges`: fp32 in place through torch.distributed's coalescing manager (RCCL: one grouped launch, no
        staging copy; gloo: allreduce_coalesced), or - `exchange_dtype` - one all-reduce of a staged reduced-precision copy of the ranges"""
        g = self.model.store.grad
        ranges = [(a, b) for a, b in ranges if b > a]
        if not ranges:
            return
        if self.exchange_dtype is not None:
            buf = torch.cat([g[a:b].to(self.exchange_dtype) for a, b in ranges])
            h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self.handles.append(h); self.staged.append((ranges, buf))
            else:
                self._unstage(ranges, buf)
        elif len(ranges) == 1:
            a, b = ranges[0]
            h = dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self.handles.append(h)
        elif self._coalesce():
            with dist._coalescing_manager(group=self.group, async_ops=async_op) as cm:
                for a, b in ranges:
                    dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.group)
            if async_op:
                self.handles.append(cm)
        else:
            # public API only: one all-reduce per range (two collective launches per layer group instead of one).  THE DEFAULT until a real
            # multi-rank RCCL run has validated the private `_coalescing_manager` path (TFX_DP_COALESCE=1 selects it)
            for a, b in ranges:
                h = dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                if async_op:
                    self.handles.append(h)
            self.launches += len(ranges) - 1
        self.done += ranges
        self.launches += 1

    @staticmethod
    def _coalesce() -> bool:
        return os.environ.get('TFX_DP_COALESCE', '0') == '1' and hasattr(dist, '_coalescing_manager')

    def _unstage(self, ranges, buf):
        g, off = self.model.store.grad, 0
        for a, b in ranges:
            g[a:b].copy_(buf[off:off + b - a]); off += b - a

    launches = 0             # collective launches of the current step (<= groups + 1)

    def begin(self):
        self.handles, self.done, self.staged = [], [], []
        self.exchanged = False
        self.last_launches, self.launches = self.launches, 0

    last_launches = 0        # collective launches of the previous (finished) step

    exchanged = False        # a backward has sent layer groups out and no optimizer step has consumed them yet
    defer = False            # inside `FusedAdam.no_sync()`: this backward only accumulates (no cut, no collective) - a later one exchanges

    def group_ready(self, lo: int, hi: int):
        """called between two segments of the backward list: layers lo..hi are final (and, at the first cut, nothing else is)"""
        self._reduce(self.ranges(lo, hi), async_op=True)
        self.exchanged = True

    def check_fresh(self):
        """one backward per optimizer step (like DDP without `no_sync`): a second backward would accumulate into ranges that are already summed
        over the ranks (counting the first micro-batch `world` times) and race with the collectives in flight"""
        if self.exchanged:
            raise RuntimeError('overlap_grad_sync: a second backward() before optimizer.step() after the layer groups went out - run the micro-batches '
                               'that only accumulate under `with opt.no_sync():` (the LAST backward of the step, outside it, exchanges the sums)')

    def finish(self):
        """after the backward: exchange whatever no cut covered, then make the current stream wait for every collective"""
        n = self.model.store.numel
        covered = sorted(self.done)
        rest, pos = [], 0
        for a, b in covered:
            if a > pos:
                rest.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            rest.append((pos, n))
        self._reduce(rest, async_op=False)
        for h in self.handles:
            h.wait()
        self.handles = []
        for ranges, buf in self.staged:               # reduced-precision ranges come back into the fp32 buffer
            self._unstage(ranges, buf)
        self.staged = []
        assert sum(b - a for a, b in self.done) == n, 'every gradient element must be reduced exactly once'


class FusedAdam:
    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0., max_grad_norm=None, process_group=None,
                 average_grads=True):
        self.model, self.lr, self.betas, self.eps, self.weight_decay = model, lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.group = process_group
        self.average = average_grads
        self.step_count = 0
        self.always_sync = False        # run the collective even at world size 1 (exercises the RCCL path on a 1-GPU box)
        self.time_exchange = False      # bench.py: bracket the exchange section of every step with events on the compute stream (`exchange_ms`)
        self._xev = []
        self.m = self.v = self.sumsq = None
        self.reducer = None             # set by `overlap_grad_sync`: the exchange then runs in layer groups during the backward
        # parameters that live OUTSIDE the flat buffer: the positional-embedding MLPs and the user's pre / post transformer encoder-decoder
        # modules (PyTorch modules with autograd gradients).  They are few and small: a stock Adam steps them, under the SAME global clip
        # coefficient (their squared gradient norm is added to the flat buffer's before the fused kernel reads it)
        self.ext_params = list(model.external_parameters()) if hasattr(model, 'external_parameters') else []
        self.ext_opt = torch.optim.Adam(self.ext_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) if self.ext_params else None

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation (reference practice: `accelerator.accumulate(model)`, train_text_only.py:40, 117 = DDP's `no_sync()` around every
        micro-batch but the last): a backward inside this context only ADDS into the flat gradient buffer - with the overlapped exchange no layer
        group leaves (the buffer still changes), with the plain exchange nothing differs (its one all-reduce runs in `step()`).  The last
        micro-batch's backward, outside the context, sends the accumulated groups out as they become final."""
        red = self.reducer
        if red is None:
            yield
            return
        prev, red.defer = red.defer, True
        try:
            yield
        finally:
            red.defer = prev

    def overlap_grad_sync(self, groups: int = 4, exchange_dtype=None):
        """exchange the gradients in `groups` layer groups DURING the backward (GradReducer) instead of one all-reduce after it;
        `exchange_dtype=torch.bfloat16` halves the bytes on the links (the sum is formed in bf16; master gradients stay fp32)"""
        self.reducer = GradReducer(self.model, self.group, groups, exchange_dtype)
        self.model._grad_reducer = self.reducer
        self.model._dp_groups = self.reducer.groups
        return self

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def sync_grads(self):
        """the ONE collective of a data-parallel step: all-reduce(sum) of the flat gradient buffer (RCCL over xGMI)."""
        world = self._world()
        pending = self.reducer is not None and bool(self.reducer.done or self.reducer.exchanged)
        # (`pending`: the backward took the overlapped path - it does whenever torch.distributed is initialised, also at world size 1, e.g.
        #  `torchrun --nproc-per-node 1` - so its handles must be waited for and the reducer re-armed here whatever the world size)
        if world > 1 or pending or (self.always_sync and dist.is_initialized()):
            timed = self.time_exchange and self.model.store.grad is not None and self.model.store.grad.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if pending:
                self.reducer.finish()                  # the groups went out during the backward: tail + wait
                self.reducer.begin()
            else:
                dist.all_reduce(self.model.store.grad, op=dist.ReduceOp.SUM, group=self.group)
            if self.ext_params:
                # the external parameters: one more (small) collective over ALL of them, in declaration order, zeros where this rank's batch
                # produced no gradient (ragged multi-modal data: a rank without a modality type, or with text only, has `grad is None` for that
                # type's MLP / encoder) - every rank issues the same collective and, below, the same Adam step
                flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in self.ext_params])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                off = 0
                for p in self.ext_params:
                    g = flat[off:off + p.numel()].view_as(p).to(p.dtype); off += p.numel()
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
            if timed:
                e1.record(); self._xev.append((e0, e1))
        return world

    def exchange_ms(self):
        """mean time per step the compute stream spent in the exchange section of `step()` - the tail collective plus the wait for the groups still
        in flight, i.e. what the overlap did NOT hide (call after a device synchronize; `time_exchange` must be on)"""
        ms = [a.elapsed_time(b) for a, b in self._xev]
        self._xev = []
        return sum(ms) / len(ms) if ms else 0.0

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
        gscale = (1.0 / world) if (self.average and world > 1) else 1.0
        ext_grads = [p.grad for p in self.ext_params if p.grad is not None]
        if max_norm > 0:
            self.sumsq.zero_()
            capi.check(capi.lib().tfx_sumsq(ps.grad.data_ptr(), ps.numel, self.sumsq.data_ptr(), stream), 'tfx_sumsq')
            if ext_grads:
                self.sumsq += torch.stack([g.float().pow(2).sum() for g in ext_grads]).sum()
        if ext_grads:                                       # same scaling as adam_k: grad_scale * min(1, max_norm / (|g| grad_scale + 1e-6)), on the device
            coef = torch.full((), gscale, device=ps.flat.device)
            if max_norm > 0:
                coef = coef * (max_norm / (self.sumsq[0].sqrt() * gscale + 1e-6)).clamp(max=1.)
            for g in ext_grads:
                g.mul_(coef)
            for grp in self.ext_opt.param_groups:          # a schedule that sets `opt.lr = ...` (re-read by the fused kernel every step) reaches these too
                grp['lr'], grp['betas'], grp['eps'], grp['weight_decay'] = self.lr, tuple(self.betas), self.eps, self.weight_decay
            self.ext_opt.step()
        a = capi.make_args('tfx_adam_args', p=ps.flat, g=ps.grad, m=self.m, v=self.v, n=ps.numel, lr=self.lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, weight_decay=self.weight_decay, max_norm=max_norm,
                           grad_scale=gscale, step=self.step_count, sumsq=self.sumsq)
        capi.call('tfx_adam_step', a, stream)
        # the master changed behind autograd's back: new weights epoch (shadows rebuilt, kept decode plans dropped)
        ps.mark_dirty()

    def zero_grad(self, set_to_none: bool = True):
        ps = self.model.store
        if self.ext_opt is not None:
            self.ext_opt.zero_grad(set_to_none=set_to_none)
        if set_to_none:
            for p in ps.params.values():
                p.grad = None
        elif ps.grad is not None:
            ps.grad.zero_()
