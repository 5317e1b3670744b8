"""`Transfusion` - the reference's Python surface (transfusion_pytorch/__init__.py:1-6, constructor T:1292-1322,
forward T:2926-2948) in front of the native MI355X engine.

What is native: everything between the packed batch and the scalar loss, forward AND backward (engine.py).
What stays Python: the structure scan of the ragged input (pure bookkeeping, MP:206-263) and API glue.
Unsupported reference options raise NotImplementedError - there is no silent PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os
from typing import NamedTuple

import numpy as np
import torch
from torch import nn

from . import capi
from .engine import Plan
from .packing import fast_signature, is_int_tensor, scan_batch, scan_signature, token_maps, token_segments
from .axial import ContinuousAxialPositionalEmbedding
from .params import ModelDims, ParamStore


class LossBreakdown(NamedTuple):            # T:105-110
    total: torch.Tensor
    text: torch.Tensor
    flow: list
    velocity: list | None = None
    recon: list | None = None


from .modality_processing import PROCESSING_STRATEGIES, ProcessedModalityBatch, process_modalities      # noqa: E402  the registry seam (MP:1050-1058)


def exists(v):                                            # T:130-131
    return v is not None


def random_modality_length_to_time_fn(num_modalities):    # T:181-184: an independent uniform time per modality slot
    return torch.rand((num_modalities.shape[0], int(num_modalities.amax()) if num_modalities.numel() else 0), device=num_modalities.device)


def default_modality_length_to_time_fn(num_modalities):   # T:186-200 (the model's default draws the same numbers: Transfusion._default_times)
    b, dev = num_modalities.shape[0], num_modalities.device
    m = int(num_modalities.amax()) if num_modalities.numel() else 0
    if m == 0:
        return torch.empty((b, 0), device=dev, dtype=torch.float)
    decoded = torch.floor(torch.rand(b, device=dev) * num_modalities.float())
    cur = torch.rand(b, device=dev)
    return torch.where(torch.arange(m, device=dev)[None, :] < decoded[:, None], torch.full((), 0.5, device=dev), cur[:, None].expand(b, m))


def char_tokenize(text: str, offset: int = 0):            # T:242-249: one token per character of a shape string
    return torch.tensor([ord(c) + offset for c in text], dtype=torch.long)


def stack_same_shape_tensors_with_inverse(tensors):       # T:476-519
    """group tensors by shape, stack each group; `inverse(dict shape -> batched result)` puts the rows back in the original order"""
    slots, groups = [], {}
    for x in tensors:
        key = tuple(x.shape)
        slots.append((key, len(groups.setdefault(key, []))))
        groups[key].append(x)
    counts = {k: len(v) for k, v in groups.items()}

    def inverse(batched):
        assert {k: len(v) for k, v in batched.items()} == counts
        return [batched[k][i] for k, i in slots]
    return {k: torch.stack(v) for k, v in groups.items()}, inverse


def filter_with_inverse(cond, inp):                       # T:521-548
    """elements of `inp` satisfying `cond`; `inverse(new elements)` returns `inp` with those positions replaced"""
    picked = [i for i, el in enumerate(inp) if cond(el)]

    def inverse(new):
        assert len(new) == len(picked)
        out = list(inp)
        for i, el in zip(picked, new):
            out[i] = el
        return out
    return [inp[i] for i in picked], inverse


def apply_fn_modality_type(fn, modalities, modality_type=0, return_untransformed=False):      # T:550-589
    """run `fn` on every modality tensor of one type inside a sample or a list of samples, same-shaped tensors stacked into one call; bare
    float tensors count as type 0.  Structure (lists) is preserved; transformed parts come back as (type, tensor[, original])."""
    single = not (len(modalities) > 0 and isinstance(modalities[0], list))
    batch = [modalities] if single else modalities
    flat = [(0, p) if (torch.is_tensor(p) and p.is_floating_point()) else p for sample in batch for p in sample]
    chosen, put_back = filter_with_inverse(lambda el: isinstance(el, tuple) and el[0] == modality_type, flat)
    tensors = [x for _, x in chosen]
    stacked, unstack = stack_same_shape_tensors_with_inverse(tensors)
    res = unstack({k: fn(v) for k, v in stacked.items()}) if tensors else []
    new = [(modality_type, r, x) if return_untransformed else (modality_type, r) for r, x in zip(res, tensors)]
    flat = put_back(new)
    out, i = [], 0
    for sample in batch:
        out.append(flat[i:i + len(sample)]); i += len(sample)
    return out[0] if single else out


def default_to_modality_shape_fn(maybe_shape_str):       # T:176-177
    return tuple([*map(int, maybe_shape_str.split(','))])


def cast_tuple(t, length=1):
    return t if isinstance(t, tuple) else ((t,) * length)


def _bucket(x: int, min_step: int) -> int:
    """round a ragged count up to a bucket: steps of a quarter of its power of two (1950 instances -> 2048, 7700 rows -> 8192), never below
    `min_step` - a ragged corpus then lands on a handful of training plans instead of one per (64, 256) cell (a plan costs ~0.5 s to build and
    owns every activation of the step; the plan cache holds 8).  Padding instances / rows are referenced by nothing and cost a sliver of the
    small per-instance and per-row GEMMs."""
    if x <= 0:
        return x
    step = max(min_step, (1 << (x.bit_length() - 1)) // 4)
    return -(-x // step) * step


_TRAIN_PAD = 64


class ModalityInfo(NamedTuple):              # T:112-126
    encoder: object
    decoder: object
    latent_to_model: object
    model_to_latent: object
    add_pos_emb: bool
    pos_emb_mlp: object
    num_dim: object
    dim_latent: int
    default_shape: object
    som_id: int
    eom_id: int
    to_shape_fn: object
    channel_first_latent: bool
    modality_type: int


class KVCacheView(torch.Tensor):
    """the kv-cache tensor `forward(..., return_kv_cache=True)` hands out: reference layout (layers, key/value, batch, heads, seq, dim_head),
    bf16, a view of the native cache buffer it remembers (so that passing it back as `cache=` appends in place instead of copying)."""


class Transformer:
    """config holder with the reference's constructor signature (T:1043-1059); the layers live in `Transfusion`."""

    def __init__(self, dim, *, depth, dim_head=64, heads=8, dropout=0., ff_expansion_factor=4, attn_kwargs: dict = dict(),
                 ff_kwargs: dict = dict(), attn_laser=False, unet_skips=True, use_flex_attn=False, qk_rmsnorm=True, use_value_residual=False):
        # use_flex_attn only selects the reference's attention BACKEND (same scores, same mask, T:998-1027): accepted, the native kernel runs
        unsupported = dict(dropout=dropout != 0., attn_kwargs=bool(attn_kwargs), ff_kwargs=bool(ff_kwargs), attn_laser=attn_laser,
                           unet_skips=not unet_skips, qk_rmsnorm=not qk_rmsnorm, use_value_residual=use_value_residual)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f'Transformer options not supported by the native MI355X path: {bad}')
        if dim_head > 64 or dim_head % 2 or dim_head < 2:
            raise NotImplementedError('the native attention kernels hold 64 columns per head: dim_head must be even and <= 64 '
                                      '(smaller heads run zero-padded)')
        if dim % 64 != 0:
            raise NotImplementedError('the native kernels need dim to be a multiple of 64')
        self.dim, self.depth, self.dim_head, self.heads, self.ff_expansion_factor = dim, depth, dim_head, heads, ff_expansion_factor


class _MoveDim(nn.Module):
    """the Rearrange('b d ... -> b ... d') / Rearrange('b ... d -> b d ...') the reference wraps around channel-first encoders / decoders (T:1487-1491)"""

    def __init__(self, src, dst):
        super().__init__()
        self.src, self.dst = src, dst

    def forward(self, x):
        return x.movedim(self.src, self.dst)


class _NativeLoss(torch.autograd.Function):
    """connects the engine's hand-written backward to autograd.

    Beyond the scalar loss it carries the rows that cross the PyTorch / engine line for modality types whose maps live in PyTorch:
      inputs  `rows` - token rows PyTorch produced for the engine, one per entry of spec['in'] = [('tok' | 'add', type)]: the user's encoder
                       output (`pre_post_transformer_enc_dec`) or the axial positional embedding rows; their gradient is d loss / d x0 at those rows
      outputs        - the final-embedding rows of spec['out'] types, for the user's decoder; their incoming gradient seeds the backward next to the loss"""

    @staticmethod
    def forward(ctx, anchor, model, loss, spec, *rows):
        ctx.model, ctx.step_id, ctx.spec = model, model._step_id, spec
        plan = model._live[0]
        outs = [loss.clone()]
        for t in spec['out']:
            outs.append(plan.embed.index_select(0, plan.row_tok[t].long().clamp(min=0)).float())
        return tuple(outs)

    @staticmethod
    def backward(ctx, grad_loss, *grad_rows):
        model, spec = ctx.model, ctx.spec
        plan = model._live[0]
        for t, g in zip(spec['out'], grad_rows):
            plan.lat[t]['gemb'].copy_(g)
        model._native_backward(grad_loss, ctx.step_id)
        back = [plan.dx0.index_select(0, plan.row_tok[t].long().clamp(min=0)).float() for _, t in spec['in']]
        return (None, None, None, None, *back)


_NO_ROWS = {'in': [], 'out': []}


class Transfusion(nn.Module):
    def __init__(
        self, *, num_text_tokens, transformer, model_output_clean=False, dim_latent=None, channel_first_latent=False, add_pos_emb=False,
        modality_encoder=None, modality_decoder=None, pre_post_transformer_enc_dec=None, modality_default_shape=None,
        fallback_to_default_shape_if_invalid=False, modality_num_dim=None, to_modality_shape_fn=default_to_modality_shape_fn,
        ignore_index=-1, flow_loss_weight=1., text_loss_weight=1., velocity_consistency_loss_weight=0.1, reconstruction_loss_weight=0.,
        modality_encoder_decoder_requires_batch_dim=True, odeint_kwargs: dict = dict(atol=1e-5, rtol=1e-5, method='midpoint'),
        eps=1e-2, prob_uncond=0.1, modality_processing: str = 'auto',
    ):
        super().__init__()
        self._init_kwargs = dict(num_text_tokens=num_text_tokens, transformer=transformer, model_output_clean=model_output_clean, dim_latent=dim_latent,
                                 channel_first_latent=channel_first_latent, add_pos_emb=add_pos_emb, modality_default_shape=modality_default_shape,
                                 pre_post_transformer_enc_dec=pre_post_transformer_enc_dec,
                                 modality_encoder=modality_encoder, modality_decoder=modality_decoder,
                                 modality_encoder_decoder_requires_batch_dim=modality_encoder_decoder_requires_batch_dim,
                                 fallback_to_default_shape_if_invalid=fallback_to_default_shape_if_invalid, modality_num_dim=modality_num_dim,
                                 to_modality_shape_fn=to_modality_shape_fn, ignore_index=ignore_index, flow_loss_weight=flow_loss_weight,
                                 text_loss_weight=text_loss_weight, velocity_consistency_loss_weight=velocity_consistency_loss_weight,
                                 reconstruction_loss_weight=reconstruction_loss_weight, odeint_kwargs=odeint_kwargs, eps=eps, prob_uncond=prob_uncond, modality_processing=modality_processing)
        assert modality_processing in PROCESSING_STRATEGIES, \
            f'unknown modality processing strategy `{modality_processing}`, available: {list(PROCESSING_STRATEGIES)}'      # MP:1254-1256
        self.modality_processing = modality_processing
        self.reconstruction_loss_weight = float(reconstruction_loss_weight)                 # T:1522-1525
        self.has_recon_loss = self.reconstruction_loss_weight > 0.
        if odeint_kwargs.get('method', 'midpoint') != 'midpoint':
            raise NotImplementedError('only the fixed-grid midpoint solver is implemented')
        if isinstance(transformer, dict):
            transformer = Transformer(**transformer)
        self.transformer_config = transformer
        self.dim = dim = transformer.dim
        dim_latent = dim if dim_latent is None else dim_latent
        self.dim_latents = cast_tuple(dim_latent)
        self.num_modalities = len(self.dim_latents)
        if modality_default_shape is None or (isinstance(modality_default_shape, tuple) and all(isinstance(i, int) for i in modality_default_shape)):
            modality_default_shape = (modality_default_shape,) * self.num_modalities           # T:1362-1363
        self.modality_default_shape = modality_default_shape
        assert len(self.modality_default_shape) == self.num_modalities
        # channel-first latents (T:1352, T:1481-1489) and frozen modality encoders / decoders (T:1405-1418): host-side layout / pre- and
        # post-processing around the native path - the kernels always see (*axial, dim_latent) rows
        self.channel_first_latent = cast_tuple(channel_first_latent, self.num_modalities)
        assert len(self.channel_first_latent) == self.num_modalities
        enc = cast_tuple(modality_encoder, 1 if modality_encoder is not None else self.num_modalities)
        dec = cast_tuple(modality_decoder, 1 if modality_decoder is not None else self.num_modalities)
        assert len(enc) == self.num_modalities and len(dec) == self.num_modalities
        self.modality_encoder, self.modality_decoder = nn.ModuleList(enc), nn.ModuleList(dec)
        self._enc_dec_batch_dim = bool(modality_encoder_decoder_requires_batch_dim)
        self.fallback_to_default_shape_if_invalid = fallback_to_default_shape_if_invalid
        if modality_num_dim is None:
            modality_num_dim = tuple(len(s) if s is not None else None for s in self.modality_default_shape)
        self.modality_num_dim = cast_tuple(modality_num_dim, self.num_modalities)
        self.to_modality_shape_fn = cast_tuple(to_modality_shape_fn, self.num_modalities)
        # vocabulary layout  T:1420-1449
        self.num_text_tokens = num_text_tokens
        self.sos_id, self.eos_id, self.null_text_id = num_text_tokens, num_text_tokens + 1, num_text_tokens + 2
        M = self.num_modalities
        self.som_ids = [num_text_tokens + 3 + i for i in range(M)]
        self.eom_ids = [num_text_tokens + 3 + M + i for i in range(M)]
        self.meta_id = num_text_tokens + 3 + 2 * M
        self.ignore_index = ignore_index
        self.flow_loss_weight, self.text_loss_weight = flow_loss_weight, text_loss_weight
        self.velocity_consistency_loss_weight = velocity_consistency_loss_weight
        self.eps, self.prob_uncond = eps, prob_uncond
        self.model_output_clean = bool(model_output_clean)
        self.odeint_kwargs = dict(odeint_kwargs)

        # axial positional embedding per modality type (T:1384-1403): a host MLP produces the rows, the engine adds them into the token stream
        self.add_pos_emb = cast_tuple(add_pos_emb, self.num_modalities)
        assert len(self.add_pos_emb) == self.num_modalities
        self.pos_emb_mlp = nn.ModuleList([])
        for flag, ndim in zip(self.add_pos_emb, self.modality_num_dim):
            if not flag:
                self.pos_emb_mlp.append(None)
                continue
            assert ndim is not None, '`modality_num_dim` must be set if you wish to automatically inject axial positional embeddings'     # T:1396
            self.pos_emb_mlp.append(ContinuousAxialPositionalEmbedding(dim=dim, num_axial_dims=ndim))
        # learnable encoder / decoder pairs around the transformer (`pre_post_transformer_enc_dec`, the paper's U-Net down / up, T:1451-1494):
        # they REPLACE latent_to_model / model_to_latent of their type and run in PyTorch; the engine takes their token rows and returns
        # the embedding rows (and the gradients both ways)
        ppe = pre_post_transformer_enc_dec
        if isinstance(ppe, tuple) and len(ppe) == 2 and all(isinstance(m, nn.Module) for m in ppe):
            ppe = (ppe,)                                                                     # one (encoder, decoder) pair, T:1453-1454
        ppe = cast_tuple(ppe, self.num_modalities)
        assert len(ppe) == self.num_modalities
        ext_modules = {}
        for t, pair in enumerate(ppe):
            if pair is None:
                continue
            pre, post = pair
            assert pre is not None and post is not None, 'pre_post_transformer_enc_dec takes (encoder, decoder) pairs'
            if self.channel_first_latent[t]:                                                # T:1487-1491
                pre, post = nn.Sequential(pre, _MoveDim(1, -1)), nn.Sequential(_MoveDim(-1, 1), post)
            ext_modules[t] = (pre, post)

        self.md = ModelDims(num_text_tokens=num_text_tokens, dim=dim, depth=transformer.depth, heads=transformer.heads,
                            dim_head=transformer.dim_head, dim_latents=tuple(self.dim_latents), ff_expansion_factor=transformer.ff_expansion_factor, model_output_clean=bool(model_output_clean), clean_eps=float(eps),
                            pos_types=tuple(t for t, f in enumerate(self.add_pos_emb) if f), ext_types=tuple(sorted(ext_modules)))
        self.store = ParamStore(self.md, self)
        if ext_modules:       # the reference's attribute names: latent_to_model_projs[t] / model_to_latent_projs[t] ARE the user's modules (T:1493-1494)
            for name in ('latent_to_model_projs', 'model_to_latent_projs'):
                if not hasattr(self, name):
                    setattr(self, name, nn.ModuleList())
                ml = getattr(self, name)
                while len(ml) < self.num_modalities:
                    ml.append(None)
            for t, (pre, post) in ext_modules.items():
                self.latent_to_model_projs[t], self.model_to_latent_projs[t] = pre, post
        self._ext = set(ext_modules)
        # state_dict names of a channel-first type's projections: the reference wraps them in nn.Sequential(Rearrange, Linear) / (Linear,
        # Rearrange) (T:1481-1483), i.e. `latent_to_model_projs.t.1.*` and `model_to_latent_projs.t.0.*`: translate on load and on save
        self._key_renames = {}
        for t, cf in enumerate(self.channel_first_latent):
            if cf and t not in self._ext:
                self._key_renames[f'latent_to_model_projs.{t}.weight'] = f'latent_to_model_projs.{t}.1.weight'
                self._key_renames[f'latent_to_model_projs.{t}.bias'] = f'latent_to_model_projs.{t}.1.bias'
                self._key_renames[f'model_to_latent_projs.{t}.weight'] = f'model_to_latent_projs.{t}.0.weight'
        if self._key_renames:
            def _pre(state_dict, prefix, *a):
                for ours, ref in self._key_renames.items():
                    if prefix + ref in state_dict:
                        state_dict[prefix + ours] = state_dict.pop(prefix + ref)
            def _post(module, state_dict, prefix, local_metadata):
                for ours, ref in self._key_renames.items():
                    if prefix + ours in state_dict:
                        state_dict[prefix + ref] = state_dict.pop(prefix + ours)
                return state_dict
            self._register_load_state_dict_pre_hook(_pre)
            self._register_state_dict_hook(_post)
        self._plans = {}
        self._struct_cache = {}
        self._step_id = 0
        self._live = None
        self._consumed = -1
        self._bwd_scale = None
        self._anchor = None
        self._rope = None
        self._gen_noise_override = None      # test hook: initial noise of generate_modality_only
        self._noise_override = None          # test hook: type -> (R, dl) noise (parity runs inject the oracle's noise)

    # ------------------------------------------------------------------ nn.Module plumbing
    @property
    def device(self):
        return self.store.flat.device

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn)
        dev = next(iter(self.store.params.values())).device
        self.store.fourier_w = self.transformer.to_time_cond[0].weights
        self.store.reflatten(dev)
        self._plans.clear()
        self._struct_cache.clear()
        self._rope = None
        return out

    def vocab_size(self):
        return self.md.vocab

    # ------------------------------------------------------------------ encoders / decoders / channel-first layout
    def parameters_without_encoder_decoder(self):                     # T:1650-1655
        return set(self.parameters()) - set(self.modality_encoder.parameters()) - set(self.modality_decoder.parameters())

    def get_modality_info(self, modality_type=None):                  # T:1547-1590
        """per-type record with the reference's field names.  `latent_to_model` / `model_to_latent` are the user's modules for
        `pre_post_transformer_enc_dec` types and None otherwise (the native projections live in the flat parameter buffer: `model_to_latent()`)"""
        t = 0 if modality_type is None else int(modality_type)
        ext = t in self._ext
        return ModalityInfo(encoder=self.modality_encoder[t], decoder=self.modality_decoder[t],
                            latent_to_model=self.latent_to_model_projs[t] if ext else None, model_to_latent=self.model_to_latent_projs[t] if ext else None,
                            add_pos_emb=bool(self.add_pos_emb[t]), pos_emb_mlp=self.pos_emb_mlp[t], num_dim=self.modality_num_dim[t], dim_latent=self.dim_latents[t],
                            default_shape=self.modality_default_shape[t], som_id=self.som_ids[t], eom_id=self.eom_ids[t], to_shape_fn=self.to_modality_shape_fn[t],
                            channel_first_latent=bool(self.channel_first_latent[t]), modality_type=t)

    def get_all_modality_info(self):                                   # T:1591-1592
        return [self.get_modality_info(i) for i in range(self.num_modalities)]

    def char_tokenizer(self, text: str):                               # T:1446: characters of a shape string -> token ids behind [meta]
        return char_tokenize(text, offset=self.meta_id + 1)

    def __deepcopy__(self, memo):
        """`copy.deepcopy(model)` (the reference's tests build their EMA teachers that way): a fresh model of the same architecture with the
        parameters copied - plans, launch lists and device index caches hold raw pointers and are rebuilt by the copy on first use"""
        import copy
        kw = dict(self._init_kwargs)
        kw['transformer'] = self.transformer_config
        for k in ('pre_post_transformer_enc_dec', 'modality_encoder', 'modality_decoder'):
            kw[k] = copy.deepcopy(kw.get(k), memo)
        new = Transfusion(**kw)
        if self.device.type == 'cuda':
            new = new.to(self.device)
        new.load_state_dict(self.state_dict())
        new.train(self.training)
        memo[id(self)] = new
        return new

    def external_parameters(self):
        """learnable parameters outside the flat native buffer: the axial positional-embedding MLPs and the user's pre / post transformer
        encoder-decoder modules (optim.FusedAdam steps them with a stock Adam under the same global clip)"""
        mods = [m for m in self.pos_emb_mlp if m is not None]
        mods += [m for t in sorted(self._ext) for m in (self.latent_to_model_projs[t], self.model_to_latent_projs[t])]
        return [p for m in mods for p in m.parameters() if p.requires_grad]

    def _apply_fn_modality_type(self, fn, samples, modality_type):
        """apply_fn_modality_type (T:542-582): run `fn` on every modality of one type in a list of samples, same-shaped tensors stacked"""
        single = bool(samples) and not isinstance(samples[0], list)
        batch = [samples] if single else samples
        groups = {}
        for si, sample in enumerate(batch):
            for pi, part in enumerate(sample):
                if torch.is_tensor(part) and part.is_floating_point():
                    part = (0, part)
                if isinstance(part, tuple) and part[0] == modality_type:
                    groups.setdefault(tuple(part[1].shape), []).append((si, pi, part[1]))
        out = [list(sample) for sample in batch]
        for shape, items in groups.items():
            res = fn(torch.stack([x for _, _, x in items]))
            for (si, pi, _), r in zip(items, res):
                out[si][pi] = (modality_type, r)
        return out[0] if single else out

    def _encode_modalities(self, samples):
        """frozen modality encoders (T:3094-3101), no gradients"""
        for t, enc in enumerate(self.modality_encoder):
            if enc is not None:
                with torch.no_grad():
                    enc.eval()
                    samples = self._apply_fn_modality_type(enc, samples, t)
        return samples

    @torch.no_grad()
    def decode_modalities(self, samples):                              # T:1826-1840
        for t, dec in enumerate(self.modality_decoder):
            if dec is not None:
                dec.eval()
                samples = self._apply_fn_modality_type(dec, samples, t)
        return samples

    def _to_channel_last(self, samples):
        """(dim_latent, *axial) -> (*axial, dim_latent) for the channel-first modality types (T:1481-1489); no copy for the others"""
        if not any(self.channel_first_latent):
            return samples
        out = []
        for sample in samples:
            s = []
            for part in sample:
                if torch.is_tensor(part) and part.is_floating_point():
                    part = (0, part)
                if isinstance(part, tuple) and self.channel_first_latent[part[0]] and part[0] not in self._ext:
                    part = (part[0], part[1].movedim(0, -1))          # (`ext` types stay in the layout their encoder takes)
                s.append(part)
            out.append(s)
        return out

    def _from_channel_last(self, ty, x):
        return x.movedim(-1, 0) if self.channel_first_latent[ty] else x

    # ------------------------------------------------------------------ rows PyTorch computes for the engine: positional embedding, user encoders
    def _pos_rows(self, t, shapes):
        """axial positional embedding rows (sum of lengths, dim) of modality type t for instances of the given (projected) axial shapes, in scan
        order (T:2795-2796; MP:1003-1045 evaluates the same per-axis MLPs once at the maximum extents and slices - same values)"""
        mlp, cache, out = self.pos_emb_mlp[t], {}, []
        for shp in shapes:
            shp = tuple(int(a) for a in shp)
            assert len(shp) == mlp.num_axial_dims, f'received modalities of ndim {len(shp)} but expected {mlp.num_axial_dims}'     # T:2786
            if shp not in cache:
                cache[shp] = mlp(shp, flatten=True)
            out.append(cache[shp])
        return out[0] if len(out) == 1 else torch.cat(out)

    def _ext_preprocess(self, modalities, times, return_loss):
        """`pre_post_transformer_enc_dec` types in the interleaved forward: noising and the user's encoder run in PyTorch, one instance at a
        time in scan order (process_instance, MP:715-745; a conv / U-Net encoder mixes tokens, so instances are never concatenated, MP:626-632).
        Returns the batch with those parts replaced by shape-only placeholders of the PROJECTED axial shape - positions, the meta shape string and
        the token count use that one (MP:738-741) - and per type the token rows / flow targets / shapes in scan order."""
        dev, d = self.device, self.md.dim
        ctx = {t: dict(tok=[], flow=[], shape=[], eps=[], noised=[], time=[]) for t in self._ext}
        out = []
        for bi, sample in enumerate(modalities):
            m, row = 0, []
            for part in sample:
                if torch.is_tensor(part) and part.is_floating_point():
                    part = (0, part)
                if isinstance(part, tuple):
                    ty, x = int(part[0]), part[1]
                    if ty in self._ext:
                        c = ctx[ty]
                        x = x.to(dev, torch.float32)
                        if times is not None:
                            c['time'].append(times[bi, m])
                        if return_loss:
                            tt = times[bi, m]
                            ov = self._noise_override
                            eps = ov[ty][len(c['tok'])].to(dev, torch.float32) if ov is not None else torch.randn_like(x)
                            noised, flow = x * tt + eps * (1. - tt), x - eps                # MP:717-719
                            c['flow'].append(flow); c['eps'].append(eps); c['noised'].append(noised)
                        else:
                            noised = x
                        pre = self.latent_to_model_projs[ty]
                        tok = pre(noised[None])[0] if self.channel_first_latent[ty] else pre(noised)       # MP:729-732
                        assert tok.shape[-1] == d, f'the encoder of modality {ty} must produce model-dimension ({d}) tokens'
                        c['tok'].append(tok.reshape(-1, d)); c['shape'].append(tuple(tok.shape[:-1]))
                        part = (ty, torch.empty((*tok.shape[:-1], self.dim_latents[ty]), device='meta'))
                    m += 1
                row.append(part)
            out.append(row)
        return out, ctx

    def _ext_decode(self, t, rows, ctx):
        """the user's decoder on each instance's embedding rows (add_temp_batch_dim(model_to_latent), T:3300-3301), one instance at a time in scan
        order.  `model_output_clean`: the rows go through the model-space conversion first - (embed - projected tokens) / max(1 - t, eps), the
        decorator build_record_closures puts around every closure (MP:786-792, MP:99-126); the subtrahend is the encoder's output WITH its
        autograd history, as there.  Returns the per-instance predictions in the decoder's layout."""
        post, d = self.model_to_latent_projs[t], self.md.dim
        preds, lo = [], 0
        for j, shape in enumerate(ctx['shape']):
            L = int(np.prod(shape))
            e = rows[lo:lo + L]
            if self.model_output_clean:
                e = (e - ctx['tok'][j]) / (1. - ctx['time'][j]).clamp_min(self.md.clean_eps)
            preds.append(post(e.reshape(1, *shape, d))[0])
            lo += L
        return preds

    def _ext_flow_loss(self, t, preds, ctx):
        """flow loss of an `ext` type: ONE mse over all instances of the type packed together (T:3356-3364); with a reconstruction loss also the
        mean over the instances of mse(noised, noise + pred (1 - t)) (MP:177-200, T:3422-3426).  Returns (flow loss, reconstruction loss | None)."""
        fl = torch.nn.functional.mse_loss(torch.cat([p.reshape(-1) for p in preds]), torch.cat([f.reshape(-1) for f in ctx['flow']]))
        rec = None
        if self.has_recon_loss:
            rec = sum(torch.nn.functional.mse_loss(nz, eps + p * (1. - tt)) for p, nz, eps, tt in zip(preds, ctx['noised'], ctx['eps'], ctx['time'])) / len(preds)
        return fl, rec

    # ------------------------------------------------------------------ helpers
    def mark_weights_changed(self):
        """tell the model its weights were edited where nothing can see it - in place through `.data` (`p.data.mul_(0.5)`) or by a kernel of the
        caller's: the bf16 shadows are rebuilt on the next call.  Not needed after optimizers, `load_state_dict`, `p.copy_` / `p.add_` under
        `no_grad` (version counters) or `p.data = w` (re-adopted into the flat buffer, `ParamStore.params_version`)."""
        self.store.mark_dirty()

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def release_decode_cache(self):
        """drop the KV cache + decode plans `sample_many` keeps between calls (sampling.py; up to TFX_DECODE_KEEP_GB).  `train()` does it too; an EMA / eval model
        sampled now and then inside a training process never switches modes - call this after sampling to get the memory back."""
        self._decode_keep = None

    def train(self, mode: bool = True):
        """nn.Module.train; switching INTO training drops the decode cache + plans `sample_many` keeps between calls (sampling.py: up to
        TFX_DECODE_KEEP_GB of KV cache stays allocated for a serving process; a training process gets the memory back)"""
        if mode:
            self._decode_keep = None
        return super().train(mode)

    def _require_gpu(self):
        if self.device.type != 'cuda':
            raise capi.TfxError('the Transfusion hot path only runs on an MI355X (model.cuda()); there is no CPU fallback')
        capi.lib()

    def _rope_tables(self, max_pos: int):
        if self._rope is None or self._rope[0].shape[0] <= max_pos:
            P = max(2048, 1 << (max_pos + 1).bit_length())
            freqs = self.store.rot_param.detach().float().cpu()
            ang = torch.arange(P, dtype=torch.float32)[:, None] * freqs[None, :]      # rotary_embedding_torch: pos * freq, fp32
            cos, sin = torch.ones(P, 32), torch.zeros(P, 32)                           # kernel tables: 32 pairs per head; identity past dim_head / 2
            cos[:, :freqs.numel()] = ang.cos(); sin[:, :freqs.numel()] = ang.sin()
            self._rope = (cos.to(self.device).contiguous(), sin.to(self.device).contiguous())
        return self._rope

    def _plan(self, b, n, I, R, training):
        dp_groups = getattr(self, '_dp_groups', 0) if training else 0
        key = (b, n, I, tuple(sorted(R.items())), training, dp_groups)
        plan = self._plans.pop(key, None)
        if plan is None:
            while len(self._plans) >= 8:                             # least recently used plan goes first (dict order = use order)
                self._plans.pop(next(iter(self._plans)))
            plan = Plan(self.store, b, n, I, R, training=training, dp_groups=dp_groups)
            # a plan owns every activation of its step (tens of GB for a training plan at dim 1024 / depth 24): besides the count, the cache is
            # bounded by bytes - TFX_PLAN_BUDGET_GB, default 40 % of the device memory - oldest first, the new plan always stays
            budget = float(os.environ.get('TFX_PLAN_BUDGET_GB', 0)) * 2 ** 30 or 0.4 * torch.cuda.get_device_properties(self.device).total_memory
            while self._plans and sum(p.nbytes for p in self._plans.values()) + plan.nbytes > budget:
                self._plans.pop(next(iter(self._plans)))
        self._plans[key] = plan                                      # (re-)insert at the most recent position
        return plan

    def _build_structure(self, modalities, return_loss, add_meta=True, pad_n=1, presig=None):
        """full structure scan (host) + upload of every derived index array; cached per structure signature.
        `add_meta=False`: the decode-time layout (`return_embed` in the reference, MP:330): no [meta][shape][som][eom]
        tokens are added around modalities.  `pad_n`: round the packed length up (keeps the number of distinct plans small).
        `presig`: (signature, text tensors, latent tensors) of `fast_signature` on the same batch - the scan then runs on the signature alone."""
        dev = self.device
        if presig is not None:
            P = scan_signature(*presig, num_modalities=self.num_modalities, dim_latents=self.dim_latents, sos_id=self.sos_id, eos_id=self.eos_id,
                               meta_id=self.meta_id, som_ids=self.som_ids, eom_ids=self.eom_ids, add_sos_eos=return_loss, add_meta=add_meta)
        else:
            P = self._scan(modalities, add_sos_eos=return_loss, add_meta=add_meta)
        b = P.b
        n = P.n_full - 1 if return_loss else P.n_full
        n_true = n
        if pad_n > 1 and n % pad_n:
            # padding columns sit AFTER every real token of a row: causal attention keeps them out of the real tokens' sight, they carry label -1
            # (no loss, zero gradient) and text id -1 (embedded as id 0); `total_tokens` and the loss weights count real tokens only
            n = (n + pad_n - 1) // pad_n * pad_n
            n_new = n + 1 if return_loss else n                  # the training layout keeps one extra column: labels are ids shifted by one (T:3144)
            old = P.n_full
            th = np.full((b, n_new), -1, dtype=np.int32); th[:, :old] = P.text_host; P.text_host = th
            cd = np.zeros((b, n_new), dtype=bool); cd[:, :old] = P.cfg_droppable; P.cfg_droppable = cd
            P.text_dest = (P.text_dest // old) * n_new + (P.text_dest % old)
            for t in P.row_pos:
                rp = P.row_pos[t].astype(np.int64)
                P.row_pos[t] = ((rp // old) * n_new + (rp % old)).astype(np.int32)
            P.n_full = n_new
        tm = token_maps(P, n, self.num_modalities)
        seg_start, seg_len = token_segments(tm.tok_inst, balance=True)
        # every index array of the structure goes up in ONE pinned, asynchronous copy (an int32 arena; the device tensors below are views of it): a
        # structure miss used to issue ~16 pageable host-to-device copies, each a blocking round trip of the host (0.55 ms apiece under load)
        R = {t: int(len(v)) for t, v in P.row_inst.items()}
        num_mod = np.bincount(P.inst_b, minlength=b)
        host = dict(text_host=P.text_host, text_dest=P.text_dest, cfg_droppable=P.cfg_droppable, tok_inst=tm.tok_inst, kv_end=tm.kv_end, q_start=tm.q_start,
                    rot_pos=tm.rot_pos, inst_b=P.inst_b, inst_m=P.inst_m, seg_start=seg_start, seg_len=seg_len, num_mod=num_mod)
        for t in R:
            rp = P.row_pos[t].astype(np.int64)
            rb, rl = rp // P.n_full, rp % P.n_full
            host[('row_tok', t)] = np.where(rl < n, rb * n + rl, -1)
            host[('row_inst', t)] = P.row_inst[t]
        offs, pos = {}, 0
        for k, a in host.items():
            offs[k] = (pos, a.size, a.shape); pos += (a.size + 3) // 4 * 4              # 16-byte aligned slices
        arena = torch.empty(max(pos, 4), dtype=torch.int32, pin_memory=(dev.type == 'cuda'))
        av = arena.numpy()
        for k, a in host.items():
            o, sz, _ = offs[k]
            av[o:o + sz] = np.asarray(a).reshape(-1)
        darena = arena.to(dev, non_blocking=True)
        D = lambda k: darena[offs[k][0]:offs[k][0] + offs[k][1]].view(offs[k][2])
        tok_inst = D('tok_inst')
        S = dict(P=P, tm=tm, b=b, n=n, n_true=n_true, I=len(P.inst_b), R=R, num_mod=num_mod,
                 text_host=D('text_host'), text_dest=D('text_dest').long(), cfg_droppable=D('cfg_droppable') != 0, tok_inst=tok_inst,
                 kv_end=D('kv_end').reshape(-1), q_start=D('q_start').reshape(-1), rot_pos=D('rot_pos').reshape(-1),
                 is_mod=tok_inst >= 0, minus1=torch.full((b, n), -1, dtype=torch.int32, device=dev),
                 inst_b=D('inst_b').long(), inst_m=D('inst_m').long(), row_tok={t: D(('row_tok', t)) for t in R}, row_inst={t: D(('row_inst', t)) for t in R},
                 seg_start=D('seg_start'), seg_len=D('seg_len'), _arena=(arena, darena))
        S['num_mod_dev'] = D('num_mod').float()
        D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if self.has_recon_loss and return_loss:
            # reconstruction loss = mean over the instances of a type of the per-instance mse (T:3422-3426): every row weighs 1 / (instances x its rows)
            n_inst = np.bincount(P.inst_type, minlength=self.num_modalities).astype(np.float64)
            S['recw'] = {t: D((1. / (n_inst[t] * P.inst_len[P.row_inst[t]].astype(np.float64))).astype(np.float32)) for t in R}
        P.user_text, P.latents = None, None          # the cache keeps structure only, never the caller's tensors
        return S

    def _scan(self, modalities, add_sos_eos, add_meta=True):
        return scan_batch(modalities, num_modalities=self.num_modalities, dim_latents=self.dim_latents, sos_id=self.sos_id, eos_id=self.eos_id,
                          meta_id=self.meta_id, som_ids=self.som_ids, eom_ids=self.eom_ids, add_sos_eos=add_sos_eos, add_meta=add_meta)

    def _forward_plain(self, samples, times, add_meta=False, pad_n=64):
        """inference forward over explicit samples (nothing added, no noising): fills and runs a non-training plan up to the
        fp32 logits.  Returns (plan, structure).  Used by the sampler's prefills (T:2194-2201, T:2389-2406)."""
        self._require_gpu()
        dev, stream = self.device, self._stream()
        ext_ctx = None
        if self._ext:                                # the user's encoders produce the token rows (encoder layout in, as forward() hands it over)
            with torch.no_grad():
                samples, ext_ctx = self._ext_preprocess(samples, None, return_loss=False)
        sig, user_text, latents = fast_signature(samples)
        key = (sig, 'plain', add_meta, pad_n)
        S = self._struct_cache.pop(key, None)
        if S is None:
            while len(self._struct_cache) >= 32:
                self._struct_cache.pop(next(iter(self._struct_cache)))
            S = self._build_structure(samples, False, add_meta=add_meta, pad_n=pad_n)
        self._struct_cache[key] = S
        tm, b, n, I, R = S['tm'], S['b'], S['n'], S['I'], S['R']
        self.store.refresh_shadows(stream)
        # The prefill plans of a decode loop differ only in their instance / latent-row counts from one modality phase to the next: round
        # both up (instances to 64, rows to 256) so that ONE plan per (batch, padded length) serves them all - a plan holds every activation
        # of the forward, building one costs more than running it.  Padding rows scatter nowhere (row_tok = -1), padding instances are
        # referenced by no token.
        Ip = -(-I // 64) * 64 if I > 0 else 0
        Rp = {t: -(-r // 256) * 256 for t, r in R.items()}
        if Ip > 0:
            for t in range(self.num_modalities):
                Rp.setdefault(t, 256)
        plan = self._plan(b, n, Ip, Rp, training=False)
        plan.set_rope_tables(*self._rope_tables(int(tm.rot_pos.max()) if tm.rot_pos.size else 0))
        plan.tok_inst.copy_(S['tok_inst'].view(-1)); plan.kv_end.copy_(S['kv_end']); plan.q_start.copy_(S['q_start']); plan.rot_pos.copy_(S['rot_pos'])
        plan.loaded_structure = S
        text_full = S['text_host'].clone()
        if user_text:
            text_full.view(-1).index_copy_(0, S['text_dest'], torch.cat(user_text).to(dev, torch.int32))
        plan.text_ids.copy_(text_full[:, :n].reshape(-1))
        if I > 0:
            plan.inst_time.zero_()
            plan.inst_time[:I].copy_(times.to(dev, torch.float32)[S['inst_b'], S['inst_m']])
        for t in Rp:
            plan.row_tok[t].fill_(-1); plan.row_inst[t].zero_()
            if t in R:
                r = R[t]
                plan.row_tok[t][:r].copy_(S['row_tok'][t]); plan.row_inst[t][:r].copy_(S['row_inst'][t])
                if t in plan.ext:
                    plan.lat[t]['tok'][:r].copy_(torch.cat(ext_ctx[t]['tok']))
                else:
                    plan.lat[t]['x'][:r].copy_(torch.cat(latents[t]).to(dev, torch.float32))
                if t in plan.ext_add:                                 # prompted modalities carry their axial positional embedding (T:3173-3176)
                    P = S['P']
                    plan.lat[t]['add'][:r].copy_(self._pos_rows(t, [P.inst_shape[g] for g in range(I) if int(P.inst_type[g]) == t]))
            if t not in plan.ext:
                plan.set_noise(t, None)
        Plan.run(plan.fwd, stream, 0, plan.fwd_logits_end)
        return plan, S

    def _default_times(self, num_modalities_host: np.ndarray, nm=None):
        """default_modality_length_to_time_fn, T:186-200 (device RNG).  `nm`: cached device copy of the counts
        (a pageable host->device copy here would serialise the host with the GPU queue every step)."""
        b, m = len(num_modalities_host), int(num_modalities_host.max()) if len(num_modalities_host) else 0
        if m == 0:
            return torch.empty((b, 0), device=self.device)
        if nm is None:
            nm = torch.from_numpy(num_modalities_host.astype(np.float32)).to(self.device)
        rand_num = torch.floor(torch.rand(b, device=self.device) * nm)
        seq = torch.arange(m, device=self.device)
        prev = seq[None, :] < rand_num[:, None]
        cur = torch.rand(b, device=self.device)
        return torch.where(prev, torch.full((), 0.5, device=self.device), cur[:, None].expand(b, m))

    # ------------------------------------------------------------------ forward
    def forward(self, modalities, times=None, num_modalities_to_times_fn=None, modality_type=None, cache=None, decode_length=None,
                decoding_text_or_modality=None, velocity_consistency_ema_model=None, velocity_consistency_delta_time=1e-3,
                return_only_pred_flows=False, return_loss=True, return_breakdown=False, return_embed=False, return_hiddens=False,
                return_kv_cache=False, return_times=False, prob_uncond=None):
        self._require_gpu()
        # the reference looks its packer up in the registry on every call (get_processing_strategy, MP:1252-1256, called at T:3104-3107).  The fused
        # step packs with the native packer only (every name the reference ships maps to it): an entry someone replaced or added must not be
        # ignored silently
        from .modality_processing import process_native
        if PROCESSING_STRATEGIES.get(self.modality_processing) is not process_native:
            raise NotImplementedError(f'PROCESSING_STRATEGIES[{self.modality_processing!r}] is not the native packer: Transfusion.forward runs the fused MI355X step, '
                                      'whose packing (host structure scan + tfx_noise_mix + the row-scattered latent_to_model GEMM) is part of its launch list; '
                                      'call a custom strategy directly (it receives the model) or restore the registry entry')
        if torch.is_tensor(modalities):
            if modalities.dtype in (torch.int32, torch.int64):                             # T:2967-2968
                return self.forward_text(modalities, return_loss=return_loss, return_embed=return_embed, cache=cache,
                                         return_hiddens=return_hiddens, return_kv_cache=return_kv_cache)
            return self.forward_modality(modalities, times=times, modality_type=modality_type, return_loss=return_loss,             # T:2989-2990
                                         velocity_consistency_ema_model=velocity_consistency_ema_model, return_loss_breakdown=return_breakdown)
        is_decoding = decoding_text_or_modality is not None
        velocity_modalities = modalities                                                   # the EMA teacher gets the caller's raw samples (T:3004-3008): it encodes them itself
        if isinstance(modalities, list) and (any(self.channel_first_latent) or any(e is not None for e in self.modality_encoder)):
            modalities = [list(sample) for sample in modalities]                           # T:3010-3012: never mutate the caller's lists
            if not is_decoding:
                modalities = self._encode_modalities(modalities)                           # T:3094-3101
            modalities = self._to_channel_last(modalities)
        if cache is not None or decoding_text_or_modality is not None or return_kv_cache or return_hiddens:
            if return_loss and not return_embed:
                raise NotImplementedError('kv cache / hiddens are returned by the inference forward only (return_loss = False or return_embed = True), '
                                          'as in the reference\'s own decode calls (T:1917-1924, T:1998-2006)')
            return self._forward_decode(modalities, times, cache, decode_length, decoding_text_or_modality, return_embed=return_embed,
                                        return_kv_cache=return_kv_cache, return_hiddens=return_hiddens, return_times=return_times)
        ema = velocity_consistency_ema_model
        if ema is not None and hasattr(ema, 'ema_model'):                                  # EMA wrapper, T:2967-2969
            ema = ema.ema_model
        if ema is not None and not isinstance(ema, Transfusion):
            raise NotImplementedError('velocity_consistency_ema_model must be a native Transfusion (or the EMA wrapper of one)')
        if ema is self:
            raise ValueError('velocity_consistency_ema_model is the model itself: the teacher pass would overwrite the activations the backward needs - pass an EMA copy (create_ema)')
        return_loss = (return_loss and not return_embed) or return_only_pred_flows or ema is not None
        dev = self.device
        stream = self._stream()
        ps, md = self.store, self.md

        # ---- modality types whose encoder / decoder are user modules: their part of the packing happens in PyTorch, BEFORE the structure scan
        ext_ctx = orig_times = None
        packer_in = modalities                                # what the packer sees (channel-last; `ext` types in their encoder's layout)
        if self._ext:
            is_mod = lambda p: isinstance(p, tuple) or (torch.is_tensor(p) and p.is_floating_point())
            if times is None:                                                              # T:3075-3082 (drawn before the packing, as there)
                num_mod = np.array([sum(1 for p in sample if is_mod(p)) for sample in modalities], dtype=np.int64)
                fn = num_modalities_to_times_fn
                times = fn(torch.from_numpy(num_mod).to(dev)) if fn is not None else self._default_times(num_mod)
            times = times.to(dev, torch.float32)
            if ema is not None:                                                            # T:3086-3088: the packing below noises at the shortened times
                orig_times = times.clone()
                times = times * (1. - velocity_consistency_delta_time)
            modalities, ext_ctx = self._ext_preprocess(modalities, times, return_loss)

        # ---- structure: one cheap signature pass; everything derived from it is cached ON THE DEVICE per signature
        sig, user_text, latents = fast_signature(modalities)
        add_meta = return_loss or not return_embed           # MP:330: `return_embed` (the decode-time call) packs WITHOUT [meta][shape][som] ... [eom]
        skey = (sig, return_loss, add_meta)
        S = self._struct_cache.pop(skey, None)
        if S is None:
            while len(self._struct_cache) >= 32:                      # least recently used structure goes first (dict order = use order)
                self._struct_cache.pop(next(iter(self._struct_cache)))
            # training lengths are bucketed to multiples of 64 (_TRAIN_PAD): ragged data then shares a handful of plans - a plan
            # owns every activation of the step and its launch lists, building one costs far more than the padding columns
            S = self._build_structure(modalities, return_loss, add_meta=add_meta, pad_n=_TRAIN_PAD if return_loss else 1, presig=(sig, user_text, latents))
        self._struct_cache[skey] = S
        P, tm, b, n, I, R = S['P'], S['tm'], S['b'], S['n'], S['I'], S['R']
        self._live_n_true = S['n_true']

        # ---- times (T:3075-3082)
        if times is None:
            fn = num_modalities_to_times_fn
            times = fn(S['num_mod_dev'].long()) if fn is not None else self._default_times(S['num_mod'], S['num_mod_dev'])
        times = times.to(dev, torch.float32)
        if ema is not None and orig_times is None:                                         # T:3086-3088
            orig_times = times.clone()
            times = times * (1. - velocity_consistency_delta_time)

        ps.refresh_shadows(stream)
        # Ragged corpora change the number of modality instances and of latent rows with every batch.  A training plan owns every activation of the
        # step and its launch lists (building one costs far more than a step), so in the plain training case the plan is built for both counts ROUNDED
        # UP - in steps of a quarter of the count's power of two, at least 64 instances / 256 rows per type (`_bucket`) - and shared: padding instances are referenced by no token (their table gradients stay zero),
        # padding rows scatter nowhere and are kept out of the losses (engine.Plan.set_rows).  TFX_PLAN_BUCKETS=0: exact counts (one plan per pair).
        bucket = (return_loss and not self._ext and not md.pos_types and ema is None and not return_only_pred_flows and not self.has_recon_loss
                  and not md.model_output_clean and os.environ.get('TFX_PLAN_BUCKETS', '1') != '0')
        Ip = _bucket(I, 64) if (bucket and I > 0) else I
        Rp = {t: _bucket(r, 256) for t, r in R.items()} if bucket else R
        plan = self._plan(b, n, Ip, Rp, training=return_loss)
        if md.model_output_clean:
            plan.set_clean_mode('model')                    # interleaved path: the model-space conversion (MP:786-792)
        if plan.loaded_structure is not S:
            plan.set_rope_tables(*self._rope_tables(int(tm.rot_pos.max()) if tm.rot_pos.size else 0))
            plan.tok_inst.copy_(S['tok_inst'].view(-1)); plan.kv_end.copy_(S['kv_end']); plan.q_start.copy_(S['q_start']); plan.rot_pos.copy_(S['rot_pos'])
            plan.set_segments(S['seg_start'], S['seg_len'])
            for t, r in R.items():
                if r < Rp[t]:
                    plan.row_tok[t].fill_(-1); plan.row_inst[t].zero_()
                plan.row_tok[t][:r].copy_(S['row_tok'][t]); plan.row_inst[t][:r].copy_(S['row_inst'][t])
            if return_loss:
                plan.set_rows(R)
            plan.loaded_structure = S

        # ---- token ids on device (values never visit the host)
        text_full = S['text_host'].clone()
        if user_text:
            vals = torch.cat(user_text).to(dev, torch.int32)
            text_full.view(-1).index_copy_(0, S['text_dest'], vals)
        prob_uncond = self.prob_uncond if prob_uncond is None else prob_uncond
        if self.training and prob_uncond > 0:                                              # CFG text drop, T:3027-3043
            drop_rows = torch.rand(b, device=dev) < prob_uncond
            text_full = text_full.masked_fill(S['cfg_droppable'] & drop_rows[:, None], self.null_text_id)
        plan.text_ids.copy_(text_full[:, :n].reshape(-1))
        if return_loss:
            lab = text_full[:, 1:]                                                          # T:3144
            lab = torch.where(S['is_mod'] | (lab == self.null_text_id), S['minus1'], lab)    # T:3320-3323
            plan.labels.copy_(lab.reshape(-1))
        if I > 0:
            plan.inst_time[:I].copy_(times[S['inst_b'], S['inst_m']])
        rows_in = []                                                      # rows PyTorch hands to the engine: (kind, type, tensor with autograd history)
        for t in R:
            lt = plan.lat[t]
            if t in plan.ext_add:                                         # axial positional embedding of the (projected) instance shapes
                rows = self._pos_rows(t, [P.inst_shape[g] for g in range(I) if int(P.inst_type[g]) == t])
                lt['add'].copy_(rows.detach())
                rows_in.append(('add', t, rows))
            if t in plan.ext:
                rows = torch.cat(ext_ctx[t]['tok']) if len(ext_ctx[t]['tok']) > 1 else ext_ctx[t]['tok'][0]
                lt['tok'].copy_(rows.detach())
                rows_in.append(('tok', t, rows))
                continue
            lt['x'][:R[t]].copy_(torch.cat(latents[t]), non_blocking=True)   # one cat on the source device, one transfer
            if return_loss:
                if self._noise_override is not None:
                    lt['eps'][:R[t]].copy_(self._noise_override[t])
                else:
                    lt['eps'][:R[t]].normal_()                                              # MP:654 (the REAL rows only: the RNG stream a seed consumes does not depend on the plan's bucket padding)
            # without a loss there is no noising (MP:658-660): noise_mix with eps = NULL copies x
            plan.set_noise(t, lt['eps'].data_ptr() if return_loss else None)

        if not return_loss:
            end = plan.fwd_embed_end if return_embed else plan.fwd_logits_end
            Plan.run(plan.fwd, stream, 0, end)
            if return_embed:                                                               # T:3275-3276: (embed, get_pred_flows)
                out = (plan.embed.view(b, n, md.dim).float(), self._pred_flow_closures(P, self._clean_sources(packer_in, times) if md.model_output_clean else None))
                return out if not return_times else (out, times)
            logits = plan.logits.view(b, n, md.vp)[..., :md.vocab].clone()
            return (logits, times) if return_times else logits

        if return_only_pred_flows:                                                        # T:3313-3316 (the EMA teacher's call)
            Plan.run(plan.fwd, stream, 0, plan.fwd_pred_end)
            # types with user modules: their decoder on the embedding rows of every instance, in PyTorch (the teacher's call runs under no_grad)
            ext_preds = {t: self._ext_decode(t, plan.embed.index_select(0, plan.row_tok[t].long().clamp(min=0)).float(), ext_ctx[t]) for t in R if t in plan.ext}
            if getattr(self, '_flat_pred_flows', False):
                return {t: (ext_preds[t] if t in plan.ext else plan.lat[t]['pred']) for t in R}
            out = [[] for _ in range(self.num_modalities)]
            cursor = {t: 0 for t in R}
            for gi in range(len(P.inst_b)):
                t, L = int(P.inst_type[gi]), int(P.inst_len[gi])
                if t in plan.ext:
                    out[t].append(ext_preds[t][cursor[t]]); cursor[t] += 1
                    continue
                rows = plan.lat[t]['pred'][cursor[t]:cursor[t] + L]; cursor[t] += L
                out[t].append(self._from_channel_last(t, rows.view(*P.inst_shape[gi], md.dim_latents[t]).clone()))
            return out

        # ---- loss seeds (the token-count normalisers cancel: d loss / d logit = w / total_tokens, T:3331)
        total = float(P.total_tokens)
        mse_scales = {}
        for t, r in R.items():
            if t in plan.ext:
                continue
            w_t = float(tm.is_type[t]) / total                                              # T:3343
            mse_scales[t] = 2.0 * self.flow_loss_weight * w_t / (r * md.dim_latents[t])
        plan.set_loss_scales(self.text_loss_weight / total, mse_scales)
        plan.set_ce_vocab(md.vocab)
        self._bwd_scale = None
        plan.acc.zero_()
        Plan.run(plan.fwd, stream, graph='auto')

        acc = plan.acc
        text_loss = acc[0] / acc[1].clamp(min=1.)
        loss = self.text_loss_weight * acc[0] / total
        flow_losses = {}
        for t, r in sorted(R.items()):
            if t in plan.ext:
                continue
            fl = acc[2 + t] / (r * md.dim_latents[t])
            flow_losses[t] = fl
            loss = loss + self.flow_loss_weight * fl * (float(tm.is_type[t]) / total)

        velocity_losses = None
        if ema is not None:                                                                # T:3383-3418
            M = self.num_modalities
            was_training = ema.training
            ema.eval()
            ema._flat_pred_flows = True
            try:
                with torch.no_grad():
                    teacher = ema(velocity_modalities, times=orig_times + velocity_consistency_delta_time, return_only_pred_flows=True)
            finally:
                ema._flat_pred_flows = False
                ema.train(was_training)
            velocity_losses = {}
            for t, r in sorted(R.items()):
                if t in plan.ext:                         # the student's side of these needs the user's decoder: below, with the flow loss
                    continue
                w_t = float(tm.is_type[t]) / total
                plan.lat[t]['vel'].copy_(teacher[t])
                va = plan._vel_args[t]
                va.pred, va.flow = plan.lat[t]['pred'].data_ptr(), plan.lat[t]['vel'].data_ptr()
                va.grad_scale = 2.0 * self.velocity_consistency_loss_weight * w_t / (r * md.dim_latents[t])
            Plan.run(plan.vel, stream)
            for t, r in sorted(R.items()):
                if t in plan.ext:
                    continue
                vl = plan.acc[2 + M + t] / (r * md.dim_latents[t])
                velocity_losses[t] = vl
                loss = loss + self.velocity_consistency_loss_weight * vl * (float(tm.is_type[t]) / total)

        recon_losses = {}
        if self.has_recon_loss:                                                            # T:3420-3431
            M = self.num_modalities
            for t, r in sorted(R.items()):
                if t in plan.ext:
                    continue
                w_t = float(tm.is_type[t]) / total
                plan.lat[t]['recw'].copy_(S['recw'][t])
                ra = plan._rec_args[t]
                ra.recon_mode, ra.grad_scale = 0, 2.0 * self.reconstruction_loss_weight * w_t / md.dim_latents[t]
            Plan.run(plan.rec, stream)
            for t, r in sorted(R.items()):
                if t in plan.ext:
                    continue
                recon_losses[t] = plan.acc[2 + 2 * M + t] / md.dim_latents[t]
                loss = loss + self.reconstruction_loss_weight * recon_losses[t] * (float(tm.is_type[t]) / total)

        self._step_id += 1
        self._live = (plan, self._step_id)
        ext_out = sorted(plan.ext)
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros((), device=dev, requires_grad=True)
            spec = {'in': [(k, t) for k, t, _ in rows_in], 'out': ext_out} if (rows_in or ext_out) else _NO_ROWS
            loss, *emb_rows = _NativeLoss.apply(self._anchor, self, loss, spec, *[r for _, _, r in rows_in])
        else:
            emb_rows = [plan.embed.index_select(0, plan.row_tok[t].long().clamp(min=0)).float() for t in ext_out]
        for t, rows in zip(ext_out, emb_rows):                                              # the user's decoders and their flow losses, in PyTorch
            preds = self._ext_decode(t, rows, ext_ctx[t])
            fl, rec = self._ext_flow_loss(t, preds, ext_ctx[t])
            flow_losses[t] = fl
            loss = loss + self.flow_loss_weight * fl * (float(tm.is_type[t]) / total)
            if rec is not None:
                recon_losses[t] = rec
                loss = loss + self.reconstruction_loss_weight * rec * (float(tm.is_type[t]) / total)
            if ema is not None:                                                             # T:3397-3411: mse(student flows, teacher flows), all instances of the type packed
                vl = torch.nn.functional.mse_loss(torch.cat([p.reshape(-1) for p in preds]), torch.cat([p.reshape(-1) for p in teacher[t]]))
                velocity_losses[t] = vl
                loss = loss + self.velocity_consistency_loss_weight * vl * (float(tm.is_type[t]) / total)
        if velocity_losses is not None:
            velocity_losses = [velocity_losses[t] for t in sorted(velocity_losses)]
        flow_losses = [flow_losses[t] for t in sorted(flow_losses)]
        recon_losses = [recon_losses[t] for t in sorted(recon_losses)] if self.has_recon_loss else None
        if not return_breakdown and not return_times:
            return loss
        ret = (loss,)
        if return_breakdown:
            ret = (*ret, LossBreakdown(loss, text_loss, flow_losses, velocity_losses, recon_losses))
        if return_times:
            ret = (*ret, times)
        return ret

    # ------------------------------------------------------------------ decode contract of forward() (T:2926-2948, T:3186-3271)
    def _pred_flow_closures(self, P, clean_src=None):
        """`get_pred_flows` of the reference (build_record_closures MP:764-805, model_to_pred_flow MP:160-175): per modality type, in scan
        order, closures that cut an instance's rows out of an `embed` (b, n, d) tensor and reshape them to (*axial shape, d).
        `model_output_clean` (`clean_src`: per instance its (type, tensor as the packer saw it, time)): every closure is wrapped in the model-space
        conversion (embed - projected tokens) / max(1 - t, eps) (get_model_output_to_flow_fn as decorator, MP:99-126, MP:786-792) - the projection
        of the instance is evaluated when the closure is called (the samplers only ever call the last one)."""
        out = [[] for _ in range(self.num_modalities)]
        d, eps = self.md.dim, self.md.clean_eps
        for gi in range(len(P.inst_b)):
            bi, off, L, shape = int(P.inst_b[gi]), int(P.inst_off[gi]), int(P.inst_len[gi]), tuple(P.inst_shape[gi])

            def inner(embed, need_splice=True, bi=bi, off=off, L=L, shape=shape):
                e = embed[bi]
                if need_splice:
                    e = e[-L:] if e.shape[0] < off + L else e[off:off + L]      # MP:167-171: a decode-step embed holds the new block only
                return e.reshape(*shape, d)
            if clean_src is not None:
                def inner(embed, need_splice=True, fn=inner, src=clean_src[gi]):
                    o = fn(embed, need_splice)
                    ty, x, tt = src
                    return (o - self._packed_tokens(ty, x).reshape_as(o).to(o)) / (1. - tt).clamp_min(eps)
            out[int(P.inst_type[gi])].append(inner)
        return out

    def _packed_tokens(self, ty, x):
        """`processed.packed` of an instance (MP:729-732): latent_to_model of the (noised) latents - the user's encoder for an `ext` type (encoder
        layout in), else the Linear in fp32 from the master weights on (*axial, dim_latent) - before any positional embedding (T:3173-3176)"""
        x = x.to(self.device, torch.float32)
        if ty in self._ext:
            pre = self.latent_to_model_projs[ty]
            return pre(x[None])[0] if self.channel_first_latent[ty] else pre(x)
        if self.md.dim_latents[ty] == self.md.dim:
            return x
        return torch.nn.functional.linear(x, self.store.view(f'latent_to_model_projs.{ty}.weight'), self.store.view(f'latent_to_model_projs.{ty}.bias'))

    def _clean_sources(self, modalities, times):
        """per modality instance of the batch, in scan order: (type, tensor, time) for the `model_output_clean` closures of the decode forward"""
        out = []
        for bi, sample in enumerate(modalities):
            m = 0
            for part in sample:
                if torch.is_tensor(part) and part.is_floating_point():
                    part = (0, part)
                if isinstance(part, tuple):
                    #  the closure divides (embed - tokens) by max(1 - t, eps): without the caller's times it would silently run at t = 1,
                    # i.e. multiply by 1 / eps - the un-cached forward always has times here (drawn like the reference's, T:3075-3082); a cached
                    # text step that asks for clean flows of its prefix must pass them
                    if times is None or times.ndim != 2 or times.shape[1] == 0:       # (a ValueError, not an assert: `python -O` strips asserts)
                        raise ValueError('`model_output_clean` with `return_embed`: pass `times` (one column per modality instance) - the clean-to-flow closures '
                                         'need each instance\'s time')
                    tt = times[bi, min(m, times.shape[1] - 1)]
                    out.append((int(part[0]), part[1], tt.to(self.device, torch.float32)))
                    m += 1
        return out

    def model_to_latent(self, modality_type: int, embed_rows):
        """`model_to_latent_projs[type]` (Linear(dim, dim_latent, bias=False), T:1479) applied to rows of the final embedding - the last
        step of the reference's external decode loop (T:2013-2015) - through the native NT GEMM.  embed_rows: (..., dim) -> (..., dim_latent) fp32."""
        self._require_gpu()
        if modality_type in self._ext:                                 # the user's decoder (add_temp_batch_dim(mod.model_to_latent), T:2013); (*axial, dim) in
            return self.model_to_latent_projs[modality_type](embed_rows.to(self.device, torch.float32)[None])[0]
        md, dev, stream = self.md, self.device, self._stream()
        self.store.refresh_shadows(stream)
        x = embed_rows.reshape(-1, md.dim).to(dev, torch.bfloat16).contiguous()
        dl = md.dim_latents[modality_type]
        out = torch.empty(x.shape[0], dl, device=dev, dtype=torch.float32)
        a = capi.make_args('tfx_gemm_nt_args', A=x, lda=md.dim, B=self.store.shadows[f'outp{modality_type}'], ldb=md.dim, M=x.shape[0], N=dl, K=md.dim,
                           epi=capi.ENUMS['TFX_EPI_F32'], C=out, ldc=dl)
        capi.call('tfx_gemm_nt', a, stream)
        return out.reshape(*embed_rows.shape[:-1], dl)

    def _kv_public(self, buf, length):
        """the public kv-cache object of forward() (T:3252: `(tensor, tokens_seen)`): a VIEW of the native cache buffer
        [depth, b, capacity, k~ | v (heads x 64 each)] in the reference's layout (layers, key/value, batch, heads, seq, dim_head) (T:977, T:1264)."""
        md = self.md
        D, b, cap, _ = buf.shape
        v = buf.view(D, b, cap, 2, md.heads, 64)[:, :, :length, :, :, :md.dim_head].permute(0, 3, 1, 4, 2, 5)
        v = v.as_subclass(KVCacheView)
        v._tfx_buf, v._tfx_len = buf, length
        return v

    def _kv_native(self, kv, extra):
        """native buffer [depth, b, capacity >= length + extra, 2 * heads * 64] behind a public kv-cache tensor.  Our own views are
        appended IN PLACE while they are the newest view of their buffer (linear decoding); anything else is copied."""
        md, dev = self.md, self.device
        buf, length = getattr(kv, '_tfx_buf', None), getattr(kv, '_tfx_len', None)
        if buf is not None and getattr(buf, '_tfx_filled', None) == length and buf.shape[2] >= length + extra:
            return buf, length
        D, two, b, h, n, dh = kv.shape
        assert (D, two, h, dh) == (md.depth, 2, md.heads, md.dim_head), 'kv cache tensor must be (layers, 2, batch, heads, seq, dim_head) (T:977)'
        cap = -(-(n + extra + 64) // 64) * 64
        new = torch.zeros(D, b, cap, 2 * md.hdk, device=dev, dtype=torch.bfloat16)
        new.view(D, b, cap, 2, h, 64)[:, :, :n, :, :, :dh].copy_(kv.permute(0, 2, 4, 1, 3, 5))
        return new, n

    def _forward_decode(self, modalities, times, cache, decode_length, decoding, return_embed, return_kv_cache, return_hiddens, return_times):
        """`forward(..., return_loss=False)` with the decode-time arguments of the reference (T:2926-2948):

          cache=None          full forward over the samples (meta tokens only when logits are asked for, MP:330); `return_kv_cache` hands back
                              `(kv, tokens_seen)` with tokens_seen = last rotary position + 1 (T:3215)
          cache=(kv, seen)    only the LAST `decode_length` tokens of every sample run (T:1167-1176), against the cache: 'text' = one token at
                              rotary position `seen` (T:3196, T:3210-3211); 'modality' = the trailing modality block, conditioned on its time,
                              every token at rotary position `seen`, attending to the cache and to the whole block (T:3206-3208, T:2409-2419)
        Returns logits (b, n | decode_length, V) - or `(embed, get_pred_flows)` with `return_embed` - followed by the requested extras in the
        reference's order: kv cache, hiddens, times (maybe_pack_aux, T:3256-3271)."""
        from .sampling import Sampler
        md, dev, stream = self.md, self.device, self._stream()
        assert isinstance(modalities, list), 'the decode forward takes the list-of-samples input'
        b = len(modalities)
        if cache is None:
            if times is None:
                nm = max((sum(1 for p in s if isinstance(p, tuple) or (torch.is_tensor(p) and p.is_floating_point())) for s in modalities), default=0)
                times = self._default_times(np.array([sum(1 for p in s if isinstance(p, tuple) or (torch.is_tensor(p) and p.is_floating_point()))
                                                     for s in modalities])) if nm else torch.empty((b, 0), device=dev)
            # lengths are bucketed to multiples of 64 (an un-cached decode loop calls this once per token: one plan per bucket, not per length);
            # the padding columns sit behind every real token and are cut off again below
            plan, S = self._forward_plain(modalities, times, add_meta=not return_embed, pad_n=64)
            n_pad, n, P, tm = S['n'], S['n_true'], S['P'], S['tm']
            out = (plan.embed.view(b, n_pad, md.dim)[:, :n].float(),
                   self._pred_flow_closures(P, self._clean_sources(modalities, times) if md.model_output_clean else None)) if return_embed \
                else plan.logits.view(b, n_pad, md.vp)[:, :n, :md.vocab].clone()
            kv = None
            if return_kv_cache:
                buf = torch.zeros(md.depth, b, -(-(n_pad + 64) // 64) * 64, 2 * md.hdk, device=dev, dtype=torch.bfloat16)
                Sampler(self)._fill_cache(buf, plan, b, n_pad)
                buf._tfx_filled = n
                rp = tm.rot_pos.reshape(b, n_pad)[:, :n]
                kv = (self._kv_public(buf, n), int(rp.max()) + 1 if rp.size else 0)
        else:
            assert decode_length is not None, '`decode_length` must be passed in on forward for modality sampling. think of a cleaner way on some future date'   # T:3191
            assert decoding in ('text', 'modality')                                                                                                          # T:3192
            kv_in, seen = cache
            L = 1 if decoding == 'text' else int(decode_length)
            buf, n_cached = self._kv_native(kv_in, L)
            assert buf.shape[1] == b, 'kv cache batch does not match the number of samples'
            T = b * L
            if not hasattr(self, '_decode_plans') or len(self._decode_plans) > 8:
                self._decode_plans = {}
            smp = Sampler(self)
            is_mod = decoding == 'modality'
            plan = smp._decode_plan(('fwd', decoding, L, buf.data_ptr()), b, L, buf, is_mod)
            ids = np.zeros(T, np.int32); tok_inst = np.full(T, -1, np.int32)
            pos = (np.arange(b, dtype=np.int32)[:, None] * buf.shape[2] + n_cached + np.arange(L, dtype=np.int32)[None, :]).reshape(-1)
            if not is_mod:
                for i, sample in enumerate(modalities):
                    last = sample[-1]
                    assert is_int_tensor(last) and last.numel() > 0, 'decoding text: every sample must end in a text token'
                    ids[i] = int(last.reshape(-1)[-1])
                kve = np.full(T, n_cached + 1, np.int32)
                rot = np.full(T, seen, np.int32)
            else:
                tys = []
                for i, sample in enumerate(modalities):
                    last = sample[-1]
                    last = (0, last) if torch.is_tensor(last) and last.is_floating_point() else last
                    assert isinstance(last, tuple), 'decoding a modality: every sample must end in the modality being decoded'
                    ty, x = int(last[0]), last[1]
                    if ty in self._ext:                                    # the user's encoder makes the block's token rows (encoder layout in)
                        x = x.to(dev, torch.float32)
                        pre = self.latent_to_model_projs[ty]
                        x = pre(x[None])[0] if self.channel_first_latent[ty] else pre(x)
                    assert int(np.prod(x.shape[:-1])) == L, '`decode_length` must be the number of tokens of the trailing modality'
                    tys.append((ty, x))
                    tok_inst[i * L:(i + 1) * L] = i
                kve = np.full(T, n_cached + L, np.int32)                   # own prefix + the whole (bidirectional) block
                rot = np.full(T, seen, np.int32)                           # all tokens of the instance share one rotary position (T:3206)
                row_tok = {t: np.full(T, -1, np.int32) for t in range(self.num_modalities)}
                for t in plan.ext_add:
                    plan.lat[t]['add'].zero_()
                for i, (ty, x) in enumerate(tys):
                    row_tok[ty][i * L:(i + 1) * L] = np.arange(i * L, (i + 1) * L)
                    plan.lat[ty]['tok' if ty in plan.ext else 'x'][i * L:(i + 1) * L].copy_(x.reshape(L, -1).to(dev, torch.float32))
                    if ty in plan.ext_add:                                # the decoded block's positional embedding (T:3179-3180 under the decode slice T:1167-1176)
                        plan.lat[ty]['add'][i * L:(i + 1) * L].copy_(self._pos_rows(ty, [tuple(x.shape[:-1])]))
                up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                for t in row_tok:
                    plan.row_tok[t].copy_(up(row_tok[t])); plan.row_src[t].copy_(up(np.maximum(row_tok[t], 0)))
                    plan.row_inst[t].copy_(up(np.repeat(np.arange(b, dtype=np.int32), L)))
                    if t not in plan.ext:
                        plan.set_noise(t, None)
                assert times is not None, 'decoding a modality needs `times` (the ODE step time in the last column, T:1990-1996)'
                plan.inst_time.copy_(times.to(dev, torch.float32).reshape(b, -1)[:, -1])
            smp._load(plan, ids=ids, pos=pos, kve=kve, rot=rot, tok_inst=tok_inst)
            Plan.run(plan.fwd, stream, 0, plan.fwd_embed_end if return_embed else plan.fwd_logits_end)
            n = L
            if return_embed:
                scan_in = modalities
                if self._ext:                                                  # the closures reshape to the PROJECTED axial shapes (MP:738-741)
                    with torch.no_grad():
                        scan_in, _ = self._ext_preprocess(modalities, None, return_loss=False)
                P = self._scan(scan_in, add_sos_eos=False, add_meta=False)
                out = (plan.embed.view(b, L, md.dim).float(),
                       self._pred_flow_closures(P, self._clean_sources(modalities, times.to(dev, torch.float32).reshape(b, -1) if times is not None else None) if md.model_output_clean else None))
            else:
                out = plan.logits.view(b, L, md.vp)[..., :md.vocab].clone()
            kv = None
            buf._tfx_filled = n_cached + L
            if return_kv_cache:
                kv = (self._kv_public(buf, n_cached + L), seen + 1 if is_mod else seen + L)
        ret = (out,)
        if return_kv_cache:
            ret = (*ret, kv)
        if return_hiddens:                                                     # hiddens = [x_0 .. x_depth, final norm output] (T:1199, T:1244, T:1253)
            nn_ = plan.T // b                                                     # the plan's (bucketed) row length; only the n real tokens go out
            ret = (*ret, [plan.hid[i].view(b, nn_, md.dim)[:, :n].float() for i in range(md.depth + 1)] + [plan.embed.view(b, nn_, md.dim)[:, :n].float()])
        if return_times:
            ret = (*ret, times)
        return ret[0] if len(ret) == 1 else ret

    # ------------------------------------------------------------------ pure-text LM path (T:2586-2664)
    def forward_text(self, text, return_loss=True, return_embed=False, cache=None, return_hiddens=False, return_kv_cache=False):
        """`Transfusion.forward_text`: the hot path with zero modalities - causal mask, no conditioning, cross entropy over the
        text-only part of the vocabulary (`text_only_logits_mask`, T:1509, T:2653).  Inference calls also take / return the kv cache
        `(tensor, tokens_seen)` and the hiddens (T:2612-2645): with a cache, `text` holds the NEW tokens only (rotary positions from tokens_seen)."""
        self._require_gpu()
        if (cache is not None or return_hiddens or return_kv_cache) and return_loss:
            raise NotImplementedError('forward_text: kv cache / hiddens are returned by the inference call only (return_loss = False)')
        dev, stream, md = self.device, self._stream(), self.md
        text = text.to(dev)
        if cache is not None:
            return self._forward_text_cached(text, cache, return_embed, return_kv_cache, return_hiddens)
        inp, labels = (text[:, :-1], text[:, 1:]) if return_loss else (text, None)          # T:2603-2604
        b, n = inp.shape
        key = ('text', b, n)
        S = self._struct_cache.get(key)
        if S is None:
            ar = torch.arange(n, dtype=torch.int32)
            tok_inst = np.full((b, n), -1, dtype=np.int32)
            seg_start, seg_len = token_segments(tok_inst, balance=True)
            D = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev)
            S = self._struct_cache[key] = dict(tok_inst=D(tok_inst.reshape(-1)), kv_end=D((ar + 1).repeat(b)), q_start=D(ar.repeat(b)),
                                               rot_pos=D(ar.repeat(b)), seg_start=D(seg_start), seg_len=D(seg_len))
        self.store.refresh_shadows(stream)
        plan = self._plan(b, n, 0, {}, training=return_loss)
        if plan.loaded_structure is not S:
            plan.set_rope_tables(*self._rope_tables(n))
            plan.tok_inst.copy_(S['tok_inst']); plan.kv_end.copy_(S['kv_end']); plan.q_start.copy_(S['q_start']); plan.rot_pos.copy_(S['rot_pos'])
            plan.set_segments(S['seg_start'], S['seg_len'])
            plan.loaded_structure = S
        plan.text_ids.copy_(inp.masked_fill(inp == -1, 0).reshape(-1))                      # T:2608
        if not return_loss:
            Plan.run(plan.fwd, stream, 0, plan.fwd_embed_end if return_embed else plan.fwd_logits_end)
            out = plan.embed.view(b, n, md.dim).float() if return_embed else plan.logits.view(b, n, md.vp)[..., :md.vocab].clone()
            ret = (out,)
            if return_kv_cache:                                                            # (tensor, tokens_seen + seq_len), T:2631
                from .sampling import Sampler
                buf = torch.zeros(md.depth, b, -(-(n + 64) // 64) * 64, 2 * md.hdk, device=dev, dtype=torch.bfloat16)
                Sampler(self)._fill_cache(buf, plan, b, n)
                buf._tfx_filled = n
                ret = (*ret, (self._kv_public(buf, n), n))
            if return_hiddens:
                ret = (*ret, [plan.hid[i].view(b, n, md.dim).float() for i in range(md.depth + 1)] + [plan.embed.view(b, n, md.dim).float()])
            return ret[0] if len(ret) == 1 else ret
        lab = labels.reshape(-1)
        plan.labels.copy_(torch.where(lab == self.ignore_index, torch.full_like(lab, -1), lab))
        plan.set_loss_scales(1.0, {})                        # mean over the valid labels: the count lives on the device (see _bwd_scale)
        plan.set_ce_vocab(self.num_text_tokens)              # text_only_logits_mask (T:2653)
        plan.acc.zero_()
        Plan.run(plan.fwd, stream)
        cnt = plan.acc[1].clamp(min=1.)
        loss = plan.acc[0] / cnt                             # T:2655-2659
        self._bwd_scale = (1. / cnt).detach()
        self._step_id += 1
        self._live = (plan, self._step_id)
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros((), device=dev, requires_grad=True)
            loss = _NativeLoss.apply(self._anchor, self, loss, _NO_ROWS)[0]
        return loss

    def _forward_text_cached(self, text, cache, return_embed, return_kv_cache, return_hiddens):
        """forward_text against a kv cache (T:2612-2631): the L new tokens of every row run as ONE causal multi-token decode step - token j at cache
        position n_cached + j and rotary position tokens_seen + j, attending to the cache and to the new tokens up to itself"""
        from .sampling import Sampler
        md, dev, stream = self.md, self.device, self._stream()
        kv_in, seen = cache
        b, L = text.shape
        buf, n_cached = self._kv_native(kv_in, L)
        assert buf.shape[1] == b, 'kv cache batch does not match the number of rows'
        if not hasattr(self, '_decode_plans') or len(self._decode_plans) > 8:
            self._decode_plans = {}
        smp = Sampler(self)
        plan = smp._decode_plan(('fwd_text', L, buf.data_ptr()), b, L, buf, False)
        ar = np.arange(L, dtype=np.int32)
        ids = text.masked_fill(text == -1, 0).reshape(-1).to(torch.int32).cpu().numpy()      # T:2608
        pos = (np.arange(b, dtype=np.int32)[:, None] * buf.shape[2] + n_cached + ar[None, :]).reshape(-1)
        kve = np.tile(n_cached + 1 + ar, b)
        rot = np.tile(int(seen) + ar, b)
        smp._load(plan, ids=ids, pos=pos, kve=kve, rot=rot, tok_inst=np.full(b * L, -1, np.int32))
        Plan.run(plan.fwd, stream, 0, plan.fwd_embed_end if return_embed else plan.fwd_logits_end)
        out = plan.embed.view(b, L, md.dim).float() if return_embed else plan.logits.view(b, L, md.vp)[..., :md.vocab].clone()
        buf._tfx_filled = n_cached + L
        ret = (out,)
        if return_kv_cache:
            ret = (*ret, (self._kv_public(buf, n_cached + L), int(seen) + L))
        if return_hiddens:
            ret = (*ret, [plan.hid[i].view(b, L, md.dim).float() for i in range(md.depth + 1)] + [plan.embed.view(b, L, md.dim).float()])
        return ret[0] if len(ret) == 1 else ret

    # ------------------------------------------------------------------ pure flow path (T:2710-2869)
    def forward_modality(self, modalities, times=None, modality_type=None, encode_modality=True, velocity_consistency_ema_model=None,
                         velocity_consistency_delta_time=1e-5, return_loss=True, return_loss_breakdown=False):
        """`Transfusion.forward_modality`: every token is a latent of ONE instance per sample (`modality_only=True`): conditioned on
        the sample's time, no attention mask, no rotary embedding; loss = MSE(pred flow, x - noise).  Encoders / EMA velocity
        consistency / reconstruction loss are outside the native path."""
        self._require_gpu()
        ema = velocity_consistency_ema_model
        if ema is not None and hasattr(ema, 'ema_model'):
            ema = ema.ema_model
        if ema is not None and not isinstance(ema, Transfusion):
            raise NotImplementedError('velocity_consistency_ema_model must be a native Transfusion (or the EMA wrapper of one)')
        if ema is self:
            raise ValueError('velocity_consistency_ema_model is the model itself: the teacher pass would overwrite the activations the backward needs - pass an EMA copy (create_ema)')
        if self.num_modalities > 1 and modality_type is None:
            raise AssertionError('`modality_type` must be explicitly passed in on forward when training on greater than 1 modality')
        t = 0 if modality_type is None else int(modality_type)
        dev, stream, md = self.device, self._stream(), self.md
        modalities = modalities.to(dev)
        orig_modalities = modalities                                                       # the reconstruction target (T:2722, T:2850-2853)
        if encode_modality and self.modality_encoder[t] is not None:                       # T:2735-2738
            with torch.no_grad():
                self.modality_encoder[t].eval()
                modalities = self.modality_encoder[t](modalities).detach()
        ext = t in self._ext
        b = modalities.shape[0]
        if times is None:
            times = torch.rand(b, device=dev)                                              # T:2746-2747
        times = times.to(dev, torch.float32).reshape(b)
        if ema is not None and return_loss:                                                # T:2752-2755
            orig_times = times.clone()
            times = times * (1. - velocity_consistency_delta_time)
        dl, d = md.dim_latents[t], md.dim
        tok = flow_ext = None
        if ext:
            # the user's encoder / decoder replace latent_to_model / model_to_latent (T:1493-1494): noising, encoder, decoder and the loss are
            # PyTorch; the transformer over the encoder's tokens (and its backward) is the native engine
            raw = modalities.to(torch.float32)
            if return_loss:
                tt = times.view(b, *([1] * (raw.ndim - 1)))
                eps = self._noise_override[t].to(dev, torch.float32).view(raw.shape) if self._noise_override is not None else torch.randn_like(raw)
                noised, flow_ext = tt * raw + (1. - tt) * eps, raw - eps                   # T:2756-2762
            else:
                noised = raw
            tok = self.latent_to_model_projs[t](noised)                                     # (b, *axial', dim)
            assert tok.shape[-1] == d, f'the encoder of modality {t} must produce model-dimension ({d}) tokens'
            axial = tuple(tok.shape[1:-1])
        else:
            if self.channel_first_latent[t]:
                modalities = modalities.movedim(1, -1)                                     # (b, d, *axial) -> (b, *axial, d)
            x = modalities.to(dev, torch.float32)
            assert x.shape[-1] == dl, f'last dimension must be dim_latent = {dl}'
            axial = tuple(x.shape[1:-1])
        L = int(np.prod(axial)) if axial else 1
        rows = b * L
        key = ('modality', b, L, t)
        S = self._struct_cache.get(key)
        if S is None:
            inst = torch.arange(b, dtype=torch.int32).repeat_interleave(L)
            S = self._struct_cache[key] = dict(tok_inst=inst.to(dev), kv_end=torch.full((rows,), L, dtype=torch.int32, device=dev),
                                               zeros=torch.zeros(rows, dtype=torch.int32, device=dev),
                                               row_tok=torch.arange(rows, dtype=torch.int32, device=dev),
                                               empty=torch.zeros(0, dtype=torch.int32, device=dev))
        self.store.refresh_shadows(stream)
        plan = self._plan(b, L, b, {t: rows}, training=return_loss)
        if md.model_output_clean:
            plan.set_clean_mode('latent')                   # pure flow path: the latent-space conversion (T:2772-2810)
        if plan.loaded_structure is not S:
            plan.set_rope_tables(*self._rope_tables(0))
            plan.tok_inst.copy_(S['tok_inst']); plan.kv_end.copy_(S['kv_end']); plan.q_start.copy_(S['zeros']); plan.rot_pos.copy_(S['zeros'])
            plan.set_segments(S['empty'], S['empty'])             # one instance spans a whole row: per-token atomics for the instance gradients
            plan.row_tok[t].copy_(S['row_tok']); plan.row_inst[t].copy_(S['tok_inst'])
            if return_loss:
                plan.set_rows({t: rows})                                            # (fills the gather form of the row map)
            plan.text_ids.zero_()
            plan.loaded_structure = S
        plan.inst_time.copy_(times)
        lt = plan.lat[t]
        rows_in = []
        if t in plan.ext_add:                                                              # T:2781-2796: the same (L, dim) embedding for every sample
            assert len(axial) == self.pos_emb_mlp[t].num_axial_dims, f'received modalities of ndim {len(axial)} but expected {self.pos_emb_mlp[t].num_axial_dims}'
            pos = self._pos_rows(t, [axial]).repeat(b, 1)
            lt['add'].copy_(pos.detach())
            rows_in.append(('add', t, pos))
        if ext:
            tok_rows = tok.reshape(rows, d)
            lt['tok'].copy_(tok_rows.detach())
            rows_in.append(('tok', t, tok_rows))
        else:
            lt['x'].copy_(x.reshape(rows, dl))
        if not return_loss:
            if ext:
                Plan.run(plan.fwd, stream, 0, plan.fwd_embed_end)
                out = self.model_to_latent_projs[t](plan.embed.view(b, *axial, d).float())
                if md.model_output_clean:                                             # T:2770-2771, T:2810: the latent-space conversion around the user's decoder
                    out = (out - noised) / (1. - times.view(b, *([1] * (raw.ndim - 1)))).clamp_min(md.clean_eps)
                return out
            plan.set_noise(t, None)                                                  # T:2759-2760: no noising
            Plan.run(plan.fwd, stream, 0, plan.fwd_pred_end)
            out = lt['pred'].view(x.shape).clone()
            return out.movedim(-1, 1) if self.channel_first_latent[t] else out
        if not ext:
            if self._noise_override is not None:
                lt['eps'].copy_(self._noise_override[t].reshape(rows, dl))
            else:
                lt['eps'].normal_()                                                            # T:2753
            plan.set_noise(t, lt['eps'].data_ptr())
        plan.labels.fill_(-1)
        plan.set_loss_scales(0.0, {} if ext else {t: 2.0 / (rows * dl)})
        plan.set_ce_vocab(md.vocab)
        self._bwd_scale = None
        plan.acc.zero_()
        Plan.run(plan.fwd, stream, graph='auto')
        flow_loss = torch.zeros((), device=dev) if ext else plan.acc[2 + t] / (rows * dl)   # T:2817
        loss = flow_loss
        velocity_loss = None
        if ema is not None and ext:
            # same term for a type whose maps are user modules: the teacher's flow comes back in the raw layout, the mse is PyTorch's
            was_training = ema.training
            ema.eval()
            try:
                with torch.no_grad():
                    teacher = ema.forward_modality(raw, times=orig_times + velocity_consistency_delta_time, modality_type=t, encode_modality=False, return_loss=False)
                    velocity_loss = torch.nn.functional.mse_loss(flow_ext, teacher)
            finally:
                ema.train(was_training)
            loss = loss + self.velocity_consistency_loss_weight * velocity_loss
        elif ema is not None:
            # T:2823-2836: mse(FLOW TARGET, teacher flow at t + delta on the clean input) - no gradient reaches the student through it
            was_training = ema.training
            ema.eval()
            try:
                with torch.no_grad():
                    xin = x.view(modalities.shape)
                    teacher = ema.forward_modality(xin.movedim(-1, 1) if self.channel_first_latent[t] else xin, times=orig_times + velocity_consistency_delta_time,
                                                   modality_type=t, encode_modality=False, return_loss=False)
                    if self.channel_first_latent[t]:
                        teacher = teacher.movedim(1, -1)
            finally:
                ema.train(was_training)
            lt['vel'].copy_(teacher.reshape(rows, dl))
            va = plan._vel_args[t]
            va.pred, va.flow, va.grad_scale = lt['vel'].data_ptr(), lt['flow'].data_ptr(), 0.0
            Plan.run(plan.vel, stream)
            velocity_loss = plan.acc[2 + self.num_modalities + t] / (rows * dl)
            loss = loss + self.velocity_consistency_loss_weight * velocity_loss              # T:2858-2862
        recon_loss = None
        dec = self.modality_decoder[t]
        if self.has_recon_loss:                                                            # T:2840-2853
            assert encode_modality, 'the reconstruction loss needs the un-encoded modality as its target'
        if self.has_recon_loss and not ext:
            if dec is None:
                # recon = noise + pred (1 - t) against the clean latent: the native residual kernel (mode 1), gradient into the same seeds
                if self.modality_encoder[t] is not None:
                    raise ValueError('reconstruction loss with a modality encoder needs the matching modality decoder')
                lt['recw'].fill_(1. / rows)
                ra = plan._rec_args[t]
                ra.recon_mode, ra.grad_scale = 1, 2.0 * self.reconstruction_loss_weight / dl
                Plan.run(plan.rec, stream)
                recon_loss = plan.acc[2 + 2 * self.num_modalities + t] / dl
            else:
                # through the frozen decoder, without gradients (T:2845-2848): a reported term
                with torch.no_grad():
                    tt = times.view(b, *([1] * (x.ndim - 1)))
                    rec = lt['eps'].view(x.shape) + lt['pred'].view(x.shape) * (1. - tt)
                    dec.eval()
                    recon_loss = torch.nn.functional.mse_loss(dec(rec.movedim(-1, 1) if self.channel_first_latent[t] else rec), orig_modalities.float())
            loss = loss + self.reconstruction_loss_weight * recon_loss
        self._step_id += 1
        self._live = (plan, self._step_id)
        emb_rows = None
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != dev:
                self._anchor = torch.zeros((), device=dev, requires_grad=True)
            spec = {'in': [(k, ty) for k, ty, _ in rows_in], 'out': [t] if ext else []} if rows_in else _NO_ROWS
            loss, *emb_rows = _NativeLoss.apply(self._anchor, self, loss, spec, *[r for _, _, r in rows_in])
        elif ext:
            emb_rows = [plan.embed.float()]
        if ext:                                                                            # the user's decoder and the flow loss, in PyTorch (T:2806-2817)
            pred = self.model_to_latent_projs[t](emb_rows[0].view(b, *axial, d))
            if md.model_output_clean:                                                   # T:2770-2771, T:2810
                pred = (pred - noised) / (1. - tt).clamp_min(md.clean_eps)
            flow_loss = torch.nn.functional.mse_loss(pred, flow_ext)
            loss = loss + flow_loss
            if self.has_recon_loss:
                rec = eps + pred * (1. - tt)
                if dec is not None:
                    with torch.no_grad():
                        dec.eval()
                        rec = dec(rec)
                recon_loss = torch.nn.functional.mse_loss(rec, orig_modalities.float())
                loss = loss + self.reconstruction_loss_weight * recon_loss
        if not return_loss_breakdown:
            return loss
        zero = torch.zeros((), device=dev)
        return loss, (flow_loss.detach(), velocity_loss.detach() if velocity_loss is not None else zero, recon_loss.detach() if recon_loss is not None else zero)

    @torch.no_grad()
    def generate_modality_only(self, batch_size=1, modality_type=None, fixed_modality_shape=None, modality_steps=16,
                               return_unprocessed_modalities=False):
        """T:2871-2923: fixed-grid midpoint ODE from noise to a sample, `modality_steps` grid points on [0, 1]."""
        self._require_gpu()
        if self.num_modalities > 1 and modality_type is None:
            raise AssertionError('`modality_type` must be explicitly passed in on forward when training on greater than 1 modality')
        t = 0 if modality_type is None else int(modality_type)
        shape = fixed_modality_shape if fixed_modality_shape is not None else self.modality_default_shape[t]
        assert shape is not None
        dev = self.device
        y = self._gen_noise_override.to(dev, torch.float32).clone() if getattr(self, '_gen_noise_override', None) is not None \
            else torch.randn((batch_size, *shape, self.md.dim_latents[t]), device=dev)
        if self.channel_first_latent[t]:
            y = y.movedim(-1, 1).contiguous()                                               # T:2892-2893
        was_training = self.training
        self.eval()
        try:
            grid = torch.linspace(0., 1., modality_steps, device=dev)
            f = lambda tt, yy: self.forward_modality(yy, times=tt.expand(batch_size), modality_type=t, encode_modality=False, return_loss=False)
            for i in range(modality_steps - 1):                                            # torchdiffeq fixed-grid 'midpoint'
                t0, dt = grid[i], grid[i + 1] - grid[i]
                y_mid = y + f(t0, y) * (dt * 0.5)
                y = y + dt * f(t0 + dt * 0.5, y_mid)
        finally:
            self.train(was_training)
        if self.modality_decoder[t] is not None:                                           # T:2917-2921
            self.modality_decoder[t].eval()
            y = self.modality_decoder[t](y)
        return y

    def _native_backward(self, grad_out, step_id):
        plan, live_id = self._live
        if live_id != step_id:
            raise RuntimeError('backward() of a stale loss: the native engine keeps the activations of the latest forward only')
        if self._consumed == step_id:
            # the loss seeds (d loss / d logits, d loss / d pred flows) are scaled in place by the upstream gradient below and the
            # backward scratch is reused: a second pass would silently double-scale and double-accumulate - refuse, like autograd
            # does for a freed graph
            raise RuntimeError('backward() through the native Transfusion step a second time: the engine keeps one set of loss seeds per '
                               'forward (call forward again; retain_graph is not supported)')
        self._consumed = step_id
        ps = self.store
        ps.ensure_grad_views()
        go = grad_out.reshape(()).to(torch.float32)           # d(total)/d(loss): scales the loss seeds (everything downstream is linear)
        if self._bwd_scale is not None:
            go = go * self._bwd_scale                         # forward_text: 1 / #valid labels, known on the device only
        lib, stream = capi.lib(), self._stream()
        sp = ctypes.c_void_p(stream)
        # fp32 scalar read on the device; exactly 1.0 (plain loss.backward()) skips the pass over the seeds
        for seed in [plan.dlogits] + [lt['dpred'] for lt in plan.lat.values() if 'dpred' in lt]:
            capi.check(lib.tfx_scale_bf16_dev(seed.data_ptr(), seed.numel(), go.data_ptr(), sp), 'tfx_scale_bf16_dev')
        plan.dtables.zero_()
        if self.md.model_output_clean and plan.clean_mode == 'model' and len(plan.clean_bwd):
            Plan.run(plan.clean_bwd, stream)                # gradient paths through q = W proj (engine._clean_model_space)
        red = getattr(self, '_grad_reducer', None)
        if red is None or red.defer or not plan.bwd_cuts or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            if red is not None and red.defer:
                red.check_fresh()                               # (accumulating AFTER a backward that already exchanged would add into summed ranges)
            Plan.run(plan.bwd, stream, graph='auto')            # `opt.no_sync()`: accumulate only - the step's last backward exchanges the sums
            return
        # data parallel with overlap: replay the list group by group; a finished group's gradient ranges go out while the rest runs
        red.check_fresh()                                   # one backward per optimizer step (the groups of the previous one are already summed over the ranks)
        red.begin()
        lo = 0
        for idx, first, last in plan.bwd_cuts:
            Plan.run(plan.bwd, stream, lo, idx, graph='auto')
            red.group_ready(first, last)
            lo = idx
        Plan.run(plan.bwd, stream, lo, None, graph='auto')

    # ------------------------------------------------------------------ sampling surface (T:1842-2583)
    @torch.no_grad()
    def sample_many(self, prompts=None, max_length=2048, text_temperature=1., text_min_p=0.1, fixed_modality_shape=None,
                    force_modality_at_start=None, init_modality_noise=None, modality_steps=16, return_unprocessed_modalities=False,
                    cfg_scale=3., _pos_emb_in_decode=False):
        """T:2082-2583.  `_pos_emb_in_decode`: with `add_pos_emb`, the reference's two samplers differ - `sample_many` feeds the ODE evaluations
        the bare projected latents (project_to_model, T:2436-2444 / T:2472-2475: no positional embedding), while `sample_one`, going through
        `forward()`, adds it (T:3179-3180).  Each entry point keeps its own reference's behaviour; prompted modalities carry it in both (T:2194)."""
        from .sampling import Sampler
        was_training = self.training
        self.eval()                                              # @temp_eval in the reference
        try:
            if self._ext:
                # `pre_post_transformer_enc_dec` types: a conv / U-Net encoder may change the token count (stride 2 in the reference's examples), which
                # only the reference's UN-CACHED `sample_one` supports (its cached paths slice `modality_length` latent tokens, T:2004 / T:2441) -
                # so these models decode prompt by prompt through that loop, written against forward() (one full forward per token / ODE evaluation,
                # lengths bucketed to 64).  Text and past modalities see the [som] tokens and t = 1 re-encodings exactly as there.
                plist = prompts if isinstance(prompts, list) else [prompts]
                outs = [self._sample_one_through_forward(pr, max_length=max_length, text_temperature=text_temperature, text_min_p=text_min_p, cache_kv=False,
                                                         fixed_modality_shape=fixed_modality_shape, force_modality_at_start=force_modality_at_start,
                                                         init_modality_noise=init_modality_noise, modality_steps=modality_steps, cfg_scale=cfg_scale)
                        for pr in plist]
                return outs if return_unprocessed_modalities else self.decode_modalities(outs)
            special = any(self.channel_first_latent) or any(e is not None for e in self.modality_encoder) or any(d is not None for d in self.modality_decoder)
            if special and prompts is not None:
                prompts = [self._normalize_prompt(pr) for pr in (prompts if isinstance(prompts, list) else [prompts])]
            out = Sampler(self).sample_many(prompts, max_length=max_length, text_temperature=text_temperature, text_min_p=text_min_p,
                                            fixed_modality_shape=fixed_modality_shape, force_modality_at_start=force_modality_at_start,
                                            init_modality_noise=init_modality_noise, modality_steps=modality_steps, cfg_scale=cfg_scale,
                                            pos_emb_in_decode=_pos_emb_in_decode)
            if special:
                out = [[(p[0], self._from_channel_last(p[0], p[1])) if isinstance(p, tuple) else p for p in sample] for sample in out]
                if not return_unprocessed_modalities:
                    out = self.decode_modalities(out)                                       # T:2581-2583
            return out
        finally:
            self.train(was_training)

    def _normalize_prompt(self, pr):
        """a prompt's modalities through their frozen encoder (prepare_prompt_sample, T:1755-1758) and into the decoder's channel-last layout
        (EVERY channel-first type, also those whose encoder is a user module)"""
        if pr is None:
            return None
        single = isinstance(pr, tuple) or torch.is_tensor(pr)
        parts = self._encode_modalities([[pr] if single else list(pr)])[0]
        out = []
        for part in parts:
            if torch.is_tensor(part) and part.is_floating_point():
                part = (0, part)
            if isinstance(part, tuple) and self.channel_first_latent[part[0]]:
                part = (part[0], part[1].movedim(0, -1))
            out.append(part)
        return out[0] if single else out

    @torch.no_grad()
    def sample_one(self, prompt=None, max_length=2048, text_temperature=1., text_min_p=0.1, cache_kv=None, fixed_modality_shape=None,
                   force_modality_at_start=None, init_modality_noise=None, modality_steps=16, return_unprocessed_modalities=False, cfg_scale=3.):
        """T:1845-1858.  The reference asserts sample_many == per-prompt sample_one (tests/test_transfusion.py:758-808); here
        sample_one IS the batch-of-one case of the KV-cached decoder.
        `cache_kv`: None (the default here; the reference's default is False) and True run that decoder - it is always KV-cached.  An EXPLICIT
        False runs the reference's un-cached loop instead (T:1858-2075 written against `forward()`, `_sample_one_through_forward`: one full
        forward per text token / ODE evaluation, no cache) - the arithmetic the reference's cache-equivalence tests compare the cached path with
        (not for `model_output_clean` models: the decode contract of `forward()` raises for them, they keep the decoder)."""
        if cache_kv is False and not self._ext and not self.model_output_clean:       # (`forward()`'s decode contract has no model_output_clean conversion: those models keep the decoder)
            was_training = self.training
            self.eval()
            try:
                out = self._sample_one_through_forward(prompt, max_length=max_length, text_temperature=text_temperature, text_min_p=text_min_p, cache_kv=False,
                                                       fixed_modality_shape=fixed_modality_shape, force_modality_at_start=force_modality_at_start,
                                                       init_modality_noise=init_modality_noise, modality_steps=modality_steps, cfg_scale=cfg_scale)
                return out if return_unprocessed_modalities else self.decode_modalities(out)
            finally:
                self.train(was_training)
        return self.sample_many([prompt], max_length=max_length, text_temperature=text_temperature, text_min_p=text_min_p,
                                fixed_modality_shape=fixed_modality_shape, force_modality_at_start=force_modality_at_start,
                                init_modality_noise=init_modality_noise, modality_steps=modality_steps,
                                return_unprocessed_modalities=return_unprocessed_modalities, cfg_scale=cfg_scale, _pos_emb_in_decode=True)[0]

    sample = sample_one

    @torch.no_grad()
    def _sample_one_through_forward(self, prompt=None, max_length=2048, text_temperature=1., text_min_p=0.1, cache_kv=False, fixed_modality_shape=None,
                                    force_modality_at_start=None, init_modality_noise=None, modality_steps=16, cfg_scale=3.):
        """The reference's `sample_one` loop (T:1858-2075) written against the PUBLIC decode contract of `forward()` - one `forward(...)` per
        text token (T:1917-1924) and per ODE evaluation (T:1998-2006, T:2021-2029), with (`cache_kv=True`) or without the kv cache it returns.
        `sample_one` itself runs the batched KV-cached decoder; this loop exists so that the contract has a caller (the reference's
        cache-equivalence tests go through it, tests/test_transfusion.py:578-662) - tests/test_decode_contract_gpu.py compares the three."""
        from .sampling import Sampler, _sample_text_token, _ode_axpy
        was_training = self.training
        self.eval()
        try:
            smp = Sampler(self)
            dev, stream = self.device, self._stream()
            parts, forced_id, forced_shape = smp._prepare(self._normalize_prompt(prompt), force_modality_at_start)
            # forward() takes the public layout: channel-first types as (dim_latent, *axial)
            sample = [((p[0], self._from_channel_last(p[0], p[1])) if isinstance(p, tuple) else torch.tensor(p, dtype=torch.long, device=dev)) for p in parts]
            curr_length = 0                                                            # counts DECODED tokens only (T:1878)
            num_past = sum(isinstance(p, tuple) for p in sample)
            cache = None
            st = type('S', (), {})()
            st.curr_seq, st.forced = parts[-1] if isinstance(parts[-1], list) else [self.sos_id], (forced_id, forced_shape)
            decoding_text = not smp._maybe_transition(st, fixed_modality_shape)
            while curr_length <= max_length:
                if decoding_text:
                    # (the reference leaves `times` to the random default here, T:1917-1924; prompted / decoded modalities are conditioned at 1
                    # everywhere else in its samplers - T:1996, T:2192 - and so here)
                    logits = self.forward([sample], return_loss=False, cache=cache, decode_length=1, decoding_text_or_modality='text',
                                          return_kv_cache=cache_kv, times=torch.ones(1, max(num_past, 1), device=dev))
                    logits, new_cache = logits if cache_kv else (logits, None)
                    tok = int(_sample_text_token(logits[0, -1:].float().contiguous(), self.md.vocab, text_temperature, text_min_p, stream)[0])
                    sample[-1] = torch.cat((sample[-1], torch.tensor([tok], device=dev)))
                    st.curr_seq = sample[-1].tolist()
                    curr_length += 1
                    if cache_kv:
                        cache = new_cache
                    if tok == self.eos_id:
                        break
                    decoding_text = not smp._maybe_transition(st, fixed_modality_shape)
                    continue
                ty, shape = st.curr_modality_id, st.modality_shape
                L, dl = st.modality_length, self.md.dim_latents[ty]
                y = (init_modality_noise[:L, :dl].to(dev, torch.float32) if init_modality_noise is not None else torch.randn(L, dl, device=dev)).reshape(*shape, dl)
                y = self._from_channel_last(ty, y).contiguous()                       # T:1963-1964
                use_cfg = cfg_scale != 1.
                uncond_hist = [torch.full_like(p, self.null_text_id) if not isinstance(p, tuple) else p for p in sample]
                uncond_cache = None
                if use_cfg and cache_kv:                                            # T:1976-1988
                    _, uncond_cache = self.forward([uncond_hist], return_loss=False, return_kv_cache=True, return_embed=True,
                                                   times=torch.ones(1, max(num_past, 1), device=dev), decoding_text_or_modality='modality')
                new_cache = None

                def flow(t, yy, hist, kv):
                    tt = torch.ones(1, num_past + 1, device=dev); tt[0, -1] = t          # past modalities are conditioned at time 1 (T:1996)
                    res = self.forward([[*hist, (ty, yy)]], times=tt, return_embed=True, cache=kv, decode_length=L, return_kv_cache=cache_kv,
                                       decoding_text_or_modality='modality')
                    (emb, fns), kv_out = res if cache_kv else (res, None)
                    fl = self.model_to_latent(ty, fns[ty][-1](emb, need_splice=kv is None))
                    if ty not in self._ext:                                           # the native projection returns (*axial, dim_latent)
                        fl = self._from_channel_last(ty, fl)
                    return fl.contiguous(), kv_out

                def velocity(t, yy):
                    nonlocal new_cache
                    fc, new_cache = flow(t, yy, sample, cache if cache_kv else None)
                    fu = flow(t, yy, uncond_hist, uncond_cache)[0] if use_cfg else None
                    return fc, fu

                ts = torch.linspace(0, 1, modality_steps)
                for k in range(modality_steps - 1):                                  # torchdiffeq fixed-grid midpoint
                    t0, dt = float(ts[k]), float(ts[k + 1] - ts[k])
                    fc, fu = velocity(t0, y)
                    y_mid = _ode_axpy(y.contiguous(), fc.contiguous(), fu, cfg_scale, dt * 0.5, stream)
                    fc, fu = velocity(t0 + dt * 0.5, y_mid)
                    y = _ode_axpy(y.contiguous(), fc.contiguous(), fu, cfg_scale, dt, stream)
                sample.append((ty, y))
                sample.append(torch.tensor([self.eom_ids[ty]], device=dev))
                st.curr_seq = [self.eom_ids[ty]]
                if cache_kv:
                    cache = new_cache
                curr_length += L
                num_past += 1
                decoding_text = True
            return sample
        finally:
            self.train(was_training)

    def _clone_architecture(self):
        """a fresh model with this one's constructor arguments on the same device (weights NOT copied)"""
        import copy
        kw = dict(self._init_kwargs)
        kw['transformer'] = self.transformer_config
        kw['pre_post_transformer_enc_dec'] = copy.deepcopy(kw.get('pre_post_transformer_enc_dec'))     # learnable: the copy owns its own modules
        m = Transfusion(**kw)
        return m.to(self.device) if self.device.type == 'cuda' else m

    def create_dataloader(self, *args, **kwargs):                   # T:1676-1679
        return create_dataloader(*args, **kwargs)

    def create_ema(self, beta=0.99, *ema_kwargs):                   # T:1681-1699
        from .ema import EMA
        return EMA(self, beta=beta, forward_method_names=('sample', 'sample_one', 'sample_many', 'generate_text_only', 'generate_modality_only'))

    @torch.no_grad()
    def generate_text_only(self, prompt, seq_len, temperature=1.0, min_p=0.1, cache_kv=True):
        """T:2666-2707.  Returns the generated ids (b, seq_len - prompt_len).  cache_kv=True: the KV-cached decoder (prefill + one-row decode
        steps); cache_kv=False: the reference's un-cached loop - `forward_text` over the whole sequence for every new token (T:2686-2698)."""
        from .sampling import Sampler, _pick_text_only
        was_training = self.training
        self.eval()
        try:
            if cache_kv:
                return Sampler(self).generate_text_only(prompt, seq_len, temperature, min_p)
            out = prompt.to(self.device)
            stream = self._stream()
            for _ in range(max(seq_len - prompt.shape[-1], 0)):
                logits = self.forward_text(out, return_loss=False)                         # (b, n, vocabulary)
                last = logits[:, -1].float().contiguous()
                tok = _pick_text_only(last, last.shape[-1], self.num_text_tokens, temperature, min_p, stream).to(out.dtype)
                out = torch.cat((out, tok.reshape(-1, 1)), dim=-1)
            return out[:, prompt.shape[-1]:]
        finally:
            self.train(was_training)


def print_modality_sample(modality_sample):           # T:224-239
    output = []
    for sample in modality_sample:
        if isinstance(sample, tuple):
            modality_type, sample = sample
            output.append((f'modality:{modality_type}', sample.shape))
        elif is_int_tensor(sample):
            output.append(('text', sample.shape))
        else:
            output.append(('modality', sample.shape))
    print(output)


def collate_fn(data):                                  # T:305-306
    return [*map(list, data)]


def create_dataloader(dataset, **kwargs):              # T:308-310
    from torch.utils.data import DataLoader
    return DataLoader(dataset, collate_fn=collate_fn, **kwargs)
