"""Parameter store for the native Transfusion path.

* every learnable tensor keeps the REFERENCE's `state_dict` name and shape (reference
  constructor T:1290-1540, Transformer T:1043-1097) so `load_state_dict` from a reference checkpoint is the
  weight-transfer mechanism;
* all of them are views into ONE flat fp32 master buffer (and one flat fp32 gradient buffer): a single
  RCCL all-reduce and a single fused Adam launch cover the whole model;
* the kernels read bf16 "shadows" in kernel-friendly layouts (K padded to 64, GEGLU value/gate rows
  interleaved in blocks of 32, transposed copies for the dX GEMMs), rebuilt from the master after every
  update by `refresh_shadows` (HIP cast kernels).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn

from . import capi


def pad_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class ModelDims:
    num_text_tokens: int
    dim: int
    depth: int
    heads: int
    dim_head: int
    dim_latents: tuple
    ff_expansion_factor: float = 4.
    model_output_clean: bool = False      # T:1297: the model predicts the clean latent; flows are derived (MP:100-126)
    clean_eps: float = 1e-2               # T:1319 `eps`: floor of (1 - t) in that conversion
    pos_types: tuple = ()                 # modality types whose tokens get an additive axial positional embedding (T:1384-1403): rows computed by the host MLP
    ext_types: tuple = ()                 # modality types whose latent <-> model maps are user modules (`pre_post_transformer_enc_dec`, T:1451-1494):
                                          # their token rows come in from / their embedding rows go out to PyTorch; no projection parameters here

    @property
    def num_modalities(self): return len(self.dim_latents)
    @property
    def hd(self): return self.heads * self.dim_head
    @property
    def di(self): return int(self.dim * self.ff_expansion_factor * 2 / 3)          # T:842
    @property
    def dip(self): return pad_to(self.di, 64)
    @property
    def vocab(self): return self.num_text_tokens + 3 + 2 * self.num_modalities + 129   # T:1503
    @property
    def vp(self): return pad_to(self.vocab, 64)
    @property
    def nq(self): return 3 * self.hd + self.heads            # q | k | v | gate logits (rows of the fused projection, T:877-916)
    @property
    def hdk(self): return self.heads * 64                     # KERNEL layout: 64 columns per head; dim_head < 64 leaves zero columns
    @property
    def nqk(self): return 3 * self.hdk + self.heads           # q | k | v | gate logits in the kernel layout
    @property
    def ldq(self): return pad_to(self.nqk, 64)
    @property
    def kf(self): return pad_to(self.dim + 1, 64)            # fourier embedding width, padded
    @property
    def nt3(self): return self.depth * 2 * 3 * self.dim      # all AdaLN tables: per (layer, wrapper) gamma|beta|z
    def has_skip(self, layer_index: int) -> bool:            # T:1081-1083 (0-based index)
        return layer_index >= self.depth / 2


def param_specs(md: ModelDims):
    """ordered (name, shape) list in FLAT-BUFFER order.  Groups that one GEMM treats as a single matrix are
    laid out contiguously: all AdaLN conditioning weights ([depth*2*3d, 4d]) and, per layer, to_qk|to_v|to_gates."""
    d, hd, h, di, D = md.dim, md.hd, md.heads, md.di, md.depth
    specs = []
    for i in range(D):
        for w in (1, 2):
            p = f'transformer.layers.{i}.{w}'
            specs += [(f'{p}.to_film.weight', (2 * d, 4 * d)), (f'{p}.to_ada_ln_zero.weight', (d, 4 * d))]
    for i in range(D):
        for w in (1, 2):
            p = f'transformer.layers.{i}.{w}'
            specs += [(f'{p}.to_film.bias', (2 * d,)), (f'{p}.to_ada_ln_zero.bias', (d,))]
    specs += [('transformer.to_time_cond.1.weight', (4 * d, d + 1)), ('transformer.to_time_cond.1.bias', (4 * d,))]
    for i in range(D):
        p = f'transformer.layers.{i}'
        if md.has_skip(i):
            specs.append((f'{p}.0.weight', (d, 2 * d)))
        specs += [(f'{p}.1.fn.to_qk.0.weight', (2 * hd, d)), (f'{p}.1.fn.to_v.0.weight', (hd, d)), (f'{p}.1.fn.to_gates.0.weight', (h, d))]
        specs += [(f'{p}.1.fn.to_out.1.weight', (d, hd)),
                  (f'{p}.1.fn.q_norm.gamma', (md.dim_head,)), (f'{p}.1.fn.k_norm.gamma', (md.dim_head,)),
                  (f'{p}.1.layernorm_gamma', (d,)), (f'{p}.1.layerscale', (d,)),
                  (f'{p}.2.layernorm_gamma', (d,)), (f'{p}.2.layerscale', (d,)),
                  (f'{p}.2.fn.net.0.weight', (2 * di, d)), (f'{p}.2.fn.net.0.bias', (2 * di,)),
                  (f'{p}.2.fn.net.3.weight', (d, di)), (f'{p}.2.fn.net.3.bias', (d,))]
    specs.append(('transformer.norm.gamma', (d,)))
    # the AttentionResidual parameters sit behind the layer blocks: layer j's AttentionResidual mixes every earlier hidden, so in the pull-form
    # backward (engine.Plan, tfx_attnres_pull_bwd) its gradient keeps receiving terms until the backward reaches hidden 0 - it is final with the
    # embeddings, not with its layer, and travels in the tail of the overlapped gradient exchange (optim.GradReducer)
    for i in range(D):
        specs += [(f'transformer.layers.{i}.3.pseudo_queries', (d,)), (f'transformer.layers.{i}.3.norm_keys.gamma', (d,))]
    for t, dl in enumerate(md.dim_latents):
        if t in md.ext_types:
            continue
        if dl != d:                                                                 # T:1478
            specs += [(f'latent_to_model_projs.{t}.weight', (d, dl)), (f'latent_to_model_projs.{t}.bias', (d,))]
        specs.append((f'model_to_latent_projs.{t}.weight', (dl, d)))
    specs += [('text_embed.weight', (md.vocab, d)), ('to_text_logits.weight', (md.vocab, d))]
    return specs


def init_param_(name: str, t: torch.Tensor):
    """the reference's initialisers (nn.Linear / nn.Embedding defaults, zeros, T:659-669, T:783, T:803)."""
    if name.endswith(('to_film.weight', 'to_ada_ln_zero.weight', 'gamma', 'layernorm_gamma', 'layerscale')):
        t.zero_()
    elif name.endswith('to_ada_ln_zero.bias'):
        t.fill_(-2.)
    elif name.endswith('pseudo_queries'):
        t.normal_(std=0.02)
    elif name == 'text_embed.weight':
        t.normal_()
    elif name.endswith('weight'):
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    elif name.endswith('bias'):
        # nn.Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in recovered by the caller via attribute
        bound = getattr(t, '_fan_in_bound', 0.02)
        t.uniform_(-bound, bound)
    else:
        raise KeyError(name)


class _Holder(nn.Module):
    """structure-only module: holds parameters under the reference's attribute names."""


def _attach(root: nn.Module, name: str, value, buffer=False):
    parts = name.split('.')
    m = root
    for i, part in enumerate(parts[:-1]):
        nxt_is_index = parts[i + 1].isdigit()
        if part.isdigit():
            idx = int(part)
            while len(m) <= idx:
                m.append(None)
            if m[idx] is None:
                m[idx] = nn.ModuleList() if nxt_is_index else _Holder()
            m = m[idx]
        else:
            if not hasattr(m, part):
                setattr(m, part, nn.ModuleList() if nxt_is_index else _Holder())
            m = getattr(m, part)
    leaf = parts[-1]
    if buffer:
        m.register_buffer(leaf, value)
    else:
        m.register_parameter(leaf, value)


def geglu_phys_to_ref_rows(di: int, dip: int) -> np.ndarray:
    """row of net.0.weight ([value rows 0..di | gate rows di..2di], T:831-834) for each physical row of the
    interleaved layout (include/tfx.h 'GEGLU layout'); -1 = zero padding."""
    c = np.arange(2 * dip)
    blk, within = c // 64, c % 64
    feat = blk * 32 + within % 32
    ref = np.where(within >= 32, di + feat, feat)
    return np.where(feat < di, ref, -1).astype(np.int32)


def head_phys_to_ref_rows(heads: int, dim_head: int) -> np.ndarray:
    """row of the fused [q | k | v | gates] projection (3 * heads * dim_head + heads rows) for each row of the kernel layout
    (64 columns per head, 3 * heads * 64 + heads rows); -1 = zero padding."""
    c = np.arange(3 * heads * 64)
    part, h, e = c // (heads * 64), (c // 64) % heads, c % 64
    ref = np.where(e < dim_head, part * heads * dim_head + h * dim_head + e, -1)
    return np.concatenate([ref, 3 * heads * dim_head + np.arange(heads)]).astype(np.int32)


class ParamStore:
    """owns the flat buffers, the name->segment map and the bf16 shadows."""

    def __init__(self, md: ModelDims, root: nn.Module):
        self.md = md
        self.specs = param_specs(md)
        self.offsets, off = {}, 0
        for name, shape in self.specs:
            self.offsets[name] = (off, shape)
            off += pad_to(int(np.prod(shape)), 4)             # keep every segment 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(self.numel)
        self.grad = None
        self.params = {}
        # buffers of the reference state_dict that are not learnable
        self.fourier_w = torch.randn(md.dim // 2)                                                   # T:621
        self.rot_freqs = 1. / (10000 ** (torch.arange(0, md.dim_head, 2).float() / md.dim_head))    # rotary_embedding_torch
        fan_in = {}
        for name, shape in self.specs:
            if name.endswith('weight') and len(shape) == 2:
                fan_in[name[:-len('weight')]] = shape[1]
        with torch.no_grad():
            # the random initialisers draw in MODULE order (a layer's AttentionResidual parameters right behind its feed-forward, as the reference
            # constructs them) - not in flat-buffer order, where those parameters sit behind the layer blocks: a seed then gives the same weights
            # whatever the layout
            shapes = dict(self.specs)
            is_ar = lambda n: n.endswith(('.3.pseudo_queries', '.3.norm_keys.gamma'))
            order = [n for n, _ in self.specs if not is_ar(n)]
            for i in range(md.depth):
                at = order.index(f'transformer.layers.{i}.2.fn.net.3.bias') + 1
                order[at:at] = [f'transformer.layers.{i}.3.pseudo_queries', f'transformer.layers.{i}.3.norm_keys.gamma']
            assert sorted(order) == sorted(shapes)
            for name in order:
                shape = shapes[name]
                o = self.offsets[name][0]
                view = self.flat[o:o + int(np.prod(shape))].view(shape)
                if name.endswith('bias') and not name.endswith('to_ada_ln_zero.bias'):
                    view._fan_in_bound = 1. / math.sqrt(fan_in[name[:-len('bias')]])
                init_param_(name, view)
            # registration (hence `named_parameters()` / positional optimizer state) follows the same MODULE order: only the flat-buffer OFFSETS
            # put the AttentionResidual parameters at the tail - a param-group list or a positional `torch.optim` state_dict built in the
            # reference's module order keeps meaning the same tensors
            for name in order:
                shape = shapes[name]
                o = self.offsets[name][0]
                view = self.flat[o:o + int(np.prod(shape))].view(shape)
                prm = nn.Parameter(view, requires_grad=True)
                self.params[name] = prm
                _attach(root, name, prm)
        _attach(root, 'transformer.to_time_cond.0.weights', self.fourier_w, buffer=True)
        # reference keeps freqs as a frozen Parameter (state_dict key `rotary_emb.freqs`)
        self.rot_param = nn.Parameter(self.rot_freqs, requires_grad=False)
        _attach(root, 'rotary_emb.freqs', self.rot_param)
        self.shadows = {}
        self._shadow_version = None
        self._epoch = 0                      # bumped by every write autograd cannot see (fused Adam, EMA kernel, mark_weights_changed): part of params_version()
        self._maps = {}
        self._jobs, self._job_table = [], None
        self._exp_ptrs = None
        self.device = torch.device('cpu')

    # ------------------------------------------------------------------ flat <-> views
    def view(self, name):
        o, shape = self.offsets[name]
        return self.flat[o:o + int(np.prod(shape))].view(shape)

    def grad_view(self, name):
        o, shape = self.offsets[name]
        return self.grad[o:o + int(np.prod(shape))].view(shape)

    def grad_ptr(self, name, elem_offset=0):
        return self.grad.data_ptr() + 4 * (self.offsets[name][0] + elem_offset)

    def ptr(self, name, elem_offset=0):
        return self.flat.data_ptr() + 4 * (self.offsets[name][0] + elem_offset)

    def reflatten(self, device):
        """after nn.Module._apply moved every parameter separately: gather them back into one flat buffer on
        `device` and re-point the parameters at views of it."""
        flat = torch.zeros(self.numel, device=device)
        with torch.no_grad():
            for name, shape in self.specs:
                o = self.offsets[name][0]
                v = flat[o:o + int(np.prod(shape))].view(shape)
                v.copy_(self.params[name].data)
                self.params[name].data = v
                self.params[name].grad = None
        self.flat = flat
        self.grad = torch.zeros(self.numel, device=device) if device.type == 'cuda' else None
        self.device = device
        self.shadows = {}
        self._shadow_version = None
        self._maps = {}

    def mark_dirty(self):
        """the master buffer changed behind autograd's back (fused optimizer / EMA kernels): rebuild the bf16 shadows on next use, and move
        `params_version()` - everything keyed on it (the decode plans `sample_many` keeps, with their weight-derived AdaLN tables) is stale"""
        self._shadow_version = None
        self._epoch += 1

    def params_version(self):
        """changes autograd can see: the version counters (optimizers, `load_state_dict`, `p.copy_` ...).  A parameter whose `.data` was RE-ASSIGNED
        (`p.data = w`) no longer points into the flat master buffer the kernels and the fused optimizer read: it is copied back into its slice and
        re-pointed here.  In-place writes THROUGH `.data` (`p.data.mul_(0.5)`) bump no counter and move no pointer - nothing can see them; callers
        that edit weights that way call `Transfusion.mark_weights_changed()`."""
        ver = self.fourier_w._version + (self._epoch << 32)         # raw-kernel writes (mark_dirty) are versions too
        # (this runs on every forward: one data_ptr() and one version read per parameter against a cached list of expected addresses - the
        #  list is rebuilt when the flat buffer moves - instead of a name lookup + offset arithmetic per parameter)
        base = self.flat.data_ptr()
        exp = self._exp_ptrs
        if exp is None or exp[0] != base:
            exp = self._exp_ptrs = (base, [(name, prm, base + 4 * self.offsets[name][0]) for name, prm in self.params.items()])
        for name, prm, want in exp[1]:
            if prm.data_ptr() != want:
                v = self.view(name)
                if prm.shape != v.shape:
                    raise ValueError(f'parameter {name} was re-assigned with shape {tuple(prm.shape)}, the model was built for {tuple(v.shape)}')
                with torch.no_grad():
                    v.copy_(prm.data)
                prm.data = v
                self._shadow_version = None
            ver += prm._version
        return ver

    def ensure_grad_views(self):
        """(re)attach `.grad` views; zero the segments whose grad was None (fresh accumulation)."""
        fresh = [n for n, p in self.params.items() if p.grad is None]
        if len(fresh) == len(self.params):
            self.grad.zero_()
        else:
            for n in fresh:
                self.grad_view(n).zero_()
        for n in fresh:
            self.params[n].grad = self.grad_view(n)

    # ------------------------------------------------------------------ shadows
    def _map(self, key, arr):
        if key not in self._maps:
            self._maps[key] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)
        return self._maps[key]

    def _shadow(self, key, rows, cols):
        if key not in self.shadows:
            self.shadows[key] = torch.zeros(rows, cols, device=self.device, dtype=torch.bfloat16)
        return self.shadows[key]

    def _cast(self, stream, src_ptr, ld_src, Rs, Cs, dst, rowmap=None, transpose=False, n_rows_logical=None):
        """record one shadow job: dst[r][c] = src[map(r)][c]   (or dst[c][r] when transpose)."""
        ld_dst, Rd = dst.shape[1], dst.shape[0]
        assert ld_dst % 8 == 0
        nb = ((ld_dst + 63) // 64) * ((Rd + 63) // 64) if transpose else (Rd * ld_dst + 2047) // 2048
        self._jobs.append((dict(src=src_ptr, ld_src=ld_src, Rs=Rs, Cs=Cs, rowmap=capi.ptr(rowmap), dst=dst.data_ptr(), ld_dst=ld_dst, Rd=Rd,
                                Cd=n_rows_logical if transpose else ld_dst, transposed=int(transpose)), nb))

    def _launch_casts(self, stream):
        """all recorded shadow jobs in one tfx_cast_batch launch; the device job table is cached per flat-buffer address."""
        key = (self.flat.data_ptr(), len(self._jobs))
        if self._job_table is None or self._job_table[0] != key:
            J = capi.STRUCTS['tfx_cast_job']
            arr = (J * len(self._jobs))()
            first = 0
            for i, (f, nb) in enumerate(self._jobs):
                for k, v in f.items():
                    setattr(arr[i], k, v)
                arr[i].first_block = first
                first += nb
            import ctypes
            raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8)
            self._job_table = (key, raw.to(self.device), len(self._jobs), first)
        _, tab, nj, nblk = self._job_table
        capi.check(capi.lib().tfx_cast_batch(tab.data_ptr(), nj, nblk, stream), 'tfx_cast_batch')

    def refresh_shadows(self, stream, force=False):
        ver = self.params_version()
        if not force and ver == self._shadow_version:
            return
        md = self.md
        d, hd, di, dip, D = md.dim, md.hd, md.di, md.dip, md.depth
        S, C = self._shadow, self._cast
        self._jobs = []
        # AdaLN conditioning: one [nt3, 4d] matrix (+ transposed) for every layer / wrapper
        C(stream, self.ptr('transformer.layers.0.1.to_film.weight'), 4 * d, md.nt3, 4 * d, S('ada', md.nt3, 4 * d))
        C(stream, self.ptr('transformer.layers.0.1.to_film.weight'), 4 * d, md.nt3, 4 * d, S('ada_t', 4 * d, md.nt3), transpose=True, n_rows_logical=md.nt3)
        C(stream, self.ptr('transformer.to_time_cond.1.weight'), d + 1, 4 * d, d + 1, S('time', 4 * d, md.kf))
        gmap = self._map('geglu', geglu_phys_to_ref_rows(di, dip))
        dh = md.dim_head
        if dh != 64:
            qmap = self._map('heads', head_phys_to_ref_rows(md.heads, dh))
            gam_map = self._map('gamma', np.where(np.arange(64) < dh, np.arange(64), -1))
        for i in range(D):
            p = f'transformer.layers.{i}'
            if md.has_skip(i):
                C(stream, self.ptr(f'{p}.0.weight'), 2 * d, d, 2 * d, S(f'skip{i}', d, 2 * d))
                C(stream, self.ptr(f'{p}.0.weight'), 2 * d, d, 2 * d, S(f'skip_t{i}', 2 * d, d), transpose=True, n_rows_logical=d)
            if dh == 64:
                C(stream, self.ptr(f'{p}.1.fn.to_qk.0.weight'), d, md.nq, d, S(f'qkvg{i}', md.nq, d))
                C(stream, self.ptr(f'{p}.1.fn.to_qk.0.weight'), d, md.nq, d, S(f'qkvg_t{i}', d, md.ldq), transpose=True, n_rows_logical=md.nq)
                C(stream, self.ptr(f'{p}.1.fn.to_out.1.weight'), hd, d, hd, S(f'out{i}', d, hd))
                C(stream, self.ptr(f'{p}.1.fn.to_out.1.weight'), hd, d, hd, S(f'out_t{i}', hd, d), transpose=True, n_rows_logical=d)
            else:             # dim_head < 64: every head is zero-padded to the kernels' 64 columns
                hdk = md.hdk
                C(stream, self.ptr(f'{p}.1.fn.to_qk.0.weight'), d, md.nq, d, S(f'qkvg{i}', md.nqk, d), rowmap=qmap)
                C(stream, self.ptr(f'{p}.1.fn.to_qk.0.weight'), d, md.nq, d, S(f'qkvg_t{i}', d, md.ldq), rowmap=qmap, transpose=True, n_rows_logical=md.nqk)
                wo = self.ptr(f'{p}.1.fn.to_out.1.weight')            # [d, heads * dh] seen as [(d * heads), dh] -> [(d * heads), 64]
                C(stream, wo, dh, d * md.heads, dh, S(f'out{i}', d, hdk).view(d * md.heads, 64))
                ot = S(f'out_t{i}', hdk, d)
                for h in range(md.heads):                            # head h: columns [h dh, (h+1) dh) of to_out -> rows [64 h, 64 h + dh)
                    C(stream, wo + 4 * h * dh, hd, d, dh, ot[h * 64:(h + 1) * 64], transpose=True, n_rows_logical=d)
                for nm in ('q', 'k'):                                 # RMSNorm gammas padded to 64 (the kernels index 64 per head)
                    key = f'g{nm}{i}'
                    if key not in self.shadows:
                        self.shadows[key] = torch.zeros(64, device=self.device)
                    capi.check(capi.lib().tfx_gather_f32(self.ptr(f'{p}.1.fn.{nm}_norm.gamma'), gam_map.data_ptr(), self.shadows[key].data_ptr(), 64, stream), 'gather_f32')
            C(stream, self.ptr(f'{p}.2.fn.net.0.weight'), d, 2 * di, d, S(f'ff1{i}', 2 * dip, d), rowmap=gmap)
            C(stream, self.ptr(f'{p}.2.fn.net.0.weight'), d, 2 * di, d, S(f'ff1_t{i}', d, 2 * dip), rowmap=gmap, transpose=True, n_rows_logical=2 * dip)
            C(stream, self.ptr(f'{p}.2.fn.net.3.weight'), di, d, di, S(f'ff2{i}', d, dip))
            C(stream, self.ptr(f'{p}.2.fn.net.3.weight'), di, d, di, S(f'ff2_t{i}', dip, d), transpose=True, n_rows_logical=d)
            b1 = self.shadows.get(f'ff1b{i}')
            if b1 is None:
                b1 = self.shadows[f'ff1b{i}'] = torch.zeros(2 * dip, device=self.device)
            capi.check(capi.lib().tfx_gather_f32(self.ptr(f'{p}.2.fn.net.0.bias'), gmap.data_ptr(), b1.data_ptr(), 2 * dip, stream), 'gather_f32')
        if (md.pos_types or md.ext_types) and 'eye' not in self.shadows:
            # identity matrix: `C[rowmap[r]] += A[r]` (additive token rows, gradients of rows handed out) is the NT GEMM's mapped RESID epilogue with B = I
            self.shadows['eye'] = torch.eye(d, device=self.device, dtype=torch.bfloat16)
        for t, dl in enumerate(md.dim_latents):
            dlp = pad_to(dl, 64)
            if t in md.ext_types:
                continue
            if dl != d:
                C(stream, self.ptr(f'latent_to_model_projs.{t}.weight'), dl, d, dl, S(f'in{t}', d, dlp))
            C(stream, self.ptr(f'model_to_latent_projs.{t}.weight'), d, dl, d, S(f'outp{t}', dl, d))
            C(stream, self.ptr(f'model_to_latent_projs.{t}.weight'), d, dl, d, S(f'outp_t{t}', d, dlp), transpose=True, n_rows_logical=dl)
        C(stream, self.ptr('text_embed.weight'), d, md.vocab, d, S('embed', md.vocab, d))
        C(stream, self.ptr('to_text_logits.weight'), d, md.vocab, d, S('logits', md.vp, d))
        C(stream, self.ptr('to_text_logits.weight'), d, md.vocab, d, S('logits_t', d, md.vp), transpose=True, n_rows_logical=md.vocab)
        self._launch_casts(stream)
        self._shadow_version = ver
