"""CPU restatement (PyTorch fp32) of the reference training hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows; `T:` = /root/reference/transfusion_pytorch/transfusion.py,
`MP:` = /root/reference/transfusion_pytorch/modality_processing.py.

The restatement is functional: it consumes a reference-keyed `state_dict`
 and a reference-shaped batch (`list[list[Tensor | (int, Tensor)]]`).
Gradients come from autograd over the dict's leaf tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    num_text_tokens: int
    dim: int
    depth: int
    dim_latents: tuple
    heads: int = 8
    dim_head: int = 64
    ff_expansion_factor: float = 4.
    softcap: float = 50.
    text_loss_weight: float = 1.
    flow_loss_weight: float = 1.
    ignore_index: int = -1
    model_output_clean: bool = False      # T:1297: the model predicts the clean latent, flows are derived (MP:100-126)
    eps: float = 1e-2                     # T:1319: floor of (1 - t) in that conversion

    # vocabulary layout, T:1420-1449, T:1503
    @property
    def num_modalities(self): return len(self.dim_latents)
    @property
    def sos_id(self): return self.num_text_tokens
    @property
    def eos_id(self): return self.num_text_tokens + 1
    @property
    def null_text_id(self): return self.num_text_tokens + 2
    @property
    def som_ids(self): return [self.num_text_tokens + 3 + i for i in range(self.num_modalities)]
    @property
    def eom_ids(self): return [self.num_text_tokens + 3 + self.num_modalities + i for i in range(self.num_modalities)]
    @property
    def meta_id(self): return self.num_text_tokens + 3 + 2 * self.num_modalities
    @property
    def vocab(self): return self.num_text_tokens + 3 + 2 * self.num_modalities + 129
    @property
    def dim_ff_inner(self): return int(self.dim * self.ff_expansion_factor * 2 / 3)  # T:842

    def state_dict_shapes(self) -> dict:
        """reference `state_dict` keys and shapes."""
        d, hd, h, di = self.dim, self.heads * self.dim_head, self.heads, self.dim_ff_inner
        s = {}
        s['transformer.to_time_cond.0.weights'] = (d // 2,)
        s['transformer.to_time_cond.1.weight'] = (4 * d, d + 1)
        s['transformer.to_time_cond.1.bias'] = (4 * d,)
        for i in range(self.depth):
            p = f'transformer.layers.{i}'
            if i >= self.depth / 2:                      # T:1081-1083
                s[f'{p}.0.weight'] = (d, 2 * d)
            for w in (1, 2):
                s[f'{p}.{w}.layernorm_gamma'] = (d,)
                s[f'{p}.{w}.layerscale'] = (d,)
                s[f'{p}.{w}.to_film.weight'] = (2 * d, 4 * d)
                s[f'{p}.{w}.to_film.bias'] = (2 * d,)
                s[f'{p}.{w}.to_ada_ln_zero.weight'] = (d, 4 * d)
                s[f'{p}.{w}.to_ada_ln_zero.bias'] = (d,)
            s[f'{p}.1.fn.to_qk.0.weight'] = (2 * hd, d)
            s[f'{p}.1.fn.q_norm.gamma'] = (self.dim_head,)
            s[f'{p}.1.fn.k_norm.gamma'] = (self.dim_head,)
            s[f'{p}.1.fn.to_v.0.weight'] = (hd, d)
            s[f'{p}.1.fn.to_gates.0.weight'] = (h, d)
            s[f'{p}.1.fn.to_out.1.weight'] = (d, hd)
            s[f'{p}.2.fn.net.0.weight'] = (2 * di, d)
            s[f'{p}.2.fn.net.0.bias'] = (2 * di,)
            s[f'{p}.2.fn.net.3.weight'] = (d, di)
            s[f'{p}.2.fn.net.3.bias'] = (d,)
            s[f'{p}.3.pseudo_queries'] = (d,)
            s[f'{p}.3.norm_keys.gamma'] = (d,)
        s['transformer.norm.gamma'] = (d,)
        for t, dl in enumerate(self.dim_latents):
            if dl != d:                                   # T:1478 (Identity when equal)
                s[f'latent_to_model_projs.{t}.weight'] = (d, dl)
                s[f'latent_to_model_projs.{t}.bias'] = (d,)
            s[f'model_to_latent_projs.{t}.weight'] = (dl, d)
        s['rotary_emb.freqs'] = (self.dim_head // 2,)
        s['text_embed.weight'] = (self.vocab, d)
        s['to_text_logits.weight'] = (self.vocab, d)
        return s


# ---------------------------------------------------------------------------
# packing  (MP:206-377, MP:850-936; T:3010-3023, T:3135-3144)
# ---------------------------------------------------------------------------

@dataclass
class Packed:
    text: torch.Tensor                 # (b, n+1) int64, -1 on latent slots / padding           MP:908-915
    positions: list                    # per sample [(type, offset, length)]                      MP:348
    latents: dict                      # type -> (R, dl) fp32 concatenated in scan order          MP:642
    inst_of_row: dict                  # type -> (R,) long: global instance index of each row
    inst_b: list = field(default_factory=list)      # per global instance: sample index
    inst_type: list = field(default_factory=list)
    inst_off: list = field(default_factory=list)
    inst_len: list = field(default_factory=list)
    inst_m: list = field(default_factory=list)      # index within the sample (selects times[b, m])  MP:239
    total_tokens: int = 0                            # MP:921


def default_times(num_modalities: torch.Tensor, u_k: torch.Tensor, u_t: torch.Tensor) -> torch.Tensor:
    """`default_modality_length_to_time_fn`, T:186-200, with its two uniform draws passed in (T:193 -> u_k, T:197 -> u_t):
    per sample k = floor(u_k * m_b); instances < k (already 'decoded') sit at t = 0.5, all the others share the one time u_t."""
    m = int(num_modalities.max()) if num_modalities.numel() else 0
    if m == 0:
        return torch.empty((num_modalities.shape[0], 0))
    k = torch.floor(u_k * num_modalities.float())                         # T:193
    prev = torch.arange(m)[None, :] < k[:, None]                          # T:196
    return torch.where(prev, torch.tensor(0.5), u_t[:, None].expand(-1, m))   # T:200


def pack_batch(cfg: OracleConfig, modalities, add_sos_eos=True, uncond_rows=()) -> Packed:
    """`uncond_rows`: sample indices hit by the classifier-free-guidance drop (T:3027-3043): EVERY int tensor of those samples -
    the [sos] / [eos] added just before included - is replaced by `null_text_id`; the tokens the packer adds around modalities are not."""
    b = len(modalities)
    texts, positions, total = [], [], 0
    lat = {t: [] for t in range(cfg.num_modalities)}
    inst_rows = {t: [] for t in range(cfg.num_modalities)}
    P = Packed(None, None, None, None)
    for bi, sample in enumerate(modalities):
        sample = list(sample)
        if add_sos_eos:                                                   # T:3016-3023
            sample = [torch.tensor([cfg.sos_id]), *sample, torch.tensor([cfg.eos_id])]
        if bi in uncond_rows:                                             # T:3032-3043
            sample = [torch.full_like(p, cfg.null_text_id) if torch.is_tensor(p) and not p.is_floating_point() else p for p in sample]
        ids, pos, m = [], [], 0
        for part in sample:
            if torch.is_tensor(part) and part.is_floating_point():        # T:3060-3061: bare float tensor is type 0
                part = (0, part)
            if not isinstance(part, tuple):
                part = part.reshape(-1)
                ids.extend(int(v) for v in part.tolist())
                continue
            ty, x = part
            assert 0 <= ty < cfg.num_modalities                            # MP:152
            assert x.shape[-1] == cfg.dim_latents[ty]                      # MP:157
            axial = tuple(x.shape[:-1])
            L = math.prod(axial)
            shape_str = ','.join(map(str, axial))
            ids.append(cfg.meta_id)                                        # MP:336-350
            ids.extend(ord(c) + cfg.meta_id + 1 for c in shape_str)        # T:243-249 (char_tokenize offset)
            ids.append(cfg.som_ids[ty])
            off = len(ids)
            ids.extend([-1] * L)
            ids.append(cfg.eom_ids[ty])
            pos.append((ty, off, L))
            g = len(P.inst_b)
            P.inst_b.append(bi); P.inst_type.append(ty); P.inst_off.append(off); P.inst_len.append(L); P.inst_m.append(m)
            lat[ty].append(x.reshape(L, -1).float())
            inst_rows[ty].append(torch.full((L,), g, dtype=torch.long))
            m += 1
        texts.append(ids); positions.append(pos); total += len(ids)
    n1 = max(len(t) for t in texts)
    text = torch.full((b, n1), -1, dtype=torch.long)
    for bi, ids in enumerate(texts):
        text[bi, :len(ids)] = torch.tensor(ids, dtype=torch.long)
    P.text, P.positions, P.total_tokens = text, positions, total
    P.latents = {t: torch.cat(v) for t, v in lat.items() if v}
    P.inst_of_row = {t: torch.cat(v) for t, v in inst_rows.items() if v}
    return P


# ---------------------------------------------------------------------------
# math helpers
# ---------------------------------------------------------------------------

def rms_norm(x, gamma):                                   # T:779-786
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * (gamma + 1.)


def rope(t, pos, freqs):
    """rotary_embedding_torch semantics: angle = pos*freq duplicated into
    ADJACENT slots; out = t*cos + rotate_half(t)*sin, interleaved pairs (x1,x2)->(-x2,x1).
    t (b,h,n,dh); pos (b,n)."""
    ang = pos.float()[..., None] * freqs                  # (b,n,dh/2)
    ang = ang.repeat_interleave(2, dim=-1)[:, None]       # (b,1,n,dh)
    t2 = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-t2[..., 1], t2[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def kv_end_from_positions(positions, b, n):
    """prefix-extension form of `naive_attn_mask` (T:452-470): key j visible to query i iff
    j < kv_end[i], kv_end[i] = max(i+1, end of the instance containing i)."""
    kv_end = torch.arange(1, n + 1).repeat(b, 1)
    for bi, pos in enumerate(positions):
        for (_, off, L) in pos:
            lo, hi = off, min(off + L, n)
            if lo < n:
                kv_end[bi, lo:hi] = torch.maximum(kv_end[bi, lo:hi], torch.tensor(hi))
    return kv_end


def naive_mask(positions, b, n):
    """literal `naive_attn_mask` (T:452-470) used to cross-check `kv_end_from_positions`."""
    seq = torch.arange(n)
    mask = (seq[:, None] >= seq[None, :]).repeat(b, 1, 1)
    for bi, pos in enumerate(positions):
        for (_, off, L) in pos:
            mask[bi] |= (seq[:, None] >= off) & (seq[None, :] < off + L)
    return mask


def rotary_positions(positions, b, n):
    """T:398-415: pos[i] = i - #{j <= i : j strictly inside an instance after its first token}."""
    extra = torch.zeros(b, n, dtype=torch.long)
    for bi, pos in enumerate(positions):
        for (_, off, L) in pos:
            extra[bi, off + 1: off + L] = 1
    return torch.arange(n)[None] - extra.cumsum(-1)


# ---------------------------------------------------------------------------
# transformer  (T:1100-1266)
# ---------------------------------------------------------------------------

def transformer_forward(sd, cfg: OracleConfig, x, times_tok, is_mod, kv_end, rot_pos, return_hiddens=False):
    """x (b,n,d); times_tok (b,n) (0 on text, T:3230-3232); is_mod (b,n) bool; kv_end (b,n); rot_pos (b,n)."""
    b, n, d = x.shape
    h, dh = cfg.heads, cfg.dim_head
    g = lambda k: sd[k]
    # time conditioning  T:617-635, T:1068-1072, T:1132
    w = g('transformer.to_time_cond.0.weights')
    fr = times_tok[..., None] * w * 2 * math.pi
    four = torch.cat((times_tok[..., None], fr.sin(), fr.cos()), dim=-1)
    cond = F.silu(F.linear(four, g('transformer.to_time_cond.1.weight'), g('transformer.to_time_cond.1.bias')))
    mask = torch.arange(n)[None, None, :] < kv_end[:, :, None]           # (b,i,j)
    freqs = g('rotary_emb.freqs')
    im = is_mod[..., None]

    def ada_pre(p, x):                                                    # T:747-755
        xh = F.layer_norm(x, (d,))
        gam, bet = F.linear(cond, g(f'{p}.to_film.weight'), g(f'{p}.to_film.bias')).chunk(2, dim=-1)
        return torch.where(im, xh * (gam + 1.) + bet, xh * (g(f'{p}.layernorm_gamma') + 1.))

    def ada_post(p, y):                                                   # T:763-769
        z = F.linear(cond, g(f'{p}.to_ada_ln_zero.weight'), g(f'{p}.to_ada_ln_zero.bias')).sigmoid()
        return torch.where(im, y * z, y * (g(f'{p}.layerscale') + 1.))

    skips, hiddens = [], [x]
    for li in range(cfg.depth):
        p = f'transformer.layers.{li}'
        layer = li + 1
        if layer <= cfg.depth // 2:                                       # T:1206-1219
            skips.append(x)
        elif f'{p}.0.weight' in sd:
            skip = skips.pop()
            x = F.linear(torch.cat((x, skip), dim=-1), g(f'{p}.0.weight')) + x
        # attention  T:918-1039
        u = ada_pre(f'{p}.1', x)
        qk = F.linear(u, g(f'{p}.1.fn.to_qk.0.weight')).reshape(b, n, 2, h, dh)
        q, k = qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2)    # (b,h,n,dh)   '(qk h d)'
        v = F.linear(u, g(f'{p}.1.fn.to_v.0.weight')).reshape(b, n, h, dh).transpose(1, 2)
        q = rms_norm(q, g(f'{p}.1.fn.q_norm.gamma'))
        k = rms_norm(k, g(f'{p}.1.fn.k_norm.gamma'))
        q, k = rope(q, rot_pos, freqs), rope(k, rot_pos, freqs)
        sim = torch.einsum('bhid,bhjd->bhij', q * dh ** -0.5, k)
        sim = torch.tanh(sim / cfg.softcap) * cfg.softcap                  # T:1001
        sim = torch.where(mask[:, None], sim, torch.tensor(-torch.finfo(sim.dtype).max))
        attn = sim.softmax(dim=-1)
        o = torch.einsum('bhij,bhjd->bhid', attn, v)
        gates = F.linear(u, g(f'{p}.1.fn.to_gates.0.weight')).sigmoid()    # (b,n,h)   T:1027
        o = o * gates.transpose(1, 2)[..., None]
        o = o.transpose(1, 2).reshape(b, n, h * dh)
        y = F.linear(o, g(f'{p}.1.fn.to_out.1.weight'))
        x = ada_post(f'{p}.1', y) + x
        # feedforward  T:831-853
        u = ada_pre(f'{p}.2', x)
        a, gate = F.linear(u, g(f'{p}.2.fn.net.0.weight'), g(f'{p}.2.fn.net.0.bias')).chunk(2, dim=-1)
        y = F.linear(F.gelu(gate) * a, g(f'{p}.2.fn.net.3.weight'), g(f'{p}.2.fn.net.3.bias'))
        x = ada_post(f'{p}.2', y) + x
        hiddens.append(x)
        # attention residual  T:790-829
        H = torch.stack(hiddens)                                          # (l,b,n,d)
        keys = rms_norm(H, g(f'{p}.3.norm_keys.gamma'))
        sim = torch.einsum('lbnd,d->bnl', keys, g(f'{p}.3.pseudo_queries')) * d ** -0.5
        x = torch.einsum('bnl,lbnd->bnd', sim.softmax(dim=-1), H)
    out = rms_norm(x, g('transformer.norm.gamma'))                         # T:1250
    if return_hiddens:
        return out, hiddens
    return out


# ---------------------------------------------------------------------------
# Transfusion.forward, list branch, training  (T:2926-3450)
# ---------------------------------------------------------------------------

def forward_train(sd, cfg: OracleConfig, modalities, times, noise, return_all=False, uncond_rows=()):
    """times (b, m_max) fp32 (injected; T:2933); noise: type -> (R, dl) in scan order (see detdata.det_noise).
    `uncond_rows`: samples whose text is dropped for classifier-free guidance (see pack_batch).
    Returns loss (and a dict of intermediates when `return_all`)."""
    P = pack_batch(cfg, modalities, add_sos_eos=True, uncond_rows=uncond_rows)
    b, n1 = P.text.shape
    n = n1 - 1
    d = cfg.dim
    # noising + latent_to_model  MP:617-689
    tokens_mod = torch.zeros(b, n1, d)
    flows, projected = {}, {}
    inst_time = torch.stack([times[bi, m] for bi, m in zip(P.inst_b, P.inst_m)]) if P.inst_b else torch.zeros(0)
    for t, x in P.latents.items():
        tt = inst_time[P.inst_of_row[t]][:, None]
        eps = noise[t]
        xt = x * tt + eps * (1. - tt)
        flows[t] = x - eps
        key = f'latent_to_model_projs.{t}.weight'
        proj = F.linear(xt, sd[key], sd[f'latent_to_model_projs.{t}.bias']) if key in sd else xt
        projected[t] = proj
        r = 0
        for gi in P.inst_of_row[t].unique_consecutive().tolist():
            L = P.inst_len[gi]
            tokens_mod[P.inst_b[gi], P.inst_off[gi]:P.inst_off[gi] + L] = proj[r:r + L]
            r += L
    # drop last token, labels  T:3135-3144
    text, labels = P.text[:, :-1], P.text[:, 1:]
    tokens_mod = tokens_mod[:, :-1]
    is_mod = torch.zeros(b, n, dtype=torch.bool)
    is_type = torch.zeros(cfg.num_modalities, b, n, dtype=torch.bool)
    times_tok = torch.zeros(b, n)
    for gi in range(len(P.inst_b)):
        bi, off, L = P.inst_b[gi], P.inst_off[gi], P.inst_len[gi]
        is_mod[bi, off:off + L] = True                                    # T:421-450 (within the n-long view)
        is_type[P.inst_type[gi], bi, off:off + L] = True
        times_tok[bi, off:off + L] = inst_time[gi]                        # T:3230-3232
    emb = sd['text_embed.weight'][text.clamp(min=0)]                      # T:3173-3175
    tokens = torch.where(is_mod[..., None], tokens_mod, emb)              # T:3184
    kv_end = kv_end_from_positions(P.positions, b, n)
    rot = rotary_positions(P.positions, b, n)
    embed = transformer_forward(sd, cfg, tokens, times_tok, is_mod, kv_end, rot)
    logits = F.linear(embed, sd['to_text_logits.weight'])                 # T:3280
    # flow predictions  T:3290-3311
    pred = {}
    for t in P.latents:
        rows = []
        for gi in P.inst_of_row[t].unique_consecutive().tolist():
            bi, off, L = P.inst_b[gi], P.inst_off[gi], P.inst_len[gi]
            rows.append(embed[bi, off:off + L])
        out_rows = torch.cat(rows)
        if cfg.model_output_clean:
            # MP:786-792: in the interleaved path the conversion happens in MODEL space, against the PROJECTED noised tokens,
            # before model_to_latent (forward_modality converts in latent space instead, T:2772-2810)
            out_rows = (out_rows - projected[t]) / (1. - inst_time[P.inst_of_row[t]][:, None]).clamp_min(cfg.eps)
        pred[t] = F.linear(out_rows, sd[f'model_to_latent_projs.{t}.weight'])
    # losses  T:3320-3376
    lab = labels.masked_fill(is_mod, cfg.ignore_index)
    lab = lab.masked_fill(lab == cfg.null_text_id, cfg.ignore_index)
    text_loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), lab.reshape(-1), ignore_index=cfg.ignore_index)
    w_text = (lab != cfg.ignore_index).sum() / P.total_tokens
    flow_losses, flow_w = [], []
    for t in range(cfg.num_modalities):
        w_t = is_type[t].sum() / P.total_tokens
        if t in pred:
            flow_losses.append(F.mse_loss(pred[t], flows[t]))
            flow_w.append(w_t)
    flow_loss = sum(fl * w for fl, w in zip(flow_losses, flow_w)) if flow_losses else torch.zeros(())
    total = text_loss * w_text * cfg.text_loss_weight + flow_loss * cfg.flow_loss_weight
    if not return_all:
        return total
    return dict(loss=total, text_loss=text_loss, flow_losses=flow_losses, logits=logits, embed=embed,
                pred_flows=pred, flows=flows, labels=lab, packed=P, kv_end=kv_end, rot_pos=rot,
                is_mod=is_mod, times_tok=times_tok, tokens=tokens)


# ---------------------------------------------------------------------------
# Transfusion.forward_text  (T:2586-2664): pure-text LM path = the transformer with a causal mask, no conditioning,
# cross entropy over the text-only part of the vocabulary (text_only_logits_mask, T:1509-1510, T:2653)
# ---------------------------------------------------------------------------

def forward_text(sd, cfg: OracleConfig, text, return_all=False):
    """text: (b, n+1) int64 token ids (-1 = padding / ignore).  Returns the loss (T:2655-2659)."""
    inp, labels = text[:, :-1], text[:, 1:]                               # T:2603-2604
    b, n = inp.shape
    tokens = sd['text_embed.weight'][inp.masked_fill(inp == -1, 0)]       # T:2608-2609
    kv_end = torch.arange(1, n + 1).repeat(b, 1)                          # causal_mask=True  (T:2627)
    rot = torch.arange(n)[None].repeat(b, 1)                              # T:2617-2621
    is_mod = torch.zeros(b, n, dtype=torch.bool)
    embed = transformer_forward(sd, cfg, tokens, torch.zeros(b, n), is_mod, kv_end, rot)
    logits = F.linear(embed, sd['to_text_logits.weight'])                 # T:2639
    masked = logits[..., :cfg.num_text_tokens]                            # == masked_fill(~text_only_logits_mask, -max) under softmax
    loss = F.cross_entropy(masked.reshape(-1, masked.shape[-1]), labels.reshape(-1), ignore_index=cfg.ignore_index)
    if not return_all:
        return loss
    return dict(loss=loss, logits=logits, embed=embed)


# ---------------------------------------------------------------------------
# Transfusion.forward_modality  (T:2710-2869): pure flow path = the transformer with `modality_only=True`
# (every token conditioned on its sample's time, no mask, no rotary embedding), MSE on the predicted flow
# ---------------------------------------------------------------------------

def forward_modality(sd, cfg: OracleConfig, x, times, noise=None, modality_type=0, return_all=False):
    """x: (b, *axial, dl) latents; times (b,); noise like x (None = no noising, the `return_loss=False` call)."""
    b, dl = x.shape[0], x.shape[-1]
    t = modality_type
    if noise is not None:
        tt = times.reshape(b, *([1] * (x.ndim - 1)))
        xt = tt * x + (1. - tt) * noise                                    # T:2755
        flow = x - noise                                                   # T:2757
    else:
        xt = x
    key = f'latent_to_model_projs.{t}.weight'
    tok = F.linear(xt, sd[key], sd[f'latent_to_model_projs.{t}.bias']) if key in sd else xt     # T:2770
    tok = tok.reshape(b, -1, cfg.dim)                                      # pack 'b * d'  T:2783
    n = tok.shape[1]
    is_mod = torch.ones(b, n, dtype=torch.bool)
    kv_end = torch.full((b, n), n)                                         # no mask  (modality_only, T:2800-2804)
    rot = torch.zeros(b, n, dtype=torch.long)                              # no rotary embedding is passed: position 0 = identity rotation
    embed = transformer_forward(sd, cfg, tok, times[:, None].expand(b, n), is_mod, kv_end, rot)
    pred = F.linear(embed, sd[f'model_to_latent_projs.{t}.weight']).reshape(x.shape)            # T:2808
    if cfg.model_output_clean:                                             # T:2772-2773, T:2810
        pred = (pred - xt) / (1. - times.reshape(b, *([1] * (x.ndim - 1)))).clamp_min(cfg.eps)
    if noise is None:
        return pred
    loss = F.mse_loss(pred, flow)                                          # T:2817
    if not return_all:
        return loss
    return dict(loss=loss, pred_flow=pred, flow=flow, embed=embed)


# ---------------------------------------------------------------------------
# velocity-consistency term of Transfusion.forward  (T:3084-3088, T:3378-3418; consistency flow matching, arXiv 2407.02398)
# ---------------------------------------------------------------------------

def forward_velocity(sd, sd_teacher, cfg: OracleConfig, modalities, times, noise, noise_teacher, delta=1e-3, vc_weight=0.1, return_all=False):
    """student at times * (1 - delta); EMA teacher (no grad, its own noise) at times + delta; per modality type
    MSE(student pred flow, teacher pred flow), weighted like the flow losses (token share of the type) and by `vc_weight`."""
    st = forward_train(sd, cfg, modalities, times * (1. - delta), noise, return_all=True)
    with torch.no_grad():
        te = forward_train(sd_teacher, cfg, modalities, times + delta, noise_teacher, return_all=True)
    P = st['packed']
    total = st['loss']
    vel = []
    for t in sorted(st['pred_flows']):
        w_t = sum(L for ty, L in zip(P.inst_type, P.inst_len) if ty == t) / P.total_tokens
        v = F.mse_loss(st['pred_flows'][t], te['pred_flows'][t])
        vel.append(v)
        total = total + vc_weight * v * w_t
    if not return_all:
        return total
    return dict(loss=total, velocity=vel, student=st, teacher=te)


def train_step(sd, cfg, modalities, times, noise, opt_state, lr=3e-4, clip=0.5, betas=(0.9, 0.999), eps=1e-8):
    """One `train_toy.py:50-57` step on the restatement: fwd + bwd + clip_grad_norm_(0.5) + Adam(3e-4).
    `sd` values that require grad are updated in place.  Used by bench.py's cpu_baseline ("port")."""
    params = [v for v in sd.values() if v.requires_grad]
    for p in params:
        p.grad = None
    loss = forward_train(sd, cfg, modalities, times, noise)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, clip)
    if 'opt' not in opt_state:
        opt_state['opt'] = torch.optim.Adam(params, lr=lr, betas=betas, eps=eps)
    opt_state['opt'].step()
    return float(loss)
