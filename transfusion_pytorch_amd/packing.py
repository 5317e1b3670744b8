"""Host side of the native packer: ONE structure scan of the interleaved text + latent batch, producing the
index arrays the HIP kernels consume.  No device synchronisation: token VALUES never come to the host; only
shapes are read here, and values are placed on the device with one index_copy.

Follows the reference's packing contract (MP:206-377, MP:850-936): per sample `[sos] parts... [eos]`; every
modality instance becomes `[meta] chars(shape) [som_t] <L latent slots> [eom_t]`; latent slots carry text id -1;
rows are right-padded with -1; `total_tokens` = sum of packed lengths; positions are `(type, offset, length)`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


def is_int_tensor(t):
    return torch.is_tensor(t) and t.dtype in (torch.int, torch.long)


@dataclass
class PackedBatch:
    b: int
    n_full: int                       # packed length (max over samples) BEFORE the training last-token drop
    text_host: np.ndarray             # (b, n_full) int32: meta/special tokens filled, -1 elsewhere
    user_text: list                   # device/cpu int tensors in scan order (values placed via text_dest)
    text_dest: np.ndarray             # flat destination index (into b*n_full) of every user text element
    cfg_droppable: np.ndarray         # (b, n_full) bool: positions holding user text / sos / eos (CFG null-able, T:3029-3043)
    positions: list                   # per sample [(type, offset, length)]   (public `modality_positions`)
    inst_b: np.ndarray                # per instance: sample index
    inst_m: np.ndarray                # per instance: index inside the sample (selects times[b, m], MP:239)
    inst_type: np.ndarray
    inst_off: np.ndarray
    inst_len: np.ndarray
    inst_shape: list                  # axial shape per instance
    latents: dict                     # type -> list of (L, dl) float tensors in scan order
    row_inst: dict                    # type -> (R,) int32 global instance id per latent row
    row_pos: dict                     # type -> (R,) int32 flat position b*n_full + offset + i  (in the FULL layout)
    lens: np.ndarray                  # packed length per sample
    total_tokens: int = 0


def _ranges(starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """concatenation of arange(starts[i], starts[i] + lens[i]) without a Python loop"""
    lens = np.asarray(lens, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    first = np.cumsum(lens) - lens                      # index of every range's first element in the output
    return np.repeat(np.asarray(starts, dtype=np.int64) - first, lens) + np.arange(total, dtype=np.int64)


def scan_batch(modalities, *, num_modalities, dim_latents, sos_id, eos_id, meta_id, som_ids, eom_ids,
               add_sos_eos: bool, add_meta: bool = True) -> PackedBatch:
    """one pass over the parts (never over the tokens): every part contributes a run (ids array | fill value, length, droppable flag) to its
    sample's row, and the per-token arrays are assembled from the runs with array operations"""
    b = len(modalities)
    positions = []
    user_text, dest_b, dest_o, dest_l = [], [], [], []
    inst_b, inst_m, inst_type, inst_off, inst_len, inst_shape = [], [], [], [], [], []
    latents = {t: [] for t in range(num_modalities)}
    run_b, run_o, run_l, run_fill, run_drop = [], [], [], [], []      # constant runs: (sample, offset, length, id value, droppable)
    lit_b, lit_o, lit_ids = [], [], []                                # literal runs: the [meta] shape string [som] tokens (never droppable)
    meta_cache = {}
    lens = np.zeros(b, dtype=np.int64)
    is_tensor = torch.is_tensor
    for bi, sample in enumerate(modalities):
        cur = 0
        m = 0
        pos = []
        if add_sos_eos:
            run_b.append(bi); run_o.append(0); run_l.append(1); run_fill.append(sos_id); run_drop.append(True)
            cur = 1
        for part in sample:
            if is_tensor(part) and part.is_floating_point():              # bare float tensor = modality type 0 (T:3060)
                part = (0, part)
            if not isinstance(part, tuple):
                assert is_int_tensor(part), 'text must be an int / long tensor'
                assert part.ndim <= 1
                L = part.numel()
                user_text.append(part if part.ndim == 1 else part.reshape(-1))
                dest_b.append(bi); dest_o.append(cur); dest_l.append(L)
                run_b.append(bi); run_o.append(cur); run_l.append(L); run_fill.append(-1); run_drop.append(True)
                cur += L
                continue
            ty, x = part[0], part[1]
            assert 0 <= ty < num_modalities, f'received a modality index that is out of range. only {num_modalities} modalities specified'
            assert x.shape[-1] == dim_latents[ty], f'mismatch for modality latent dimension - expected {dim_latents[ty]} but received {x.shape[-1]}'
            axial = tuple(x.shape[:-1])
            L = math.prod(axial)
            if add_meta:
                key = (axial, ty)
                lit = meta_cache.get(key)
                if lit is None:
                    shape_str = ','.join(map(str, axial))
                    lit = meta_cache[key] = np.array([meta_id, *(ord(c) + meta_id + 1 for c in shape_str), som_ids[ty]], dtype=np.int32)
                lit_b.append(bi); lit_o.append(cur); lit_ids.append(lit)
                cur += len(lit)
            off = cur
            cur += L                                                      # latent slots: id -1, not droppable = the arrays' initial values
            if add_meta:
                run_b.append(bi); run_o.append(cur); run_l.append(1); run_fill.append(eom_ids[ty]); run_drop.append(False)
                cur += 1
            inst_b.append(bi); inst_m.append(m); inst_type.append(ty); inst_off.append(off); inst_len.append(L); inst_shape.append(axial)
            pos.append((ty, off, L))
            latents[ty].append(x if x.ndim == 2 else x.reshape(L, -1))
            m += 1
        if add_sos_eos:
            run_b.append(bi); run_o.append(cur); run_l.append(1); run_fill.append(eos_id); run_drop.append(True)
            cur += 1
        lens[bi] = cur
        positions.append(pos)
    n_full = int(lens.max()) if b else 0
    text_host = np.full(b * n_full, -1, dtype=np.int32)
    cfg_drop = np.zeros(b * n_full, dtype=bool)
    if run_b:
        rb, ro, rl = np.asarray(run_b, np.int64), np.asarray(run_o, np.int64), np.asarray(run_l, np.int64)
        idx = _ranges(rb * n_full + ro, rl)
        text_host[idx] = np.repeat(np.asarray(run_fill, np.int32), rl)
        cfg_drop[idx] = np.repeat(np.asarray(run_drop, bool), rl)
    if lit_b:
        ll = np.fromiter((len(a) for a in lit_ids), dtype=np.int64, count=len(lit_ids))
        text_host[_ranges(np.asarray(lit_b, np.int64) * n_full + np.asarray(lit_o, np.int64), ll)] = np.concatenate(lit_ids)
    text_dest = _ranges(np.asarray(dest_b, np.int64) * n_full + np.asarray(dest_o, np.int64), dest_l) if dest_b else np.zeros(0, np.int64)
    ib, it, io, il = np.array(inst_b, np.int64), np.array(inst_type, np.int32), np.array(inst_off, np.int32), np.array(inst_len, np.int32)
    row_inst, row_pos = {}, {}
    for t in range(num_modalities):
        sel = np.flatnonzero(it == t)
        if sel.size:
            row_pos[t] = _ranges(ib[sel] * n_full + io[sel], il[sel]).astype(np.int32)
            row_inst[t] = np.repeat(sel.astype(np.int32), il[sel])
        else:
            latents.pop(t)
    return PackedBatch(
        b=b, n_full=n_full, text_host=text_host.reshape(b, n_full), user_text=user_text, text_dest=text_dest, cfg_droppable=cfg_drop.reshape(b, n_full),
        positions=positions, inst_b=ib, inst_m=np.array(inst_m, np.int64), inst_type=it, inst_off=io, inst_len=il,
        inst_shape=inst_shape, latents=latents, row_inst=row_inst, row_pos=row_pos, lens=lens, total_tokens=int(lens.sum()))


def scan_signature(sig, user_text, latents, *, num_modalities, dim_latents, sos_id, eos_id, meta_id, som_ids, eom_ids,
                   add_sos_eos: bool, add_meta: bool = True) -> PackedBatch:
    """`scan_batch` from the structure SIGNATURE of `fast_signature` (per sample a tuple of entries: int = a text run of that length,
    (type, *axial, dim_latent) = a modality instance) and the tensors it collected: no tensor is touched again and the per-token arrays come out of
    array operations over the parts - the Python work is one pass over the entries plus one dictionary lookup per instance.  Same result as
    `scan_batch` on the same batch (tests/test_host_cpu.py)."""
    b = len(sig)
    nparts = np.fromiter((len(ss) for ss in sig), dtype=np.int64, count=b)
    entries = [e for ss in sig for e in ss]
    N = len(entries)
    part_b = np.repeat(np.arange(b, dtype=np.int64), nparts)
    is_mod = np.fromiter((type(e) is tuple for e in entries), dtype=bool, count=N)
    tlen = np.fromiter((0 if type(e) is tuple else e for e in entries), dtype=np.int64, count=N)
    mods = [e for e in entries if type(e) is tuple]
    M = len(mods)
    inst_type = np.fromiter((e[0] for e in mods), dtype=np.int32, count=M)
    assert M == 0 or (0 <= int(inst_type.min()) and int(inst_type.max()) < num_modalities), \
        f'received a modality index that is out of range. only {num_modalities} modalities specified'
    inst_shape = [e[1:-1] for e in mods]
    for e in mods:
        assert e[-1] == dim_latents[e[0]], f'mismatch for modality latent dimension - expected {dim_latents[e[0]]} but received {e[-1]}'
    inst_len = np.fromiter((math.prod(a) for a in inst_shape), dtype=np.int64, count=M)
    lit_cache, lits = {}, []
    if add_meta:
        for ty, ax in zip(inst_type.tolist(), inst_shape):
            lit = lit_cache.get((ax, ty))
            if lit is None:
                lit = lit_cache[(ax, ty)] = np.array([meta_id, *(ord(c) + meta_id + 1 for c in ','.join(map(str, ax))), som_ids[ty]], dtype=np.int32)
            lits.append(lit)
    lit_len = np.fromiter((len(a) for a in lits), dtype=np.int64, count=M) if add_meta else np.zeros(M, np.int64)
    # tokens per part, and every part's offset inside its sample's row
    cnt = tlen.copy()
    cnt[is_mod] = lit_len + inst_len + (1 if add_meta else 0)
    lead = 1 if add_sos_eos else 0
    csum = np.cumsum(cnt) - cnt                                          # exclusive, over all parts
    first_part = np.cumsum(nparts) - nparts                              # index of every sample's first part
    per_sample = np.add.reduceat(cnt, first_part[nparts > 0]) if N else np.zeros(0, np.int64)
    tok_per_sample = np.zeros(b, np.int64); tok_per_sample[nparts > 0] = per_sample
    sample_base = np.zeros(b, np.int64); sample_base[nparts > 0] = csum[first_part[nparts > 0]]
    off = csum - sample_base[part_b] + lead                              # offset of the part in its row
    lens = tok_per_sample + 2 * lead
    n_full = int(lens.max()) if b else 0
    text_host = np.full(b * n_full, -1, dtype=np.int32)
    cfg_drop = np.zeros(b * n_full, dtype=bool)
    row0 = np.arange(b, dtype=np.int64) * n_full
    if add_sos_eos:
        text_host[row0] = sos_id; cfg_drop[row0] = True
        text_host[row0 + lens - 1] = eos_id; cfg_drop[row0 + lens - 1] = True
    # text runs: ids come from the device (text_dest), the slots are CFG-droppable
    t_sel = ~is_mod
    t_start = part_b[t_sel] * n_full + off[t_sel]
    text_dest = _ranges(t_start, tlen[t_sel])
    cfg_drop[text_dest] = True
    # instances: [meta] shape [som] literal, L latent slots (id -1, not droppable = the initial values), [eom]
    m_b, m_off = part_b[is_mod], off[is_mod]
    inst_off = m_off + lit_len
    if add_meta and M:
        text_host[_ranges(m_b * n_full + m_off, lit_len)] = np.concatenate(lits)
        text_host[m_b * n_full + inst_off + inst_len] = np.asarray(eom_ids, dtype=np.int32)[inst_type]
    # index of the instance inside its sample
    first_inst = np.cumsum(np.bincount(m_b, minlength=b)) - np.bincount(m_b, minlength=b) if M else np.zeros(b, np.int64)
    inst_m = np.arange(M, dtype=np.int64) - first_inst[m_b] if M else np.zeros(0, np.int64)
    positions = [[] for _ in range(b)]
    for bi, ty, o, L in zip(m_b.tolist(), inst_type.tolist(), inst_off.tolist(), inst_len.tolist()):
        positions[bi].append((ty, o, L))
    row_inst, row_pos, lat = {}, {}, {}
    for t in range(num_modalities):
        sel = np.flatnonzero(inst_type == t)
        if sel.size:
            row_pos[t] = _ranges(m_b[sel] * n_full + inst_off[sel], inst_len[sel]).astype(np.int32)
            row_inst[t] = np.repeat(sel.astype(np.int32), inst_len[sel])
            lat[t] = latents[t]
    return PackedBatch(
        b=b, n_full=n_full, text_host=text_host.reshape(b, n_full), user_text=user_text, text_dest=text_dest, cfg_droppable=cfg_drop.reshape(b, n_full),
        positions=positions, inst_b=m_b, inst_m=inst_m, inst_type=inst_type, inst_off=inst_off.astype(np.int32), inst_len=inst_len.astype(np.int32),
        inst_shape=[tuple(a) for a in inst_shape], latents=lat, row_inst=row_inst, row_pos=row_pos, lens=lens, total_tokens=int(lens.sum()))


@dataclass
class TokenMaps:
    """per-token index arrays over the (b, n) view the transformer sees."""
    n: int
    tok_inst: np.ndarray      # (b, n) int32: global instance id or -1
    kv_end: np.ndarray        # (b, n) int32: prefix-extension mask bound (naive_attn_mask T:452-470)
    q_start: np.ndarray       # (b, n) int32: first query that sees key j
    rot_pos: np.ndarray       # (b, n) int32 (T:398-415)
    is_type: np.ndarray       # (num_modalities,) count of tokens of each type inside the view (loss weights T:3343)


def token_maps(P: PackedBatch, n: int, num_modalities: int, rot_offset: int = 0) -> TokenMaps:
    b = P.b
    tok_inst = np.full(b * n, -1, dtype=np.int32)
    ar = np.arange(n, dtype=np.int32)
    kv_end = np.tile(ar + 1, b)
    q_start = np.tile(ar, b)
    extra = np.zeros(b * n, dtype=np.int32)
    counts = np.zeros(num_modalities, dtype=np.int64)
    if len(P.inst_b):
        lo = np.minimum(P.inst_off.astype(np.int64), n)
        hi = np.minimum(P.inst_off.astype(np.int64) + P.inst_len, n)
        ln = hi - lo                                                   # tokens of the instance inside the view (0 = cut off)
        idx = _ranges(P.inst_b * n + lo, ln)
        tok_inst[idx] = np.repeat(np.arange(len(ln), dtype=np.int32), ln)
        kv_end[idx] = np.repeat(hi, ln)                                # the whole block sees itself (>= the causal i + 1 inside it)
        q_start[idx] = np.repeat(lo, ln)
        extra[idx] = 1
        extra[(P.inst_b * n + lo)[ln > 0]] = 0                         # a block takes ONE rotary position: every token after its first stays put
        counts += np.bincount(P.inst_type, weights=ln, minlength=num_modalities).astype(np.int64)[:num_modalities]
    rot = ar[None, :] - np.cumsum(extra.reshape(b, n), axis=1, dtype=np.int32) + rot_offset
    return TokenMaps(n=n, tok_inst=tok_inst.reshape(b, n), kv_end=kv_end.reshape(b, n).astype(np.int32), q_start=q_start.reshape(b, n).astype(np.int32),
                     rot_pos=rot.astype(np.int32), is_type=counts)


def fast_signature(modalities):
    """one cheap pass over the ragged input: (hashable structure signature, user text tensors in scan order,
    latent tensors per type in scan order).  The signature keys the structure cache: batches with the same
    part kinds / lengths / shapes reuse every index array (already resident on the device)."""
    sig, texts, lats = [], [], {}
    for sample in modalities:
        ss = []
        for part in sample:
            if type(part) is tuple:
                ty, x = part[0], part[1]
                ss.append((ty, *x.shape))
                lats.setdefault(ty, []).append(x if x.ndim == 2 else x.reshape(-1, x.shape[-1]))
            elif part.dtype.is_floating_point:
                ss.append((0, *part.shape))
                lats.setdefault(0, []).append(part if part.ndim == 2 else part.reshape(-1, part.shape[-1]))
            else:
                ss.append(part.numel())
                texts.append(part if part.ndim == 1 else part.reshape(-1))
        sig.append(tuple(ss))
    return tuple(sig), texts, lats


def token_segments(tok_inst: np.ndarray, max_text_run: int = 8, balance: bool = False):
    """runs of consecutive tokens (within a sample row) sharing one tok_inst value; text runs are chopped to
    <= max_text_run tokens so the waves that own them stay balanced.  Returns flat (start, length) arrays.
    balance: order the segments longest first.  The segment kernels run W resident waves, wave w taking segments w, w + W, w + 2W, ...: dealt from
    a length-sorted list every wave gets the same mix of long and short segments (their work then differs by at most one segment) instead
    of whatever run of neighbours the token order happens to give it (4-token modality instances next to 8-token text runs: up to 2x)."""
    b, n = tok_inst.shape
    if b * n == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    col = np.broadcast_to(np.arange(n, dtype=np.int64), (b, n))
    start = np.ones((b, n), dtype=bool)
    start[:, 1:] = tok_inst[:, 1:] != tok_inst[:, :-1]                  # a new run begins where the value changes (and at column 0)
    run_start = np.maximum.accumulate(np.where(start, col, 0), axis=1)  # column where the token's run began
    start |= (tok_inst < 0) & ((col - run_start) % max_text_run == 0)    # text runs: a new segment every max_text_run tokens
    starts = np.flatnonzero(start.reshape(-1))
    ends = np.append(starts[1:], b * n)
    lens = ends - starts
    if balance:
        order = np.argsort(-lens, kind='stable')
        starts, lens = starts[order], lens[order]
    return starts.astype(np.int32), lens.astype(np.int32)
