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


def scan_batch(modalities, *, num_modalities, dim_latents, sos_id, eos_id, meta_id, som_ids, eom_ids,
               add_sos_eos: bool, add_meta: bool = True) -> PackedBatch:
    b = len(modalities)
    rows, positions = [], []
    user_text, dest = [], []
    droppable = []
    inst_b, inst_m, inst_type, inst_off, inst_len, inst_shape = [], [], [], [], [], []
    latents = {t: [] for t in range(num_modalities)}
    row_inst = {t: [] for t in range(num_modalities)}
    row_pos_local = {t: [] for t in range(num_modalities)}   # (sample, offset) pairs, resolved after n_full is known
    for bi, sample in enumerate(modalities):
        ids = []          # python ints for specials, None for user text (filled on device), -1 for latent slots
        drop = []
        m = 0
        pos = []
        if add_sos_eos:
            ids.append(sos_id); drop.append(True)
        for part in sample:
            if torch.is_tensor(part) and part.is_floating_point():        # bare float tensor = modality type 0 (T:3060)
                part = (0, part)
            if not isinstance(part, tuple):
                assert is_int_tensor(part), 'text must be an int / long tensor'
                L = part.numel()
                assert part.ndim <= 1
                user_text.append(part.reshape(-1))
                dest.append((bi, len(ids), L))
                ids.extend([None] * L); drop.extend([True] * L)
                continue
            ty, x = part[0], part[1]
            assert 0 <= ty < num_modalities, f'received a modality index that is out of range. only {num_modalities} modalities specified'
            assert x.shape[-1] == dim_latents[ty], f'mismatch for modality latent dimension - expected {dim_latents[ty]} but received {x.shape[-1]}'
            axial = tuple(x.shape[:-1])
            L = math.prod(axial)
            if add_meta:
                shape_str = ','.join(map(str, axial))
                ids.append(meta_id); ids.extend(ord(c) + meta_id + 1 for c in shape_str); ids.append(som_ids[ty])
                drop.extend([False] * (len(shape_str) + 2))
            off = len(ids)
            ids.extend([-1] * L); drop.extend([False] * L)
            if add_meta:
                ids.append(eom_ids[ty]); drop.append(False)
            g = len(inst_b)
            inst_b.append(bi); inst_m.append(m); inst_type.append(ty); inst_off.append(off); inst_len.append(L); inst_shape.append(axial)
            pos.append((ty, off, L))
            latents[ty].append(x.reshape(L, -1))
            row_inst[ty].append(np.full(L, g, dtype=np.int32))
            row_pos_local[ty].append((bi, off, L))
            m += 1
        if add_sos_eos:
            ids.append(eos_id); drop.append(True)
        rows.append(ids); droppable.append(drop); positions.append(pos)
    lens = np.array([len(r) for r in rows], dtype=np.int64)
    n_full = int(lens.max()) if b else 0
    text_host = np.full((b, n_full), -1, dtype=np.int32)
    cfg_drop = np.zeros((b, n_full), dtype=bool)
    for bi, (ids, drop) in enumerate(zip(rows, droppable)):
        arr = np.array([(-1 if v is None else v) for v in ids], dtype=np.int32)
        text_host[bi, :len(ids)] = arr
        cfg_drop[bi, :len(ids)] = drop
    text_dest = np.concatenate([np.arange(L, dtype=np.int64) + (bi * n_full + o) for bi, o, L in dest]) if dest else np.zeros(0, np.int64)
    row_pos = {}
    for t in range(num_modalities):
        if row_pos_local[t]:
            row_pos[t] = np.concatenate([np.arange(L, dtype=np.int64) + (bi * n_full + o) for bi, o, L in row_pos_local[t]]).astype(np.int32)
            row_inst[t] = np.concatenate(row_inst[t])
        else:
            row_inst.pop(t); latents.pop(t)
    return PackedBatch(
        b=b, n_full=n_full, text_host=text_host, user_text=user_text, text_dest=text_dest, cfg_droppable=cfg_drop,
        positions=positions, inst_b=np.array(inst_b, np.int64), inst_m=np.array(inst_m, np.int64),
        inst_type=np.array(inst_type, np.int32), inst_off=np.array(inst_off, np.int32), inst_len=np.array(inst_len, np.int32),
        inst_shape=inst_shape, latents=latents, row_inst=row_inst, row_pos=row_pos, lens=lens, total_tokens=int(lens.sum()))


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
    tok_inst = np.full((b, n), -1, dtype=np.int32)
    ar = np.arange(n, dtype=np.int32)
    kv_end = np.tile(ar + 1, (b, 1))
    q_start = np.tile(ar, (b, 1))
    extra = np.zeros((b, n), dtype=np.int32)
    counts = np.zeros(num_modalities, dtype=np.int64)
    for g in range(len(P.inst_b)):
        bi, off, L, ty = P.inst_b[g], int(P.inst_off[g]), int(P.inst_len[g]), P.inst_type[g]
        lo, hi = min(off, n), min(off + L, n)
        if lo >= hi:
            continue
        tok_inst[bi, lo:hi] = g
        kv_end[bi, lo:hi] = np.maximum(kv_end[bi, lo:hi], hi)
        q_start[bi, lo:hi] = lo
        extra[bi, lo + 1:hi] = 1
        counts[ty] += hi - lo
    rot = ar[None, :] - np.cumsum(extra, axis=1, dtype=np.int32) + rot_offset
    return TokenMaps(n=n, tok_inst=tok_inst, kv_end=kv_end.astype(np.int32), q_start=q_start.astype(np.int32),
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


def token_segments(tok_inst: np.ndarray, max_text_run: int = 8):
    """runs of consecutive tokens (within a sample row) sharing one tok_inst value; text runs are chopped to
    <= max_text_run tokens so the waves that own them stay balanced.  Returns flat (start, length) arrays."""
    b, n = tok_inst.shape
    starts, lens = [], []
    for bi in range(b):
        row = tok_inst[bi]
        cut = np.flatnonzero(np.diff(row)) + 1
        bounds = np.concatenate(([0], cut, [n]))
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if row[lo] >= 0:
                starts.append(bi * n + lo); lens.append(hi - lo)
            else:
                for s in range(lo, hi, max_text_run):
                    starts.append(bi * n + s); lens.append(min(max_text_run, hi - s))
    return np.asarray(starts, dtype=np.int32), np.asarray(lens, dtype=np.int32)
