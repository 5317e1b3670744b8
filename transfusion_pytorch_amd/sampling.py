"""KV-cached batched decoding: the reference's `sample_many` state machine (T:2082-2583) over the native engine.

The per-sample state machine (phases text / modality / done, prompt normalisation, [som] transitions, shape parsing,
length budget) is host bookkeeping and follows the reference line by line; every tensor operation of a decoding step
runs in the HIP kernels through a *decode plan* (engine.Plan with a KV cache):

  * text step (T:2279-2349)   one new token per active sample: embed -> transformer against the cache -> fp32 logits
  * modality step (T:2354-2556) joint fixed-grid midpoint ODE (torchdiffeq semantics) over all samples
    in the modality phase; every evaluation = latent_to_model -> transformer (FiLM path, t = step time) -> model_to_latent,
    against the conditional cache and - classifier-free guidance - the null-text cache: ONE forward over 2 B rows (the two
    caches are halves of one buffer).  The null-text cache is appended between phases (`_uncond_append`), not rebuilt.

Schedule: continuous batching (`_loop_continuous`: every global step advances every live sample by a token or by one ODE evaluation in one
mixed forward) while a mixed step stays launch-bound (<= 4096 rows), else the reference's text-rounds / joint-ODE loop (`_loop_phased`);
`TFX_SAMPLE_SCHEDULE=continuous | phased` forces one.

Cache semantics reproduced exactly: the new [som] token is NOT in the conditional cache when its modality is decoded
(the modality block takes the rotary position the [som] would have had, T:2411); the K/V committed for a decoded
modality are those of the LAST conditional ODE evaluation (T:2531-2533); past modalities are conditioned at t = 1.
"""
from __future__ import annotations

import ctypes
import math
import os
import time
from dataclasses import dataclass

import numpy as np
import torch

from . import capi
from .engine import Plan


@dataclass
class _State:                                   # _SamplingState, T:1270-1287 (token values mirrored on the host)
    parts: list                                 # host view of `sample`: lists of ints (text) and (type, tensor) tuples
    curr_seq: list                              # the text part currently being built (alias of parts[-1] when it is text)
    last_token: int | None
    phase: str = 'text'
    cache_len: int = 0
    uncond_len: int = 0
    uncond_parts: int = 0                       # parts of `parts` whose tokens are in the null-text cache (incremental update between modality phases)
    uncond_pos: int = 0                         # rotary position the next token of the null-text history takes
    tokens_seen: int = 0
    num_past_modalities: int = 0
    curr_modality_id: int | None = None
    modality_shape: tuple | None = None
    modality_length: int | None = None
    num_tokens: int = 0
    forced: tuple = (None, None)
    unfed: bool = False                         # the last token of curr_seq was sampled but has not gone through the model yet
    ode_k: int = 0                              # continuous schedule: index of the next ODE evaluation of the modality being decoded
    som_pending: bool = False                   # continuous schedule: the null-text cache still lacks the [som] that opened the current modality
    commit_pending: bool = False                # continuous schedule: the null-text cache still lacks the modality just decoded (at t = 1)


def _sample_text_token(logits, V, temperature, min_p, stream):
    """sample_text_token (T:597-605) incl. min_p_filter (T:591-595) on the device: `tfx_sample_tokens` over the rows of a 2-d fp32
    logits tensor (V valid columns).  Returns int32 ids on the device."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B = logits.shape[0]
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    u = torch.rand(B, device=logits.device) if temperature != 0. else None
    capi.check(capi.lib().tfx_sample_tokens(logits.data_ptr(), logits.stride(0), B, V, float(temperature), float(min_p), capi.ptr(u), None,
                                            out.data_ptr(), ctypes.c_void_p(stream)), 'tfx_sample_tokens')
    return out


def _pick_text_only(logits, V, N, temperature, min_p, stream):
    """the draw of `generate_text_only` (T:2690-2698) on the device: temperature 0 -> argmax over ALL V logits; else the min-p filter over all V
    logits (threshold relative to the global maximum), then the text-only mask (first N columns), then the draw."""
    assert logits.dim() == 2 and logits.dtype == torch.float32 and logits.stride(1) == 1
    B = logits.shape[0]
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    u = torch.rand(B, device=logits.device) if temperature != 0. else None
    capi.check(capi.lib().tfx_sample_tokens_range(logits.data_ptr(), logits.stride(0), B, V, N if temperature != 0. else V, float(temperature), float(min_p),
                                                  capi.ptr(u), None, out.data_ptr(), ctypes.c_void_p(stream)), 'tfx_sample_tokens_range')
    return out.long()


def _ode_axpy(y, f_cond, f_uncond, cfg_scale, a, stream):
    """out = y + a * (f_uncond + cfg_scale * (f_cond - f_uncond)) (or y + a * f_cond): one fused launch (tfx_ode_axpy)"""
    out = torch.empty_like(y)
    capi.check(capi.lib().tfx_ode_axpy(y.data_ptr(), f_cond.data_ptr(), capi.ptr(f_uncond), float(cfg_scale), float(a), out.data_ptr(), y.numel(),
                                       ctypes.c_void_p(stream)), 'tfx_ode_axpy')
    return out


# continuous schedule: compacted mixed steps (1) instead of the dense (L + 1)-rows-per-sample layout.  Built, token-identical (tests/test_sampling_gpu.py runs
# both) - and OFF by default: config 5 measured 1.630 s against 1.646 s (same box).  A decode step's time is flat in its row count between ~256 and 640 rows
# (tools/bench_decode_step.py: 2.9 - 3.3 ms for 128 samples x 2 / 3 / 5 rows, 2.3 ms at 128 x 1): it is ~225 dependent launches of 6 - 20 us each, set by
# launch ramps and first-load latencies, not by rows - the rows were never what a step paid for.
_COMPACT = os.environ.get('TFX_DECODE_COMPACT', '0') == '1'
_COMPACT_STEP = int(os.environ.get('TFX_DECODE_COMPACT_STEP', '128'))     # row-count bucket of the compacted plans (a plan = buffers + launch list + hipGraph per bucket)


class Sampler:
    def __init__(self, model):
        self.m = model
        self.dev = model.device
        self.md = model.md

    # ------------------------------------------------------------------ prompt handling (T:1701-1800)
    def _meta_tokens(self, shape, ty):
        m = self.m
        s = ','.join(map(str, shape))
        return [m.meta_id] + [ord(c) + m.meta_id + 1 for c in s] + [m.som_ids[ty]]

    def _prepare(self, prompt, force_modality_at_start):
        m = self.m
        if torch.is_tensor(prompt) and prompt.is_floating_point():
            prompt = (0, prompt)
        if torch.is_tensor(prompt):                                   # text only prompt
            prompt = [prompt]
        elif isinstance(prompt, tuple):
            ty, mod = prompt
            shape = tuple(mod.shape[:-1])
            prompt = [self._meta_tokens(shape, ty), (ty, mod), [m.eom_ids[ty]]]
        elif prompt is None:
            prompt = []
        prompt = [p for p in prompt if p is not None]
        if prompt and isinstance(prompt[-1], tuple):                  # a raw modality ending the prompt is closed with its [eom]
            prompt.append([m.eom_ids[prompt[-1][0]]])
        parts = [[m.sos_id]]
        for p in prompt:
            if isinstance(p, tuple):
                parts.append((p[0], p[1].to(self.dev, torch.float32)))
                continue
            ids = p if isinstance(p, list) else [int(v) for v in torch.atleast_1d(p).reshape(-1).tolist()]
            if isinstance(parts[-1], list):                           # concat_contiguous_text, T:204-222
                parts[-1] = parts[-1] + ids
            else:
                parts.append(list(ids))
        forced_id, forced_shape = force_modality_at_start if isinstance(force_modality_at_start, tuple) else (force_modality_at_start, None)
        if forced_id is not None:                                     # maybe_force_modality_at_start, T:1701-1731
            toks = self._meta_tokens(forced_shape, forced_id) if forced_shape is not None else [m.som_ids[forced_id]]
            if isinstance(parts[-1], list):
                parts[-1] = parts[-1] + toks
            else:
                parts.append(toks)
        return parts, forced_id, forced_shape

    def _shape_from_seq(self, seq, ty, fixed_shape, forced_shape):    # get_modality_shape_from_seq, T:1607-1648
        m = self.m
        shape = forced_shape if forced_shape is not None else fixed_shape
        default_shape = m.modality_default_shape[ty]
        if m.meta_id in seq and forced_shape is None:
            after = seq[len(seq) - 1 - seq[::-1].index(m.meta_id) + 1:]
            meta = after[:-1]
            meta_str = ''.join(chr(min(max(t - (m.meta_id + 1), 0), 127)) for t in meta)
            if len(after) > 0:
                if not meta_str.isdigit() or int(meta_str) <= 0:
                    assert default_shape is not None, 'invalid modality meta information detected, please set `modality_default_shape` in order to properly fallback'
                    shape = default_shape
                else:
                    shape = m.to_modality_shape_fn[ty](meta_str)
        shape = shape if shape is not None else default_shape
        ndim = m.modality_num_dim[ty]
        if m.fallback_to_default_shape_if_invalid and ndim is not None and len(shape) != ndim:
            shape = default_shape
        assert shape is not None, f'language model did not produce a proper modality shape for modality type {ty} - please set a fallback shape with `modality_default_shape`'
        assert ndim is None or ndim == len(shape), f'expected modality type {ty} to have {ndim} dimensions but language model produced a shape of {shape}'
        return tuple(shape)

    def _maybe_transition(self, st, fixed_shape):                     # T:2203-2224 / get_modality_transition T:1802-1825
        m = self.m
        tok = st.curr_seq[-1]
        if tok not in m.som_ids:
            return False
        ty = m.som_ids.index(tok)
        forced_id, forced_shape = st.forced
        shape = self._shape_from_seq(st.curr_seq, ty, fixed_shape, forced_shape if ty == forced_id else None)
        st.curr_modality_id, st.modality_shape, st.modality_length, st.phase = ty, shape, math.prod(shape), 'modality'
        return True

    def _as_batch(self, parts_list, null_text=False):
        """host parts -> the list-of-tensors batch format the packer takes."""
        out = []
        for parts in parts_list:
            s = []
            for p in parts:
                if isinstance(p, tuple):
                    s.append(p)
                else:
                    ids = [self.m.null_text_id] * len(p) if null_text else p
                    s.append(torch.tensor(ids, dtype=torch.long, device=self.dev))
            out.append(s)
        return out

    # ------------------------------------------------------------------ caches
    def _alloc_cache(self, B, maxlen):
        return torch.zeros(self.md.depth, B, maxlen, 2 * self.md.hdk, device=self.dev, dtype=torch.bfloat16)

    def _fill_cache(self, cache, plan, B, n):
        D, hd, ldq = self.md.depth, self.md.hdk, self.md.ldq
        cache[:, :, :n, :hd].copy_(plan.qkr.view(D, B, n, 2 * hd)[..., hd:])
        cache[:, :, :n, hd:].copy_(plan.qkvg.view(D, B, n, ldq)[..., 2 * hd:3 * hd])

    def _decode_plan(self, key, B, Lq, cache, with_latents, n_inst=None, tile_attn=False, units=None):
        plans = self.m._decode_plans
        if key not in plans:
            R = {t: B * Lq for t in range(self.m.num_modalities)} if with_latents else {}
            self.m.store.refresh_shadows(self.m._stream())
            p = Plan(self.m.store, B, Lq, (n_inst or B) if with_latents else 0, R, training=False, cache=cache, tile_attn=tile_attn, units=units)
            p.q_start.zero_()
            plans[key] = p
        return plans[key]

    # ------------------------------------------------------------------ main
    def sample_many(self, prompts, max_length, text_temperature, text_min_p, fixed_modality_shape, force_modality_at_start,
                    init_modality_noise, modality_steps, cfg_scale, pos_emb_in_decode=False):
        m, md, dev = self.m, self.md, self.dev
        self.pos_emb_in_decode = pos_emb_in_decode
        m._require_gpu()
        if prompts is None:
            prompts = [None]
        elif not isinstance(prompts, list):
            prompts = [prompts]
        states = []
        for prompt in prompts:
            parts, forced_id, forced_shape = self._prepare(prompt, force_modality_at_start)
            seq_len = collapse = past = 0
            for p in parts:
                if isinstance(p, tuple):
                    L = math.prod(p[1].shape[:-1]); seq_len += L; collapse += L - 1; past += 1
                else:
                    seq_len += len(p)
            last = parts[-1]
            st = _State(parts=parts, curr_seq=last if isinstance(last, list) else [m.sos_id],
                        last_token=last[-1] if isinstance(last, list) else None, forced=(forced_id, forced_shape))
            st.tokens_seen, st.num_past_modalities, st.cache_len = seq_len - collapse, past, seq_len
            states.append(st)
        B = len(states)

        # ---- batched prefill (T:2194-2201): prompted modalities conditioned at t = 1, no meta tokens re-added
        max_past = max((s.num_past_modalities for s in states), default=0)
        times = torch.ones(B, max(max_past, 1), device=dev)
        plan, S = m._forward_plain(self._as_batch([s.parts for s in states]), times, add_meta=False)
        n0 = S['n']
        Lcap = max([math.prod(sh) for sh in m.modality_default_shape if sh is not None] + [math.prod(fixed_modality_shape or (1,)), 16])
        maxlen = (max(s.cache_len for s in states) + max_length + 2 * Lcap + 80) // 64 * 64
        # classifier-free guidance evaluates every ODE step twice - against the real history and against the null-text one (T:2468-2525).  Both
        # evaluations share the weights and a decode step is launch-bound, so they run as ONE forward over 2 B rows: the two KV caches are the halves
        # of one buffer (the text steps use the first half through a view)
        use_cfg = cfg_scale != 1.
        # Decode plans - launch lists, their captured graphs, per-plan AdaLN tables of the solver's time grid - are bound to the cache buffer they were
        # built on.  A serving process calls sample_many over and over with the same geometry: the buffer and its plans are KEPT on the model between
        # calls (plan build + graph capture were ~30 ms of every call) and reused while everything they froze still holds - rows,
        # capacity, solver grid, guidance on / off, the parameter version (shadows and cond tables are functions of the weights).  TFX_DECODE_KEEP=0
        # turns it off; TFX_DECODE_KEEP_GB bounds the kept cache (default 16); `model.train()` drops it.
        rows, need = (2 * B if use_cfg else B), max(maxlen, n0)
        keep_on = os.environ.get('TFX_DECODE_KEEP', '1') != '0'
        keep_key = (rows, int(modality_steps), bool(use_cfg), m.store.params_version(), tuple(fixed_modality_shape or ()), bool(pos_emb_in_decode))
        ent = getattr(m, '_decode_keep', None) if keep_on else None
        self._grew = False
        if ent is not None and ent['key'] == keep_key and ent['joint'].shape[2] >= need:
            joint, m._decode_plans = ent['joint'], ent['plans']
            joint.zero_()
        else:
            m._decode_keep = None
            m._decode_plans = {}
            joint = self._alloc_cache(rows, need)
        joint0 = joint
        cache = joint[:, :B]
        self._fill_cache(cache, plan, B, n0)
        logits0 = plan.logits.view(B, n0, md.vp)[..., :md.vocab]
        for st in states:
            self._maybe_transition(st, fixed_modality_shape)
        stream = m._stream()
        last_rows = torch.tensor([i * n0 + s.cache_len - 1 for i, s in enumerate(states)], device=dev)
        first = _sample_text_token(plan.logits.index_select(0, last_rows), md.vocab, text_temperature, text_min_p, stream).tolist()
        for st, tok in zip(states, first):
            if st.phase == 'text':
                st.curr_seq.append(tok); st.last_token = tok; st.num_tokens += 1; st.unfed = True
                if isinstance(st.parts[-1], list) and st.parts[-1] is not st.curr_seq:
                    st.parts[-1] = st.curr_seq
                if tok == m.eos_id:
                    st.phase = 'done'; continue
                self._maybe_transition(st, fixed_modality_shape)
            if st.num_tokens > max_length:
                st.phase = 'done'

        # Two schedules over the same per-sample state machine (samples only ever attend to their own cache, so the schedule cannot change
        # what a sample decodes - the reference's own test asserts sample_many == per-prompt sample_one):
        #   continuous (default)  every global step advances EVERY live sample - by a text token or by one ODE evaluation - in one mixed forward
        #   phased                the reference's loop (T:2226-2360): text steps until no sample is in its text phase, then one joint ODE
        # A mixed step carries (modality length + 1) rows for EVERY sample, a text step of the phased loop one: continuous batching pays while a
        # forward is launch-bound (its time does not depend on the row count) - up to a few thousand rows.  Long blocks (images) in a large batch
        # keep the phased schedule: there a mixed step would cost as much as a joint ODE evaluation of the whole batch.
        schedule = os.environ.get('TFX_SAMPLE_SCHEDULE', 'auto')
        if schedule == 'auto':
            est = [math.prod(sh) for sh in m.modality_default_shape if sh is not None] + ([math.prod(fixed_modality_shape)] if fixed_modality_shape else [])
            rows = (2 if use_cfg else 1) * B * (max(est, default=4) + 1)
            schedule = 'continuous' if rows <= 4096 else 'phased'
        if schedule != 'phased':
            self._loop_continuous(states, joint, maxlen, use_cfg, stream, max_length, text_temperature, text_min_p, fixed_modality_shape,
                                  init_modality_noise, modality_steps, cfg_scale)
        else:
            self._loop_phased(states, joint, use_cfg, stream, max_length, text_temperature, text_min_p, fixed_modality_shape,
                              init_modality_noise, modality_steps, cfg_scale)
        gib = joint0.numel() * joint0.element_size() / 2 ** 30
        if keep_on and not self._grew and gib <= float(os.environ.get('TFX_DECODE_KEEP_GB', '16')):
            m._decode_keep = {'key': keep_key, 'joint': joint0, 'plans': m._decode_plans}      # (a run that had to grow its cache rebuilt its plans: not kept)
        else:
            m._decode_keep = None
        m._decode_plans = {}
        return [[(p if isinstance(p, tuple) else torch.tensor(p, dtype=torch.long, device=dev)) for p in st.parts] for st in states]

    def _pinned(self, name, shape, dtype):
        """pinned host staging buffers, a ring of 8 per (name, shape): the per-step control arrays go up as asynchronous copies"""
        ring = self.__dict__.setdefault('_stage2', {})
        key = (name, tuple(shape), dtype)
        if key not in ring:
            ring[key] = [[torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(8)], 0]
        bufs, k = ring[key]
        ring[key][1] = (k + 1) % len(bufs)
        return bufs[k]

    def _loop_continuous(self, states, joint, maxlen, use_cfg, stream, max_length, text_temperature, text_min_p, fixed_modality_shape,
                         init_modality_noise, modality_steps, cfg_scale):
        """Continuous batching of the decode loop.  The reference's schedule (and `_loop_phased`) makes the samples that reached a [som] wait until
        the slowest writer of the round has reached its own, then decodes all waiting modalities in one joint ODE: with B desynchronised samples the
        text phase runs sum over rounds of max_i(text run) steps.  Here every global step is ONE forward over (1 | 2) B x Lq rows in which each live
        sample advances by what it needs: a text sample feeds its last token (one row), a sample inside a modality runs its next ODE evaluation
        (L rows at its own step time - the fixed-grid midpoint method as a per-sample state machine on the device), and the null-text half of
        classifier-free guidance stays in lock-step - the null id of the same token, the same evaluation against its own cache, and, right
        after a modality is finished, that block once more at t = 1 (what the reference's re-prefill of the null-text history computes, T:2386-2406).
        The run takes max_i(tokens_i + evaluations_i) steps instead.  Row roles and cache positions follow `_load_modality` / `_uncond_append`."""
        m, md, dev = self.m, self.md, self.dev
        B = len(states)
        H = 2 if use_cfg else 1
        nb = H * B
        M, dmax = m.num_modalities, max(md.dim_latents)
        ts = torch.linspace(0, 1, modality_steps)
        evals = []                                                          # (time, coefficient, sub-step): torchdiffeq fixed-grid midpoint
        for k in range(modality_steps - 1):
            t0, dt = float(ts[k]), float(ts[k + 1] - ts[k])
            evals += [(t0, dt * 0.5, 1), (t0 + dt * 0.5, dt, 2)]
        # The time conditioning of a decode step - Fourier features -> MLP -> every layer's AdaLN tables, a GEMM that streams ALL the conditioning
        # weights (1.2 GB at dim 1024 / depth 24: 0.5 ms of a 4 ms step) - depends on the step time alone, and a fixed-grid solver only ever asks
        # for 2 (S - 1) times plus t = 1 (finished blocks).  The "instances" of the mixed plan are therefore those TIMES: their tables are computed
        # once per plan, a token's instance index says which row it reads, and the conditioning launches drop out of the per-step replay.
        cond_times = torch.tensor([e[0] for e in evals] + [1.], dtype=torch.float32, device=dev)
        n_t = cond_times.numel()
        tm = None                                       # (host-side phase timing dict for debugging: set to {'steps': 0, 'mixed_steps': 0, 'prefill': 0., 'loop': 0., 'host_build': 0., 'host_issue': 0., 'wait': 0., 'host_post': 0.})
        t_a = time.perf_counter()
        if use_cfg:
            # null-text twin of the prefill: every token that went through the model so far (the sampled-but-unfed last token excluded)
            hist = []
            for st in states:
                parts = [list(p) if isinstance(p, list) else p for p in st.parts]
                if st.unfed:
                    parts[-1] = parts[-1][:-1]
                hist.append([p for p in parts if isinstance(p, tuple) or len(p)])
            past = max(max((s.num_past_modalities for s in states), default=0), 1)
            uplan, US = m._forward_plain(self._as_batch(hist, null_text=True), torch.ones(B, past, device=dev), add_meta=False)
            if joint.shape[2] < US['n']:
                joint = self._grow(joint, US['n'])
            self._fill_cache(joint[:, B:], uplan, B, US['n'])
            for st in states:
                st.uncond_len, st.uncond_pos = st.cache_len, st.tokens_seen
        if tm is not None:
            torch.cuda.synchronize(); tm['prefill'] = time.perf_counter() - t_a; t_a = time.perf_counter()

        Lc = 0
        Y = Ym = None
        multi = M > 1
        sel = torch.zeros(M, B, device=dev) if multi else None              # per type: which samples are decoding a block of that type
        lib = capi.lib()
        sp = ctypes.c_void_p(stream)
        async_steps = 0
        # host mirror of the per-sample state, one column per sample (the step's index arrays are built from it with array operations):
        PH, CL, UL, TS, UP, LT, LL, TY, OK, SP, CP = range(11)              # phase (0 done / 1 text / 2 modality), cache_len, uncond_len, tokens_seen, uncond_pos,
        A = np.zeros((11, B), np.int64)                                     # last_token, modality length, type, ode_k, som_pending, commit_pending
        code = {'done': 0, 'text': 1, 'modality': 2}

        def mirror(i, st):
            A[:, i] = (code[st.phase], st.cache_len, st.uncond_len, st.tokens_seen, st.uncond_pos, st.last_token if st.last_token is not None else 0,
                       st.modality_length or 0, st.curr_modality_id or 0, st.ode_k, st.som_pending, st.commit_pending)

        def begin_modality(i, st):
            nonlocal Lc, Y, Ym
            L, dl = st.modality_length, md.dim_latents[st.curr_modality_id]
            if L > Lc:
                newY, newYm = torch.zeros(B, L, dmax, device=dev), torch.zeros(B, L, dmax, device=dev)
                if Y is not None:
                    newY[:, :Lc], newYm[:, :Lc] = Y, Ym
                Y, Ym, Lc = newY, newYm, L
            Y[i, :L, :dl] = init_modality_noise[:L, :dl].to(dev) if init_modality_noise is not None else torch.randn(L, dl, device=dev)
            st.ode_k, st.som_pending = 0, bool(use_cfg and st.unfed)
            if multi:
                sel[:, i] = 0.; sel[st.curr_modality_id, i] = 1.

        for i, st in enumerate(states):
            if st.phase == 'modality':
                begin_modality(i, st)
            mirror(i, st)
        ar_b = np.arange(B, dtype=np.int64)
        modes = np.array([e[2] for e in evals], np.float32); coefs = np.array([e[1] for e in evals], np.float32)

        while A[PH].any():
            t_0 = time.perf_counter()
            is_t, is_m = A[PH] == 1, A[PH] == 2
            com = is_t & (A[CP] > 0)
            mixed = bool(is_m.any() or com.any())
            Lq = Lc + 1 if mixed else 1
            need = int(max(A[CL].max(), A[UL].max())) + Lq + 2
            if need > joint.shape[2]:
                joint = self._grow(joint, need + 192)
            cap = joint.shape[2]
            compact = mixed and _COMPACT
            if compact:
                # COMPACTED mixed step: the step carries exactly the rows its live samples need - a block's L rows for a sample inside a modality (and for
                # the null-text twin of a block just finished), one row for a text token / a pending [som] - packed back to back, unit (half h, sample i)
                # = cache row h B + i owning rows unit_row0 .. + unit_cnt of the plan (Plan `units`; the attention kernel reads its query rows through
                # them, tfx_attn_args.q_row0 / q_cnt).  The dense layout below spends (L + 1) rows on EVERY sample, done or not: 640 rows per step at
                # config 5 against ~270 needed.  Plans come in row-count buckets of `_COMPACT_STEP`.
                U = nb
                cap_ = cap
                blk = np.zeros((H, B), bool); txt = np.zeros((H, B), bool)
                boff = np.zeros((H, B), np.int64); tbase = np.zeros((H, B), np.int64); trot = np.zeros((H, B), np.int64)
                rblk = np.zeros((H, B), np.int64); iblk = np.zeros((H, B), np.int64); itxt = np.zeros((H, B), np.int64)
                for h in range(H):
                    base = (A[UL] if h else A[CL])
                    txt_first = is_m & (A[SP] > 0) if h else np.zeros(B, bool)
                    blk_first = com if h else np.zeros(B, bool)
                    blk[h] = is_m | blk_first
                    boff[h] = base + txt_first
                    rblk[h] = np.where(is_m, A[TS], A[UP])
                    iblk[h] = np.where(is_m, A[OK], n_t - 1)
                    txt[h] = is_t | txt_first
                    tbase[h] = base + np.where(blk_first, A[LL], 0)
                    trot[h] = np.where(is_t, (A[UP] + blk_first) if h else A[TS], A[UP])
                    itxt[h] = m.null_text_id if h else A[LT]
                LLu = np.broadcast_to(A[LL], (H, B))
                nblk = np.where(blk, LLu, 0).reshape(-1)
                cnt = nblk + txt.reshape(-1)
                row0 = np.cumsum(cnt) - cnt
                Ts = int(cnt.sum())
                Tb = max(_COMPACT_STEP, -(-Ts // _COMPACT_STEP) * _COMPACT_STEP)
                p = self._decode_plan(('mixc', nb, Tb, Lq, joint.data_ptr()), Tb, 1, joint, True, n_inst=n_t, tile_attn=True, units=(U, Lq))
                T = Tb
                u = np.repeat(np.arange(U, dtype=np.int64), cnt)
                j = np.arange(Ts, dtype=np.int64) - row0[u]
                isb = j < nblk[u]
                f = lambda a: a.reshape(-1)[u]
                ids = np.zeros(Tb, np.int32); pos = np.full(Tb, -1, np.int64); kve = np.ones(Tb, np.int64); rot = np.zeros(Tb, np.int64); tok_inst = np.full(Tb, -1, np.int64)
                ids[:Ts] = np.where(isb, 0, f(itxt))
                pos[:Ts] = u * cap_ + np.where(isb, f(boff) + j, f(tbase))
                kve[:Ts] = np.where(isb, f(boff) + f(LLu), f(tbase) + 1)
                rot[:Ts] = np.where(isb, f(rblk), f(trot))
                tok_inst[:Ts] = np.where(isb, f(iblk), -1)
                blk_rows = np.zeros(Tb, bool); blk_rows[:Ts] = isb
                row_ty = np.zeros(Tb, np.int64); row_ty[:Ts] = np.broadcast_to(A[TY], (H, B)).reshape(-1)[u]
                unit_arrays = np.zeros(4 * U, np.int32)
                unit_arrays[:U] = row0; unit_arrays[U:2 * U] = cnt
                unit_arrays[2 * U:3 * U] = np.where(blk.reshape(-1), row0, -1)
                unit_arrays[3 * U:3 * U + B] = np.where(txt[0], row0[:B] + nblk[:B], 0)
                t_1 = time.perf_counter()
                self._load(p, ids, pos, kve, rot, tok_inst, units=unit_arrays)
            else:
                # one attention kernel - the tiled forward kernel - for every plan of the run: a token's arithmetic must not depend on which plan carried it
                p = self._decode_plan(('mix' if mixed else 'txt', nb, Lq, joint.data_ptr()), nb, Lq, joint, mixed, n_inst=n_t, tile_attn=True)
                T = nb * Lq
                ids = np.zeros((H, B, Lq), np.int32); pos = np.full((H, B, Lq), -1, np.int64); kve = np.ones((H, B, Lq), np.int64)
                rot = np.zeros((H, B, Lq), np.int64); tok_inst = np.full((H, B, Lq), -1, np.int64)
                blk_any = np.zeros((H, B, Lq), bool)
                jj = np.arange(Lq, dtype=np.int64)[None, :]
                in_blk = jj < A[LL][:, None]                                    # columns a block of this sample's modality length would take
                for h in range(H):
                    r = (h * B + ar_b)[:, None]
                    base = (A[UL] if h else A[CL])
                    kve[h] = np.maximum(base, 1)[:, None]
                    txt_first = is_m & (A[SP] > 0) if h else np.zeros(B, bool)  # null-text half: the [som] the real history never feeds (T:2411), ahead of the block
                    blk_first = com if h else np.zeros(B, bool)                 # null-text half: the block just decoded, prompt-style (t = 1), ahead of the next token
                    has_blk = (is_m | blk_first)[:, None] & in_blk
                    boff = (base + txt_first)[:, None]
                    pos[h] = np.where(has_blk, r * cap + boff + jj, -1)
                    kve[h] = np.where(has_blk, boff + A[LL][:, None], kve[h])
                    rot[h] = np.where(has_blk, np.where(is_m, A[TS], A[UP])[:, None], 0)          # an ODE block sits at tokens_seen in BOTH halves; a finished one at its history position
                    tok_inst[h] = np.where(has_blk, np.where(is_m, A[OK], n_t - 1)[:, None], -1)   # instance = index of the conditioning time
                    blk_any[h] = has_blk
                    has_txt = is_t | txt_first
                    tbase = base + np.where(blk_first, A[LL], 0)
                    trot = np.where(is_t, (A[UP] + blk_first) if h else A[TS], A[UP])
                    ids[h, :, -1] = np.where(has_txt, m.null_text_id if h else A[LT], 0)
                    pos[h, :, -1] = np.where(has_txt, r[:, 0] * cap + tbase, pos[h, :, -1])
                    kve[h, :, -1] = np.where(has_txt, tbase + 1, kve[h, :, -1])
                    rot[h, :, -1] = np.where(has_txt, trot, rot[h, :, -1])
                t_1 = time.perf_counter()
                self._load(p, ids.reshape(-1), pos.reshape(-1), kve.reshape(-1), rot.reshape(-1), tok_inst.reshape(-1))
            if mixed:
                if not getattr(p, '_cont_ready', False):
                    for t in range(M):
                        p.row_inst[t].zero_()
                        p.set_noise(t, None)
                    p.inst_time.copy_(cond_times)
                    Plan.run(p.fwd, stream, *p.fwd_cond)                      # the AdaLN tables of every time the solver will ask for, once
                    p._cont_ready = True
                # row maps of all types + the solver control block: one staging buffer, one copy
                hb = self._pinned('rowbuf', (p.rowbuf.numel(),), torch.int32)
                hv = hb.numpy()
                rs = p.row_stride
                rowidx = np.arange(T, dtype=np.int32)
                flat_blk = blk_rows if compact else blk_any.reshape(-1)
                for t in range(M):
                    if compact:
                        mine = flat_blk if M == 1 else flat_blk & (row_ty == t)
                    else:
                        mine = flat_blk if M == 1 else (blk_any & (A[TY] == t)[None, :, None]).reshape(-1)
                    hv[t * rs:t * rs + T] = np.where(mine, rowidx, -1)
                    hv[(M + t) * rs:(M + t) * rs + T] = np.where(mine, rowidx, 0)
                ctl = hv[2 * M * rs:].view(np.float32)
                kk = np.minimum(A[OK], len(evals) - 1)
                ctl[:B] = np.where(is_m, modes[kk], np.where(com, 3., 0.))
                ctl[B:2 * B] = np.where(is_m, coefs[kk], 0.)
                p.rowbuf.copy_(hb, non_blocking=True)
                if md.model_output_clean:                                   # the clean-prediction conversion reads a row's time through its instance
                    for t in range(M):
                        p.row_inst[t].copy_(p.tok_inst.clamp(min=0))
                for t in range(M):
                    dl = md.dim_latents[t]
                    capi.check(lib.tfx_ode_stage(Y.data_ptr(), Ym.data_ptr(), p.ctl.data_ptr(), B, Lc, dmax, p.lat[t]['x'].data_ptr(), H, Lq, dl, p.unit_blk0.data_ptr() if compact else None, sp), 'tfx_ode_stage')
                for t in p.ext_add:                                         # blocks that carry their axial positional embedding (history-style always, T:3173-3176)
                    add = p.lat[t]['add']
                    add.zero_()
                    for i in np.flatnonzero((is_m & bool(getattr(self, 'pos_emb_in_decode', False))) | com):
                        if states[i].curr_modality_id != t:
                            continue
                        rows = m._pos_rows(t, [states[i].modality_shape])
                        for h in range(H) if is_m[i] else [1]:
                            lo = int(row0[h * B + i]) if compact else (h * B + i) * Lq
                            add[lo:lo + rows.shape[0]].copy_(rows)
                self._run(p, stream, 0, p.fwd_cond[0])
                self._run(p, stream, p.fwd_cond[1], p.fwd_pred_end)
                for t in range(M):
                    dl = md.dim_latents[t]
                    capi.check(lib.tfx_ode_update(Y.data_ptr(), Ym.data_ptr(), p.ctl.data_ptr(), B, Lc, dmax, p.lat[t]['pred'].data_ptr(), H, Lq, dl,
                                                  float(cfg_scale), sel[t].data_ptr() if multi else None, p.unit_blk0.data_ptr() if compact else None, sp), 'tfx_ode_update')
            else:
                self._run(p, stream, 0, p.fwd_logits_end)
            t_2 = time.perf_counter()
            toks = None
            if is_t.any():
                lg = p.logits.view(T, md.vp).index_select(0, p.unit_txt[:B]) if compact else p.logits.view(nb, Lq, md.vp)[:B, Lq - 1]
                toks = _sample_text_token(lg, md.vocab, text_temperature, text_min_p, stream).tolist()      # host sync
                async_steps = 0
            else:
                async_steps += 1
                if async_steps >= 4:                                        # the staging rings are 8 deep: never run further ahead of the device
                    torch.cuda.current_stream(dev).synchronize(); async_steps = 0
            t_3 = time.perf_counter()
            for i in np.flatnonzero(A[PH]):
                st = states[i]
                if st.phase == 'text':
                    tok = toks[i]
                    if use_cfg:
                        if st.commit_pending:
                            st.uncond_len += st.modality_length; st.uncond_pos += 1; st.commit_pending = False
                        st.uncond_len += 1; st.uncond_pos += 1
                    st.cache_len += 1
                    st.curr_seq.append(tok); st.last_token = tok; st.tokens_seen += 1; st.num_tokens += 1; st.unfed = True
                    if tok == m.eos_id or st.num_tokens > max_length:
                        st.phase = 'done'
                    elif self._maybe_transition(st, fixed_modality_shape):
                        begin_modality(i, st)
                else:
                    if st.som_pending:
                        st.uncond_len += 1; st.uncond_pos += 1; st.som_pending = False
                    st.ode_k += 1
                    if st.ode_k == len(evals):                                # commit, T:2531-2556
                        L, ty = st.modality_length, st.curr_modality_id
                        dl = md.dim_latents[ty]
                        st.cache_len += L
                        st.parts.append((ty, Y[i, :L, :dl].reshape(*st.modality_shape, dl).clone()))
                        st.curr_seq = [m.eom_ids[ty]]; st.parts.append(st.curr_seq); st.last_token = m.eom_ids[ty]; st.unfed = True
                        st.tokens_seen += 1; st.num_tokens += L; st.num_past_modalities += 1
                        st.phase = 'done' if st.num_tokens > max_length else 'text'
                        st.commit_pending = bool(use_cfg and st.phase == 'text')
                mirror(i, st)
            if tm is not None:
                tm['steps'] += 1; tm['mixed_steps'] += int(mixed)
                tm['host_build'] += t_1 - t_0; tm['host_issue'] += t_2 - t_1; tm['wait'] += t_3 - t_2; tm['host_post'] += time.perf_counter() - t_3
        if tm is not None:
            torch.cuda.synchronize(); tm['loop'] = time.perf_counter() - t_a
            print('[TFX_SAMPLE_TIMING]', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})

    def _loop_phased(self, states, joint, use_cfg, stream, max_length, text_temperature, text_min_p, fixed_modality_shape,
                     init_modality_noise, modality_steps, cfg_scale):
        m, md, dev = self.m, self.md, self.dev
        B = len(states)
        cache = joint[:, :B]
        tm = None                                       # (host-side phase timing dict for debugging: set to {'text': 0., 'text_steps': 0, 'uncond_prefill': 0., 'ode': 0., 'phases': 0})
        def mark():
            if tm is None:
                return 0.
            torch.cuda.synchronize(); return time.perf_counter()
        while not all(s.phase == 'done' for s in states):
            # ------------------------------------------------ text phase
            t_a = mark()
            while any(s.phase == 'text' for s in states):
                if tm is not None: tm['text_steps'] += 1
                joint = self._ensure_capacity(joint, states, 1); cache = joint[:, :B]
                p = self._decode_plan(('text', cache.data_ptr()), B, 1, cache, False)
                ids = np.zeros(B, np.int32); pos = np.full(B, -1, np.int32); kve = np.ones(B, np.int32); rot = np.zeros(B, np.int32)
                for i, st in enumerate(states):
                    kve[i] = max(st.cache_len, 1)
                    if st.phase == 'text':
                        ids[i], pos[i], kve[i], rot[i] = st.last_token, i * cache.shape[2] + st.cache_len, st.cache_len + 1, st.tokens_seen
                self._load(p, ids=ids, pos=pos, kve=kve, rot=rot, tok_inst=np.full(B, -1, np.int32))
                self._run(p, stream, 0, p.fwd_logits_end)
                toks = _sample_text_token(p.logits, md.vocab, text_temperature, text_min_p, stream).tolist()
                for st, tok in zip(states, toks):
                    if st.phase != 'text':
                        continue
                    st.cache_len += 1
                    st.curr_seq.append(tok); st.last_token = tok; st.tokens_seen += 1; st.num_tokens += 1
                    if tok == m.eos_id or st.num_tokens > max_length:
                        st.phase = 'done'; continue
                    self._maybe_transition(st, fixed_modality_shape)
            # ------------------------------------------------ modality phase
            t_b = mark()
            if tm is not None: tm['text'] += t_b - t_a
            group = [i for i, s in enumerate(states) if s.phase == 'modality']
            if not group:
                continue
            Lmax = max(states[i].modality_length for i in group)
            joint = self._ensure_capacity(joint, states, Lmax); cache = joint[:, :B]
            if use_cfg:                                               # null-text history, T:2386-2406
                flat_len = lambda parts: sum(math.prod(p[1].shape[:-1]) if isinstance(p, tuple) else len(p) for p in parts)
                incremental = os.environ.get('TFX_UNCOND_INCREMENTAL', '1') != '0' and all(states[i].uncond_parts > 0 for i in group)
                if incremental:
                    # the reference re-runs the whole null-text history before every modality (T:2386-2406).  Its keys / values for the part
                    # already in the cache do not change, so only what each sample appended since its last modality - the decoded block
                    # (conditioned at t = 1) and the text after it - goes through the model, as ONE multi-token decode step against the cache
                    need = max(flat_len(states[i].parts) for i in group) + Lmax + 8
                    if joint.shape[2] < need:
                        joint = self._grow(joint, need); cache = joint[:, :B]
                    self._uncond_append(states, group, joint[:, B:], stream)
                else:
                    hist = [states[i].parts if i in group else [[m.null_text_id]] for i in range(B)]
                    past = max(max(states[i].num_past_modalities for i in group), 1)
                    uplan, US = m._forward_plain(self._as_batch(hist, null_text=True), torch.ones(B, past, device=dev), add_meta=False)
                    un = US['n']
                    if joint.shape[2] < un + Lmax + 8:
                        joint = self._grow(joint, un + Lmax + 8); cache = joint[:, :B]
                    self._fill_cache(joint[:, B:], uplan, B, un)
                    for i in group:
                        st = states[i]
                        st.uncond_len, st.uncond_parts = flat_len(st.parts), len(st.parts)
                        st.uncond_pos = sum(1 if isinstance(p, tuple) else len(p) for p in st.parts)       # a modality block takes one rotary position (T:398-415)
            t_c = mark()
            if tm is not None: tm['uncond_prefill'] += t_c - t_b; tm['phases'] += 1
            y = torch.zeros(B, Lmax, max(md.dim_latents), device=dev)
            for i in group:
                st = states[i]; L, dl = st.modality_length, md.dim_latents[st.curr_modality_id]
                y[i, :L, :dl] = init_modality_noise[:L, :dl].to(dev) if init_modality_noise is not None else torch.randn(L, dl, device=dev)
            nb = 2 * B if use_cfg else B
            jp = self._decode_plan(('mod', nb, Lmax, joint.data_ptr()), nb, Lmax, joint, True)
            self._load_modality(jp, states, group, Lmax, joint.shape[2], halves=2 if use_cfg else 1)
            type_mask = self._type_masks(states, group, Lmax)
            if use_cfg:
                type_mask = {ty: torch.cat([mk, mk]) for ty, mk in type_mask.items()}

            def step(t, y_eval, y_base, a):
                """y_base + a * velocity(t, y_eval): the conditional and the null-text evaluation in one forward (rows [0, B) and [B, 2B)),
                guidance + the state update fused (tfx_ode_axpy)"""
                if not use_cfg:
                    return _ode_axpy(y_base, self._eval(jp, type_mask, Lmax, t, y_eval, stream), None, cfg_scale, a, stream)
                out = self._eval(jp, type_mask, Lmax, t, torch.cat([y_eval, y_eval]), stream)
                return _ode_axpy(y_base, out[:B], out[B:], cfg_scale, a, stream)

            ts = torch.linspace(0, 1, modality_steps)                  # fixed grid = the linspace itself (odeint midpoint)
            for k in range(modality_steps - 1):
                t0, dt = float(ts[k]), float(ts[k + 1] - ts[k])
                y_mid = step(t0, y, y, dt * 0.5)
                y = step(t0 + dt * 0.5, y_mid, y, dt)
            if tm is not None: tm['ode'] += mark() - t_c
            for i in group:                                            # commit, T:2531-2556
                st = states[i]; L, dl, ty = st.modality_length, md.dim_latents[st.curr_modality_id], st.curr_modality_id
                st.cache_len += L
                st.parts.append((ty, y[i, :L, :dl].reshape(*st.modality_shape, dl).clone()))
                st.curr_seq = [m.eom_ids[ty]]; st.parts.append(st.curr_seq); st.last_token = m.eom_ids[ty]
                st.tokens_seen += 1; st.num_tokens += L; st.num_past_modalities += 1
                st.phase = 'done' if st.num_tokens > max_length else 'text'
        if tm is not None:
            print('[TFX_SAMPLE_TIMING]', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})

    # ------------------------------------------------------------------ helpers
    # ------------------------------------------------------------------ pure-text generation (T:2666-2707)
    def generate_text_only(self, prompt, seq_len, temperature, min_p):
        """KV-cached greedy / min-p sampling of text only: prefill the prompt once, then one-token decode plans.
        Mirrors the reference exactly: temperature 0 takes the argmax over ALL logits, otherwise min-p filter, then the
        text-only mask (T:2690-2698)."""
        m, md, dev = self.m, self.md, self.dev
        m._require_gpu()
        prompt = prompt.to(dev)
        B, n0 = prompt.shape
        num = max(0, seq_len - n0)
        out = torch.empty(B, num, dtype=torch.long, device=dev)
        if num == 0:
            return out
        N = m.num_text_tokens

        stream = m._stream()

        def pick(logits):
            return _pick_text_only(logits, md.vocab, N, temperature, min_p, stream)

        plan, S = m._forward_plain([[row] for row in prompt], torch.ones(B, 1, device=dev), add_meta=False)
        n_pad = S['n']
        maxlen = (max(n_pad, n0 + num) + 64) // 64 * 64
        m._decode_plans = {}
        cache = self._alloc_cache(B, maxlen)
        self._fill_cache(cache, plan, B, n_pad)
        tok = pick(plan.logits.view(B, n_pad, md.vp)[:, n0 - 1].contiguous())
        out[:, 0] = tok
        p = self._decode_plan(('text', cache.data_ptr()), B, 1, cache, False)
        rows = np.arange(B, dtype=np.int32) * maxlen
        for step in range(1, num):
            L = n0 + step - 1                                          # keys already in the cache
            self._load(p, ids=tok.to(torch.int32).cpu().numpy(), pos=rows + L, kve=np.full(B, L + 1, np.int32), rot=np.full(B, L, np.int32),
                       tok_inst=np.full(B, -1, np.int32))
            self._run(p, stream, 0, p.fwd_logits_end)
            tok = pick(p.logits)
            out[:, step] = tok
        return out

    def _uncond_append(self, states, group, ucache, stream):
        """append what the samples of `group` added to their histories since their last modality phase - parts[uncond_parts:] : modality blocks
        (prompt-style: conditioned at t = 1, bidirectional inside the block, one rotary position) and text (every integer token as the null id,
        causal) - to the null-text KV cache with one decode-style forward; rows / positions as packing.token_maps would lay the full history out"""
        m, md, B = self.m, self.md, len(states)
        segs = {i: states[i].parts[states[i].uncond_parts:] for i in group}
        seg_len = lambda parts: sum(math.prod(p[1].shape[:-1]) if isinstance(p, tuple) else len(p) for p in parts)
        Lq = max(8, -(-max(seg_len(v) for v in segs.values()) // 8) * 8)           # bucketed: a handful of plans
        assert all(sum(isinstance(p, tuple) for p in v) <= 1 for v in segs.values()), 'one decoded modality per sample between two phases'
        maxlen = ucache.shape[2]
        p = self._decode_plan(('uinc', Lq, ucache.data_ptr()), B, Lq, ucache, True)
        T = B * Lq
        ids = np.zeros(T, np.int32); pos = np.full(T, -1, np.int32); kve = np.ones(T, np.int32); rot = np.zeros(T, np.int32)
        tok_inst = np.full(T, -1, np.int32)
        row_tok = {t: np.full(T, -1, np.int32) for t in range(m.num_modalities)}
        for t in p.ext_add:
            p.lat[t]['add'].zero_()
        for i in range(B):
            st = states[i]
            kve[i * Lq:(i + 1) * Lq] = max(st.uncond_len, 1)
            if i not in segs:
                continue
            j, rp = 0, st.uncond_pos
            for part in segs[i]:
                if isinstance(part, tuple):
                    ty, x = part
                    L = math.prod(x.shape[:-1])
                    sl = slice(i * Lq + j, i * Lq + j + L)
                    pos[sl] = i * maxlen + st.uncond_len + j + np.arange(L)
                    kve[sl] = st.uncond_len + j + L                       # the whole block sees itself
                    rot[sl] = rp; rp += 1
                    tok_inst[sl] = i
                    row_tok[ty][sl] = np.arange(i * Lq + j, i * Lq + j + L)
                    p.lat[ty]['x'][i * Lq + j:i * Lq + j + L].copy_(x.reshape(L, -1))
                    if ty in p.ext_add:                                   # history modalities carry their positional embedding (T:3173-3176)
                        p.lat[ty]['add'][i * Lq + j:i * Lq + j + L].copy_(m._pos_rows(ty, [tuple(x.shape[:-1])]))
                    j += L
                else:
                    n = len(part)
                    sl = slice(i * Lq + j, i * Lq + j + n)
                    ids[sl] = m.null_text_id
                    pos[sl] = i * maxlen + st.uncond_len + j + np.arange(n)
                    kve[sl] = st.uncond_len + j + 1 + np.arange(n)
                    rot[sl] = rp + np.arange(n); rp += n
                    j += n
            st.uncond_len += j; st.uncond_parts = len(st.parts); st.uncond_pos = rp
        self._load(p, ids, pos, kve, rot, tok_inst)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        for t in row_tok:
            p.row_tok[t].copy_(up(row_tok[t])); p.row_src[t].copy_(up(np.maximum(row_tok[t], 0)))
            p.row_inst[t].copy_(up(np.repeat(np.arange(B, dtype=np.int32), Lq)))
            p.set_noise(t, None)
        p.inst_time.fill_(1.)
        self._run(p, stream, 0, p.fwd_embed_end)

    def _ensure_capacity(self, cache, states, extra):
        need = max(s.cache_len for s in states) + extra + 1
        return cache if need <= cache.shape[2] else self._grow(cache, need + 192)

    def _grow(self, cache, need):
        new = self._alloc_cache(cache.shape[1], (need + 63) // 64 * 64)
        new[:, :, :cache.shape[2]].copy_(cache)
        self.m._decode_plans = {}
        self._grew = True
        return new

    def _load(self, p, ids, pos, kve, rot, tok_inst, units=None):
        """the step's five index arrays go up in ONE pinned, asynchronous copy (rows 0..4 of Plan.idx): no host sync, no pageable staging"""
        T = p.T
        # pinned staging buffers are kept in a small ring (a pinned allocation per step costs more than the step's copy); a text step ends in
        # a host sync, a modality phase issues at most four uploads before one
        ring = self.__dict__.setdefault('_stage', {})
        key = p.idx.shape[1]
        rows = 5 if units is None else 7                                 # compacted plans: the per-sample arrays ride in row 6 (row 5, q_start, stays zero)
        key = (rows, key)
        if key not in ring:
            ring[key] = [[torch.zeros(rows, key[1], dtype=torch.int32, pin_memory=True) for _ in range(8)], 0]
        bufs, k = ring[key]
        ring[key][1] = (k + 1) % len(bufs)
        host = bufs[k]
        hv = host.numpy()
        hv[0, :T], hv[1, :T], hv[2, :T], hv[3, :T], hv[4, :T] = ids, pos, kve, rot, tok_inst
        if units is not None:
            hv[6, :units.size] = units
        p.idx[:rows].copy_(host, non_blocking=True)
        p.set_rope_tables(*self.m._rope_tables(int(rot.max()) + 1))

    def _load_modality(self, p, states, group, Lmax, maxlen, halves=1):
        """index arrays of a modality decode step.  halves = 2: rows [0, B) run against the real history, rows [B, 2B) against the null-text
        history (second half of the joint cache, its own prefix lengths) - the two evaluations of classifier-free guidance in one forward"""
        B, md = len(states), self.md
        T = halves * B * Lmax
        ids = np.zeros(T, np.int32); pos = np.full(T, -1, np.int32); kve = np.ones(T, np.int32); rot = np.zeros(T, np.int32)
        tok_inst = np.full(T, -1, np.int32)
        row_tok = {t: np.full(T, -1, np.int32) for t in range(self.m.num_modalities)}
        for h in range(halves):
            for i in range(B):
                st = states[i]
                r = h * B + i                                              # batch row of the plan = cache row
                base = st.uncond_len if h == 1 else st.cache_len
                kve[r * Lmax:(r + 1) * Lmax] = max(base, 1)
                if i not in group:
                    continue
                L, ty = st.modality_length, st.curr_modality_id
                sl = slice(r * Lmax, r * Lmax + L)
                kve[r * Lmax:(r + 1) * Lmax] = base + L                    # own prefix + own (bidirectional) modality block, T:2415-2419
                pos[sl] = r * maxlen + base + np.arange(L)
                rot[r * Lmax:(r + 1) * Lmax] = st.tokens_seen              # every token of the instance shares one position, T:2411
                tok_inst[sl] = r
                row_tok[ty][sl] = np.arange(r * Lmax, r * Lmax + L)
        self._load(p, ids, pos, kve, rot, tok_inst)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        for t in row_tok:
            p.row_tok[t].copy_(up(row_tok[t])); p.row_src[t].copy_(up(np.maximum(row_tok[t], 0)))
            p.row_inst[t].copy_(up(np.repeat(np.arange(halves * B, dtype=np.int32), Lmax)))
            p.set_noise(t, None)
        for t in p.ext_add:                      # axial positional embedding of the blocks being decoded (constant over the ODE steps); zeros = none
            add = p.lat[t]['add']
            add.zero_()
            if getattr(self, 'pos_emb_in_decode', False):
                for h in range(halves):
                    for i in group:
                        st = states[i]
                        if st.curr_modality_id == t:
                            r = h * B + i
                            add[r * Lmax:r * Lmax + st.modality_length].copy_(self.m._pos_rows(t, [st.modality_shape]))

    def _run(self, p, stream, lo, hi):
        """replay a range of a decode plan: eagerly the first time (one-off kernel attribute setup happens outside any capture), as a
        hipGraph afterwards - every kernel argument of a decode plan is a fixed pointer, the per-step values live in device arrays"""
        seen = p.__dict__.setdefault('_ran', set())
        if (lo, hi) not in seen:
            seen.add((lo, hi))
            Plan.run(p.fwd, stream, lo, hi)
        else:
            Plan.run(p.fwd, stream, lo, hi, graph=True)

    def _type_masks(self, states, group, Lmax):
        """per modality type: (B, Lmax, 1) 0/1 mask of the state rows that belong to a sample decoding that type"""
        B = len(states)
        masks = {}
        for ty in range(self.m.num_modalities):
            mk = np.zeros((B, Lmax, 1), np.float32)
            for i in group:
                if states[i].curr_modality_id == ty:
                    mk[i, :states[i].modality_length] = 1.
            masks[ty] = torch.from_numpy(mk).to(self.dev)
        return masks

    def _eval(self, p, type_mask, Lmax, t, y, stream):
        """one model evaluation of the joint ODE state y (B, Lmax, dmax) at time t -> predicted flow, same layout (T:2468-2521)."""
        md, B = self.md, y.shape[0]
        p.inst_time.fill_(t)
        for ty in range(self.m.num_modalities):
            p.lat[ty]['x'].copy_(y[:, :, :md.dim_latents[ty]].reshape(B * Lmax, md.dim_latents[ty]))
        self._run(p, stream, 0, p.fwd_embed_end)
        self._run(p, stream, p.fwd_logits_end, p.fwd_pred_end)
        if self.m.num_modalities == 1 and md.dim_latents[0] == y.shape[2]:
            return p.lat[0]['pred'].view(B, Lmax, -1) * type_mask[0]
        out = torch.zeros_like(y)
        for ty in range(self.m.num_modalities):
            dl = md.dim_latents[ty]
            out[:, :, :dl] += p.lat[ty]['pred'].view(B, Lmax, dl) * type_mask[ty]
        return out
