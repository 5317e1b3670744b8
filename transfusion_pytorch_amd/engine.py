"""Native training engine: a static launch list (forward + hand-written backward) over preallocated HBM
buffers, executed through the C ABI (libtfx_hip.so) on PyTorch's current HIP stream.

A `Plan` is specialised to one batch geometry (b, n, #instances, latent rows per type): every buffer address
is fixed, so a step is `for fn, args in plan.fwd: fn(args, stream)` - no allocation, no host sync, and the whole
list can be captured in a hipGraph.  Math follows SURVEY.md Appendix A (reference T:1100-1266, T:3280-3376).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import capi
from .params import ModelDims, ParamStore, pad_to

BF16 = torch.bfloat16
E = capi.ENUMS


def _p(t, *idx):
    """device pointer of t[idx...]"""
    for i in idx:
        t = t[i]
    return t.data_ptr()


def skip_sources(md: ModelDims):
    """simulate the U-Net skip stack (T:1206-1219): for each late layer index with a skip_proj, the index j
    such that the popped skip is xres[j] (the INPUT of early layer j)."""
    stack, src = [], {}
    for i in range(md.depth):
        layer = i + 1
        if layer <= md.depth // 2:
            stack.append(i)
        elif md.has_skip(i):
            src[i] = stack.pop()
    assert not stack
    return src




_LAUNCH = capi.STRUCTS['tfx_launch']
_RAW = capi.STRUCTS['tfx_raw_args']


def raw_args(fn_name: str, values):
    """pack the positional arguments of `fn_name` (without the stream) into a tfx_raw_args: pointers fill p0.., integers i0..,
    floats f0, in declaration order (include/tfx.h "launch lists")."""
    argt = capi.FUNCTIONS[fn_name][1][:-1]
    assert len(argt) == len(values), f'{fn_name}: {len(values)} arguments for {len(argt)} parameters'
    r = _RAW()
    np_, ni, nf = 0, 0, 0
    for ty, v in zip(argt, values):
        if ty is ctypes.c_void_p:
            setattr(r, f'p{np_}', v); np_ += 1
        elif ty is ctypes.c_float:
            assert nf == 0
            r.f0 = v; nf += 1
        else:
            setattr(r, f'i{ni}', v); ni += 1
    return r


class Side(tuple):
    """a launch that runs on the library's side stream (include/tfx.h `tfx_launch.stream` = 1): the weight-gradient GEMMs, next to
    the data-gradient chain on the caller's stream.  Unpacks like any other (entry point, args) item."""


_AUTO_AFTER = 2             # a training range is captured when its fingerprint has repeated this many times in a row
TRAIN_GRAPH = os.environ.get('TFX_TRAIN_GRAPH', '0') == '1'          # replay the training step's launch lists as hipGraphs (LaunchList.replay_auto)
_SYNC_OPS = {'tfx_fork': 'TFX_OP_FORK', 'tfx_join': 'TFX_OP_JOIN', 'tfx_join_record': 'TFX_OP_JOIN_RECORD', 'tfx_join_wait': 'TFX_OP_JOIN_WAIT'}
_DUMMY = ctypes.c_int32(0)


class LaunchList(list):
    """a launch list plus its native image (array of tfx_launch) for `tfx_run_list`.  Items are (entry point name, args struct) or
    (positional entry point, argument tuple); args structs are referenced by address, so in-place updates of their fields
    (grad scales, segment counts, noise pointers) are seen by the next replay.  A MUTABLE argument list (e.g. the
    model_output_clean launch) is re-packed on every replay."""
    _image = None
    _graphs = None
    meta = None              # {list index: (family, algorithmic work)} - bookkeeping for bench.py's per-family roofline (HBM-bound launches: bytes)

    def graph(self, lo: int, hi: int):
        """hipGraph of launches[lo:hi] (created on first use).  Only for lists whose kernel arguments are fixed pointers / sizes between
        replays (decode plans): the arguments are frozen at capture time - `invalidate_graphs` after changing an args struct."""
        if self._graphs is None:
            self._graphs = {}
        g = self._graphs.get((lo, hi))
        if g is None:
            arr = self.native()
            out = ctypes.c_void_p()
            rc = capi.lib().tfx_graph_create(ctypes.byref(arr, lo * ctypes.sizeof(_LAUNCH)), hi - lo, ctypes.byref(out))
            if rc != 0 or not out.value:
                raise capi.TfxError(f'tfx_graph_create failed with code {rc}')
            g = self._graphs[(lo, hi)] = out.value
        return g

    def invalidate_graphs(self):
        for g in (self._graphs or {}).values():
            capi.lib().tfx_graph_destroy(ctypes.c_void_p(g))
        self._graphs = None
        for st in (self._auto or {}).values():
            if st[1]:
                capi.lib().tfx_graph_destroy(ctypes.c_void_p(st[1]))
        self._auto = None

    _auto = None             # {(lo, hi): [fingerprint, graph | None, consecutive replays under that fingerprint]} - training lists, see `replay_auto`

    def replay_auto(self, lo: int, hi: int, sp) -> bool:
        """TRAINING lists as hipGraphs.  Their args structs hold per-step scalars and pointers (loss seeds, row counts, noise pointers, segment
        counts) that the host rewrites in place, so a capture is only valid while `tfx_list_fingerprint` - a hash of every byte a capture
        freezes - stays what it was: the range is captured once the same fingerprint has come back `_AUTO_AFTER` times in a row (a fixed-shape
        training loop: from the third step on), replayed as ONE graph launch while it holds (a single chain of kernels, see below), and dropped the moment it changes (ragged batches keep
        running through `tfx_run_list`: a re-capture per step would cost more than the launches it saves).  Returns False when the caller has
        to run the list itself."""
        lib = capi.lib()
        arr = self.native()
        fp = ctypes.c_int64()
        rc = lib.tfx_list_fingerprint(ctypes.byref(arr, lo * ctypes.sizeof(_LAUNCH)), hi - lo, ctypes.byref(fp))
        if rc != 0:
            return False
        if self._auto is None:
            self._auto = {}
        st = self._auto.get((lo, hi))
        if st is None or st[0] != fp.value:
            if st is not None and st[1]:
                lib.tfx_graph_destroy(ctypes.c_void_p(st[1]))
            self._auto[(lo, hi)] = [fp.value, None, 0]
            return False
        st[2] += 1
        if st[1] is None:
            if st[2] < _AUTO_AFTER:
                return False
            out = ctypes.c_void_p()
            # captured as ONE chain (the side-stream launches in list order on the capture stream): with the weight-gradient GEMMs as parallel
            # branches the replay measured 34.5 ms per step against 28.5 for the chain and 28.7 for the two-stream list (config 2, same box,
            # ROCm 7.2: the graph executor does not overlap the branches, it stalls at their joins)
            # (tfx_graph_create_single: the one-stream mode belongs to this capture on this thread - the process-wide switch is not touched)
            rc = lib.tfx_graph_create_single(ctypes.byref(arr, lo * ctypes.sizeof(_LAUNCH)), hi - lo, ctypes.byref(out))
            if rc != 0 or not out.value:
                st[2] = -(1 << 30)                                   # this range does not capture (never retried): the list path keeps running it
                return False
            st[1] = out.value
        rc = lib.tfx_graph_launch(ctypes.c_void_p(st[1]), sp)
        if rc != 0:
            raise capi.TfxError(f'tfx_graph_launch failed with code {rc}')
        return True

    def __del__(self):
        try:
            self.invalidate_graphs()
        except Exception:
            pass

    def native(self):
        if self._image is None or self._image[0] != len(self):
            arr = (_LAUNCH * max(len(self), 1))()
            keep, live = [], []
            for k, item in enumerate(self):
                fn, a = item
                if isinstance(fn, str) and fn in _SYNC_OPS:          # fork / join: `a` is the event slot
                    arr[k].op = capi.ENUMS[_SYNC_OPS[fn]]
                    arr[k].stream = int(a)
                    arr[k].args = ctypes.addressof(_DUMMY)
                    continue
                arr[k].stream = 1 if isinstance(item, Side) else 0
                if isinstance(fn, str):
                    arr[k].op = capi.ENUMS['TFX_OP_' + fn[4:].upper()]
                    arr[k].args = ctypes.addressof(a)
                else:
                    name = fn.__name__
                    r = raw_args(name, a)
                    keep.append(r)
                    if isinstance(a, list):
                        live.append((name, a, r))
                    arr[k].op = capi.ENUMS['TFX_OP_' + name[4:].upper()]
                    arr[k].args = ctypes.addressof(r)
            self._image = (len(self), arr, keep, live)
        for name, a, r in self._image[3]:
            fresh = raw_args(name, a)
            ctypes.memmove(ctypes.addressof(r), ctypes.addressof(fresh), ctypes.sizeof(_RAW))
        return self._image[1]


class Plan:
    def __init__(self, ps: ParamStore, b: int, n: int, I: int, R: dict, training: bool = True, cache=None, dp_groups: int = 0, tile_attn: bool = False, units=None):
        """cache: None, or a KV cache tensor [depth, b, maxlen, 2*heads*64] (k~ | v per token).  With a cache the plan is a
        DECODE step: each layer appends this step's k~ / v rows at `cache_pos` (flat row b*maxlen + position, -1 = skip) and
        attention reads keys / values from the cache (per-token visible length in `kv_end`).
        units = (samples, max rows per sample): a COMPACTED decode plan - its `b * n` token rows are a flat list in which sample s of `samples`
        owns the rows unit_row0[s] .. + unit_cnt[s] (device arrays the step fills; 0 rows = the sample sits this step out), so a mixed step of the
        continuous schedule carries the rows its live samples need instead of (modality length + 1) rows for every sample."""
        md = ps.md
        self.units = units
        assert not units or (cache is not None and tile_attn), 'compacted plans are decode plans on the tiled attention kernel (the row ranges are its arguments)'
        self.cache = cache
        # decode plans: `tile_attn` keeps the tiled forward kernel also for one or two new rows per sample (tfx_decode_attn would pick its
        # matrix-core-free kernel there) - the continuous decode schedule runs a sample's text token in plans of different row counts, and a token
        # must come out of the same arithmetic whichever plan carried it (sample_many == per-prompt sample_one, reference tests :785-808)
        self.tile_attn = tile_attn
        assert cache is None or not training
        self.dp_groups = dp_groups          # > 0: the backward list is cut into that many layer groups for the overlapped gradient all-reduce (optim.GradReducer)
        self.bwd_cuts = []                  # [(list index, first layer of the group, last layer of the group)], in backward order
        self.ps, self.md, self.b, self.n, self.I, self.R = ps, md, b, n, I, dict(R)
        self.T = T = b * n
        dev = ps.device
        d, hd, D, dip, ldq, nt3 = md.dim, md.hdk, md.depth, md.dip, md.ldq, md.nt3
        self.nbytes = 0                     # device bytes this plan owns (the plan cache evicts by total size, Transfusion._plan)
        def z(*s, dtype=BF16):
            t = torch.zeros(*s, device=dev, dtype=dtype); self.nbytes += t.numel() * t.element_size(); return t
        def e(*s, dtype=BF16):
            t = torch.empty(*s, device=dev, dtype=dtype); self.nbytes += t.numel() * t.element_size(); return t
        I1 = max(I, 1)
        # ---- index arrays (filled per step)
        # per-token index arrays: rows of ONE int32 buffer, so that a decode step uploads its five host-built arrays
        # (ids, cache positions, visible lengths, rotary positions, instance ids) with a single pinned, asynchronous copy
        self.idx = z(7, max(T, 1, 4 * units[0] if units else 1), dtype=torch.int32)
        self.text_ids, self.cache_pos, self.kv_end, self.rot_pos, self.tok_inst, self.q_start, self.labels = (self.idx[r, :T] if r != 1 else self.idx[r] for r in range(7))
        if units:                                                           # per-sample arrays of a compacted decode plan ride in row 6 (a decode plan has no labels)
            U = units[0]
            self.unit_row0, self.unit_cnt, self.unit_blk0, self.unit_txt = (self.idx[6, k * U:(k + 1) * U] for k in range(4))
        self.inst_time = z(I1, dtype=torch.float32)
        self.row_tok = {t: z(r, dtype=torch.int32) for t, r in R.items()}
        self.rowbuf = None
        if cache is not None and R and len(set(R.values())) == 1:
            # decode plans: the per-step row maps of all types (row_tok, then row_src) and a small fp32 control area live in ONE buffer, so a step
            # uploads them with one copy
            r = next(iter(R.values()))
            self.row_stride = rs = (r + 3) // 4 * 4                       # 16-byte aligned rows
            self.rowbuf = z(2 * len(R) * rs + 2 * max(b, units[0] if units else 0), dtype=torch.int32)
            self.row_tok = {t: self.rowbuf[k * rs:k * rs + r] for k, t in enumerate(R)}
            self.ctl = self.rowbuf[2 * len(R) * rs:].view(torch.float32)
        self.row_inst = {t: z(r, dtype=torch.int32) for t, r in R.items()}
        # gather form of row_tok (dropped / padding rows, -1 there, clamped to row 0): the row-GATHERING GEMM operands index with it - what they read
        # for such rows is multiplied by zeros or ignored (`set_rows`)
        self.row_gat = {t: z(r, dtype=torch.int32) for t, r in R.items()} if (training and cache is None) else self.row_tok
        self._rows_dep = {t: [] for t in R}      # per type: (object, field or list index) holding the number of REAL latent rows of the step
        # ---- forward activations
        self.hid = e(D + 1, T, d)
        self.xres = [self.hid[0]] + [e(T, d) for _ in range(D)]
        # A training plan keeps every layer's activations for the backward.  An inference plan (prefill / decode) needs per layer only what
        # outlives the layer - the hiddens (AttentionResidual reads all of them), the AttentionResidual outputs (U-Net skips), and for a
        # prefill the rotated keys / values that fill the KV cache afterwards; everything else is ONE buffer shared by all layers (`li` = 0).
        # At dim 1024 / depth 24 a 64 x 320-token prefill plan is 5 GB this way instead of 25 GB: eight cached plans used to exhaust the 288 GB.
        nl = D if training else 1
        nkv = D if (training or cache is None) else 1
        self._li = (lambda i: i) if training else (lambda i: 0)
        self._lkv = (lambda i: i) if nkv == D else (lambda i: 0)
        xa_shared = None if training else e(T, d)
        self.xa = {i: (e(T, d) if training else xa_shared) for i in range(D) if md.has_skip(i)}
        self.stats = e(4, nl, T, dtype=torch.float32)         # mean_a, rstd_a, mean_f, rstd_f
        self.ua = e(nl, T, d); self.uf = e(nl, T, d)
        self.qkvg = z(nkv, T, ldq); self.qkr = e(nkv, T, 2 * hd); self.og = z(nl, T, hd) if units else e(nl, T, hd)       # (compacted plans: padding rows are never written by the attention - keep them finite)
        self.lse = e(nl, units[0], md.heads, units[1], dtype=torch.float32) if units else e(nl, b, md.heads, n, dtype=torch.float32)
        self.ya = e(nl, T, d); self.xb = e(nl, T, d); self.yf = e(nl, T, d)
        self.ag = e(nl, T, 2 * dip); self.hm = e(nl, T, dip)
        self.embed = e(T, d)
        self.logits = e(T, md.vp, dtype=torch.float32); self.dlogits = e(T, md.vp)
        self.fe = z(I1, md.kf); self.cond = e(I1, 4 * d); self.pre = e(I1, 4 * d)
        self.tables = z(I1, nt3, dtype=torch.float32)
        self.lat = {}
        self.ext, self.ext_add = set(md.ext_types) & set(R), set(md.pos_types) & set(R)
        for t, r in R.items():
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64)
            if t in self.ext:
                # `pre_post_transformer_enc_dec` type (T:1451-1494): the token rows are produced by the user's encoder in PyTorch (`tok`), the
                # final embedding rows go back out to the user's decoder; `gemb` receives d loss / d (those embedding rows) before the backward
                self.lat[t] = dict(tok=z(r, d), gemb=z(r, d))
                continue
            self.lat[t] = dict(x=e(r, dl, dtype=torch.float32), eps=z(r, dl, dtype=torch.float32), xt=z(r, dlp),
                               flow=e(r, dl, dtype=torch.float32), pred=e(r, dl, dtype=torch.float32), dpred=z(r, dlp))
        for t in self.ext_add:
            self.lat[t]['add'] = z(R[t], d)      # additive token rows: the axial positional embedding (T:1384-1403, T:3173-3176), bf16
        self.acc = z(max(8, 2 + 3 * len(md.dim_latents)), dtype=torch.float32)      # [ce sum, ce count, flow sse per type..., velocity sse per type..., weighted recon sse per type...]
        self.vel = LaunchList()                         # optional launches: velocity-consistency MSE against an EMA teacher's flows (T:3394-3418)
        self.rec = LaunchList()                         # optional launches: reconstruction loss on the same predictions (MP:177-200, T:2840-2853)
        self.cos_tab = self.sin_tab = None
        # per layer: the soft-cap plan tfx_qk_norm_rope_fwd derives from the layer's QK-RMSNorm gains (tfx.h) and its attention kernels - forward
        # and, later, backward - read: polynomial degree and coefficients from the BOUND on the scores, no look at the data.  TFX_SC_PLAN=0: decide from the scores
        self.sc_plan = z(D, 8, dtype=torch.float32)
        self.fwd, self.bwd = LaunchList(), LaunchList()
        self.noise_args = {}
        self.loaded_structure = None
        self._seg_args = []
        self.seg_start = z(max(T, 1), dtype=torch.int32); self.seg_len = z(max(T, 1), dtype=torch.int32)
        # AttentionResidual backward in pull form (tfx_attnres_pull_bwd): every layer's depth softmax is kept per token by the forward;
        # TFX_ATTNRES_PULL=0 keeps the push form (tfx_attnres_bwd: one read-modify-write sweep over all earlier hiddens per layer; A/B)
        self.pull = training and D <= 32 and md.dim <= 1024
        if self.pull:
            self.arsave = [e(T, i + 2, 4, dtype=torch.float32) for i in range(D)]
            self.arerr = e(D, T, d)                          # what rounding each AttentionResidual output to bf16 dropped
        self._build_forward()
        if training:
            self.dH = e(D + 1, T, d)
            self.gx = {i: e(T, d) for i in range(D) if md.has_skip(i)}; self.du = e(T, d)
            # weight-gradient GEMMs run on a side stream one layer behind the data-gradient chain (TFX_SIDE_STREAM=0: one stream):
            # the buffers they read are kept per wrapper and double-buffered by layer parity
            # measured (A/B on one box): dim 512 -2 % step time; through round 5's first session dim 768 +2 % and dim 1024 +4 % (two 8-wave GEMM kernels sharing the
            # chip cost more than the token-wise overlap gained), so the side stream stopped at dim 512.  With the one-wave weight-gradient kernels and their
            # grouped launches it pays at every BASELINE width: dim 768 84.2 -> 83.0 ms, dim 1024 184.2 -> 181.8 ms per step (gpurun_out/ow41.txt, two rounds)
            # (TFX_SIDE_STREAM=1 forces it on, =0 off)
            env = os.environ.get('TFX_SIDE_STREAM')
            self.side = D <= 30 and (env == '1' or (env is None and md.dim <= 1024))
            nb = 2 if self.side else 1
            self.dy_f = [e(T, d) for _ in range(nb)]; self.dy_a = [e(T, d) for _ in range(nb)] if self.side else self.dy_f
            self.dskip = {j: e(T, d) for j in set(skip_sources(md).values())}
            self.dqkvg_p = [z(T, ldq) for _ in range(nb)]; self.dog = e(T, hd); self.do_eff = e(T, hd)
            # d q~ | d k~: written only when the QK-norm / RoPE backward is its own launch (TFX_ATTN_QKNR=0); the fused epilogues never touch dq / dk
            self.dqk = e(T, 2 * hd) if os.environ.get('TFX_ATTN_QKNR', '1') == '0' else None
            self.delta = e(b, md.heads, n, dtype=torch.float32)
            self.dag_p = [e(T, 2 * dip) for _ in range(nb)]; self.dembed = e(T, d); self.gfin = e(T, d); self.dx0 = e(T, d)
            self.dtables = z(I1, nt3, dtype=torch.float32); self.dtab_bf = e(I1, nt3)
            self.dcond = e(I1, 4 * d); self.dpre = e(I1, 4 * d)
            self.onehot = e(T, md.vp)
            self._build_backward()

    # ------------------------------------------------------------------------------------ helpers
    def _nt(self, lst, algo_n=None, algo_k=None, **kw):
        a = capi.make_args('tfx_gemm_nt_args', **kw)
        a._algo_flops = 2.0 * kw['M'] * (algo_n or kw['N']) * (algo_k or kw['K'])      # algorithmic (unpadded) work of this launch
        lst.append(('tfx_gemm_nt', a))

    def _tn(self, lst, M, N, K, **kw):
        splits = 0                # the library picks the split count from the tile count of the kernel it launches (tfx.h, gemm.hip tn_auto_splits)
        kw.setdefault('k_valid', K)
        algo_n = kw.pop('algo_n', None)
        side = kw.pop('side', False)
        a = capi.make_args('tfx_gemm_tn_args', M=M, N=N, K=K, splits=splits, accumulate=1, alpha=1.0, **kw)
        a._algo_flops = 2.0 * M * (algo_n or N) * kw['k_valid']
        lst.append(Side(('tfx_gemm_tn', a)) if side else ('tfx_gemm_tn', a))

    def _tn_group(self, lst, M, problems, side=False):
        """Weight-gradient products over the same M rows as ONE launch (tfx.h group_next): `problems` = [(N, K, kwargs of _tn), ...].  The library runs the chain
        on one grid of its one-wave kernel - half the fp32 atomics of the split-M sums, 256 x 256 tiles for the 512 x 512 products - or, where a product does not
        qualify, one by one.  TFX_TN_GROUP=0 keeps them separate launches (what the list held through round 5's first session)."""
        if len(problems) < 2 or os.environ.get('TFX_TN_GROUP', '1') == '0':
            for (N, K, kw) in problems:
                self._tn(lst, M, N, K, side=side, **kw)
            return
        structs, flops = [], 0.0
        for (N, K, kw) in problems:
            kw = dict(kw); kw.setdefault('k_valid', K)
            algo_n = kw.pop('algo_n', None)
            a = capi.make_args('tfx_gemm_tn_args', M=M, N=N, K=K, splits=0, accumulate=1, alpha=1.0, **kw)
            flops += 2.0 * M * (algo_n or N) * kw['k_valid']
            structs.append(a)
        for a, b in zip(structs, structs[1:]):
            a.group_next = ctypes.addressof(b)
        head = structs[0]
        head._chain = structs[1:]                                # (keeps the chained structs alive with the list)
        head._algo_flops = flops
        lst.append(Side(('tfx_gemm_tn', head)) if side else ('tfx_gemm_tn', head))

    def _k(self, lst, fn, struct, **kw):
        lst.append((fn, capi.make_args(struct, **kw)))

    def _raw(self, lst, fn, *args):
        lst.append((fn, args))

    def _hbm(self, lst, passes, extra_bytes=0):
        """tag the launch just appended as HBM-bound with its ALGORITHMIC traffic: `passes` sweeps of a [T, d] bf16 matrix (+ extra bytes).
        Read by bench.py (`roofline_by_family`); no effect on the replay."""
        if lst.meta is None:
            lst.meta = {}
        lst.meta[len(lst) - 1] = ('hbm', passes * self.T * self.md.dim * 2 + extra_bytes)

    def _tab(self, i, w):
        """pointer to (layer i, wrapper w) slice of the AdaLN tables / their grads: gamma | beta | z."""
        off = ((i * 2 + w) * 3 * self.md.dim) * 4
        return self.tables.data_ptr() + off, (self.dtables.data_ptr() + off if hasattr(self, 'dtables') else 0)

    # ------------------------------------------------------------------------------------ forward
    def _build_forward(self):
        ps, md, T, I = self.ps, self.md, self.T, self.I
        d, hd, D, dip, ldq, nt3, H = md.dim, md.hdk, md.depth, md.dip, md.ldq, md.nt3, md.heads
        S = ps.shadows
        L = self.fwd
        pp = ps.ptr
        if self.cache is not None:
            # decode plans: `row_src` = row_tok with the dropped rows (-1) clamped to 0, so a gather never reads out of bounds (those rows are ignored by the caller)
            self.row_src = {t: torch.zeros(r, device=self.ps.device, dtype=torch.int32) for t, r in self.R.items()}
            if self.rowbuf is not None:
                r, M, rs = next(iter(self.R.values())), len(self.R), self.row_stride
                self.row_src = {t: self.rowbuf[(M + k) * rs:(M + k) * rs + r] for k, t in enumerate(self.R)}
        self._clean_launch, self.clean_mode = {}, ('model' if self.cache is not None else 'latent')
        for t, r in self.R.items():
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            if t in self.ext:     # rows from the user's encoder: scatter them into the stream
                self._raw(L, capi.lib().tfx_scatter_rows_bf16, lt['tok'].data_ptr(), d, d, self.hid[0].data_ptr(), d, self.row_tok[t].data_ptr(), r)
            else:
                self._k(L, 'tfx_noise_mix', 'tfx_noise_mix_args', R=r, dl=dl, x=lt['x'], eps=lt['eps'], row_inst=self.row_inst[t],
                        inst_time=self.inst_time, xt=lt['xt'], ld_xt=dlp, flow=lt['flow'])
                self.noise_args[t] = L[-1][1]
                if dl == d:       # nn.Identity latent_to_model (T:1478): the noised rows ARE the tokens - scatter them into the stream
                    self._raw(L, capi.lib().tfx_scatter_rows_bf16, lt['xt'].data_ptr(), dlp, d, self.hid[0].data_ptr(), d, self.row_tok[t].data_ptr(), r)
                    lt['proj'] = lt['xt']
                elif md.model_output_clean:
                    # `model_output_clean`: the projected noised tokens (`processed.packed`, MP:786-792) are kept as rows of their own - the model-space
                    # conversion subtracts W proj, and its backward needs proj again, WITHOUT what joins the stream afterwards (the positional embedding,
                    # T:3173-3176).  latent_to_model writes them to `proj`, one more launch scatters them into the stream.
                    lt['proj'] = torch.zeros(r, d, device=self.ps.device, dtype=BF16); self.nbytes += r * d * 2
                    self._nt(L, algo_k=dl, A=lt['xt'], lda=dlp, B=S[f'in{t}'], ldb=dlp, M=r, N=d, K=dlp, epi=E['TFX_EPI_BF16'], C=lt['proj'], ldc=d,
                             bias=pp(f'latent_to_model_projs.{t}.bias'))
                    self._raw(L, capi.lib().tfx_scatter_rows_bf16, lt['proj'].data_ptr(), d, d, self.hid[0].data_ptr(), d, self.row_tok[t].data_ptr(), r)
                else:
                    self._nt(L, algo_k=dl, A=lt['xt'], lda=dlp, B=S[f'in{t}'], ldb=dlp, M=r, N=d, K=dlp, epi=E['TFX_EPI_BF16'], C=self.hid[0], ldc=d,
                             bias=pp(f'latent_to_model_projs.{t}.bias'), rowmap=self.row_tok[t])
            if md.model_output_clean and t not in self.ext:
                self._clean_q(L, t, r)                    # q = model_to_latent(proj)
            if t in self.ext_add:     # tokens += positional embedding rows (T:3173-3176): identity GEMM, mapped RESID epilogue
                self._nt(L, A=lt['add'], lda=d, B=S['eye'], ldb=d, M=r, N=d, K=d, epi=E['TFX_EPI_RESID'], C=self.hid[0], ldc=d,
                         R=self.hid[0], ldr=d, resid_mapped=1, rowmap=self.row_tok[t])
        self._k(L, 'tfx_embed_fwd', 'tfx_embed_args', T=T, d=d, text_ids=self.text_ids, tok_inst=self.tok_inst, table=S['embed'], x=self.hid[0])
        self._hbm(L, 1)
        self.fwd_cond = (len(L), len(L))      # [begin, end) of the time-conditioning launches: inst_time -> per-instance AdaLN tables
        if I > 0:
            self._k(L, 'tfx_fourier', 'tfx_fourier_args', I=I, half=d // 2, times=self.inst_time, w=ps.fourier_w, out=self.fe, ld=md.kf)
            self._nt(L, algo_k=d + 1, A=self.fe, lda=md.kf, B=S['time'], ldb=md.kf, M=I, N=4 * d, K=md.kf, epi=E['TFX_EPI_SILU'], C=self.cond, ldc=4 * d,
                     C2=self.pre, ldc2=4 * d, bias=pp('transformer.to_time_cond.1.bias'))
            self._nt(L, A=self.cond, lda=4 * d, B=S['ada'], ldb=4 * d, M=I, N=nt3, K=4 * d, epi=E['TFX_EPI_F32'], C=self.tables, ldc=nt3,
                     bias=pp('transformer.layers.0.1.to_film.bias'))
            self.fwd_cond = (self.fwd_cond[0], len(L))
        src = skip_sources(md)
        fused_pre = {}                        # layer -> its attention-side AdaLN-pre args when the previous layer's end launch runs them
        # token-wise launches that follow each other on the same rows run as ONE launch (bit-identical to the separate kernels, tests/test_kernels_gpu.py):
        # wrapper output + next wrapper input, and the end of a layer (feed-forward output side, AttentionResidual, the next layer's input side).
        # One HBM round trip of the row instead of two / three; decode plans are launch-bound
        fuse = True
        for i in range(D):
            p = f'transformer.layers.{i}'
            li, lkv = self._li(i), self._lkv(i)
            x_in = self.xres[i]
            if md.has_skip(i):
                self._nt(L, A=x_in, lda=d, A2=self.xres[src[i]], lda2=d, K1=d, B=S[f'skip{i}'], ldb=2 * d, M=T, N=d, K=2 * d,
                         epi=E['TFX_EPI_RESID'], C=self.xa[i], ldc=d, R=x_in, ldr=d)
                x_a = self.xa[i]
            else:
                x_a = x_in
            ta, _ = self._tab(i, 0); tf, _ = self._tab(i, 1)
            a_pre_attn = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=x_a, u=self.ua[li], tok_inst=self.tok_inst, table=ta, ld_table=nt3,
                                        gamma_text=pp(f'{p}.1.layernorm_gamma'), mean=_p(self.stats, 0, li), rstd=_p(self.stats, 1, li))
            if i in fused_pre:                # already done by the previous layer's end launch (tfx_layer_end_fwd)
                assert fused_pre[i].x == a_pre_attn.x and fused_pre[i].u == a_pre_attn.u
            else:
                L.append(('tfx_adaln_pre_fwd', a_pre_attn))
                self._hbm(L, 2, 8 * T)                                  # x in, u out, mean / rstd out
            gam = (lambda nm: S[f'g{nm}{i}']) if md.dim_head != 64 else (lambda nm: pp(f'{p}.1.fn.{nm}_norm.gamma'))
            plan_kw = dict(sc_plan=self.sc_plan[i], softcap=50.0) if self.sc_plan is not None else {}
            # decode plans (round 4): the same epilogue in the decode-step GEMM kernel, with the KV-cache append (`qk_cache`): one launch per layer
            # less in a step that is nothing but ~225 dependent launches.  Training plans keep the token-wise launch: fused there the projection grows from 150 to
            # 195 us for the 53 us launch it removes (27.50 vs 27.50 ms per step, round 4)
            fuse_qk = self.cache is not None
            if fuse_qk:
                # SURVEY K4 (T:946-965): QK-RMSNorm + RoPE ride in the epilogue of the [q | k | v | gates] projection (TFX_EPI_QKV_NORM_ROPE: the raw
                # projection AND q~ | k~ leave the GEMM; shapes off the decode-step kernel run as two launches inside the call), with the KV-cache append
                self._nt(L, algo_n=md.nq, A=self.ua[li], lda=d, B=S[f'qkvg{i}'], ldb=d, M=T, N=md.nqk, K=d, epi=E['TFX_EPI_QKV_NORM_ROPE'], C=self.qkvg[lkv], ldc=ldq,
                         C2=self.qkr[lkv], ldc2=2 * hd, qk_heads=H, qk_gamma_q=gam('q'), qk_gamma_k=gam('k'), qk_rot_pos=self.rot_pos, qk_cos=0, qk_sin=0,
                         qk_q_scale=md.dim_head ** -0.5, qk_norm_scale=md.dim_head ** 0.5,
                         **({'qk_plan': plan_kw['sc_plan'], 'qk_softcap': 50.0} if plan_kw else {}),
                         **(dict(qk_cache=self.cache[i], qk_ld_cache=2 * hd, qk_cache_pos=self.cache_pos) if self.cache is not None else {}))
                self._rope_nt_args = getattr(self, '_rope_nt_args', []) + [L[-1][1]]
                self._k(L, 'tfx_attn_fwd' if (self.cache is None or self.tile_attn) else 'tfx_decode_attn', 'tfx_attn_args', **self._attn_kw(i))
            else:
                self._nt(L, algo_n=md.nq, A=self.ua[li], lda=d, B=S[f'qkvg{i}'], ldb=d, M=T, N=md.nqk, K=d, epi=E['TFX_EPI_BF16'], C=self.qkvg[lkv], ldc=ldq)
                self._qknr_separate(L, i, li, lkv, gam, plan_kw)
            self._nt(L, algo_k=md.hd, A=self.og[li], lda=hd, B=S[f'out{i}'], ldb=hd, M=T, N=d, K=hd, epi=E['TFX_EPI_BF16'], C=self.ya[li], ldc=d)
            a_post = capi.make_args('tfx_adaln_post_args', T=T, d=d, x=x_a, y=self.ya[li], out=self.xb[li], tok_inst=self.tok_inst,
                                    table=ta, ld_table=nt3, layerscale=pp(f'{p}.1.layerscale'))
            a_pre = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=self.xb[li], u=self.uf[li], tok_inst=self.tok_inst, table=tf, ld_table=nt3,
                                   gamma_text=pp(f'{p}.2.layernorm_gamma'), mean=_p(self.stats, 2, li), rstd=_p(self.stats, 3, li))
            if fuse:                          # both sides in one launch
                self._keep = getattr(self, '_keep', []) + [a_post, a_pre]
                self._raw(L, capi.lib().tfx_adaln_post_pre_fwd, ctypes.addressof(a_post), ctypes.addressof(a_pre))
                self._hbm(L, 4, 8 * T)                                  # x, y in; out, u out
            else:
                L.append(('tfx_adaln_post_fwd', a_post))
                L.append(('tfx_adaln_pre_fwd', a_pre))
            self._nt(L, algo_n=2 * md.di, A=self.uf[li], lda=d, B=S[f'ff1{i}'], ldb=d, M=T, N=2 * dip, K=d, epi=E['TFX_EPI_GEGLU'], C=self.ag[li], ldc=2 * dip,
                     C2=self.hm[li], ldc2=dip, bias=S[f'ff1b{i}'])
            self._nt(L, algo_k=md.di, A=self.hm[li], lda=dip, B=S[f'ff2{i}'], ldb=dip, M=T, N=d, K=dip, epi=E['TFX_EPI_BF16'], C=self.yf[li], ldc=d,
                     bias=pp(f'{p}.2.fn.net.3.bias'))
            a_postf = capi.make_args('tfx_adaln_post_args', T=T, d=d, x=self.xb[li], y=self.yf[li], out=self.hid[i + 1], tok_inst=self.tok_inst,
                                     table=tf, ld_table=nt3, layerscale=pp(f'{p}.2.layerscale'))
            a_ar = capi.make_args('tfx_attnres_args', T=T, d=d, L=i + 2, hiddens=self.hid, stride_h=T * d,
                                  gamma=pp(f'{p}.3.norm_keys.gamma'), pq=pp(f'{p}.3.pseudo_queries'), out=self.xres[i + 1],
                                  save=self.arsave[i] if self.pull else None, err=self.arerr[i] if self.pull else None)
            if fuse:
                # the end of the layer is ONE launch - feed-forward output side, AttentionResidual, and (when the next layer reads
                # the result directly, i.e. has no U-Net skip projection in front) the next layer's attention-side AdaLN-pre
                nxt = None
                if i + 1 < D and not md.has_skip(i + 1):
                    pn = f'transformer.layers.{i + 1}'
                    nxt = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=self.xres[i + 1], u=self.ua[self._li(i + 1)], tok_inst=self.tok_inst, table=self._tab(i + 1, 0)[0],
                                         ld_table=nt3, gamma_text=pp(f'{pn}.1.layernorm_gamma'), mean=_p(self.stats, 0, self._li(i + 1)), rstd=_p(self.stats, 1, self._li(i + 1)))
                    fused_pre[i + 1] = nxt
                self._keep = getattr(self, '_keep', []) + [a_postf, a_ar, nxt]
                self._raw(L, capi.lib().tfx_layer_end_fwd, ctypes.addressof(a_postf), ctypes.addressof(a_ar), ctypes.addressof(nxt) if nxt is not None else None)
                # x, y in; hidden i+1 out; hiddens 0..i in (the new one comes from registers); result out (+ its rounding residual and the depth softmax
                # for the pull-form backward); next wrapper's input out
                self._hbm(L, 3 + (i + 1) + 1 + (1 if self.pull else 0) + (1 if nxt is not None else 0), (16 * (i + 2) * T if self.pull else 0) + 8 * T)
            else:
                L.append(('tfx_adaln_post_fwd', a_postf))
                L.append(('tfx_attnres_fwd', a_ar))
        self._k(L, 'tfx_rmsnorm_fwd', 'tfx_rmsnorm_args', T=T, d=d, x=self.xres[D], y=self.embed, gamma=pp('transformer.norm.gamma'))
        self._hbm(L, 2)
        self.fwd_embed_end = len(L)          # launches up to here produce `embed` (return_embed / decode paths stop here)
        self._nt(L, A=self.embed, lda=d, B=S['logits'], ldb=d, M=T, N=md.vp, K=d, algo_n=md.vocab, epi=E['TFX_EPI_F32'], C=self.logits, ldc=md.vp)   # zero pad rows: N % 4 == 0 keeps the LDS-DMA kernel
        self.fwd_logits_end = len(L)
        if self.cache is not None:
            # decode plans: flow prediction only (model_to_latent on the modality rows), no losses
            for t, r in self.R.items():
                if t in self.ext:
                    continue
                dl = md.dim_latents[t]; lt = self.lat[t]
                self._nt(L, A=self.embed, lda=d, a_rowmap=self.row_src[t], B=S[f'outp{t}'], ldb=d, M=r, N=dl, K=d, epi=E['TFX_EPI_F32'], C=lt['pred'], ldc=dl)
            if md.model_output_clean:        # decode: always the model-space form (T:2446-2456), see below
                for t, r in self.R.items():
                    if t not in self.ext:
                        self._clean_flow(L, t, r)
            self.fwd_pred_end = len(L)
            self._chain_weight_prefetch(L)
            return
        native = {t: r for t, r in self.R.items() if t not in self.ext}      # types whose projections / losses run here
        for t, r in native.items():          # flow predictions first, so that a loss-free forward can stop at fwd_pred_end
            dl = md.dim_latents[t]; lt = self.lat[t]
            self._nt(L, A=self.embed, lda=d, a_rowmap=self.row_gat[t], B=S[f'outp{t}'], ldb=d, M=r, N=dl, K=d, epi=E['TFX_EPI_F32'], C=lt['pred'], ldc=dl)
        if md.model_output_clean:            # pred <- (pred - noised) / max(1 - t, eps): the model predicts the clean latent (MP:100-126)
            for t, r in native.items():
                self._clean_flow(L, t, r)
        self.fwd_pred_end = len(L)
        self._ce_args = capi.make_args('tfx_ce_args', T=T, V=md.vocab, logits=self.logits, ld=md.vp, labels=self.labels, grad_scale=0.0,
                                       dlogits=self.dlogits, ld_d=md.vp, acc=self.acc)
        L.append(('tfx_ce_fwd_bwd', self._ce_args))
        L.meta = L.meta or {}
        L.meta[len(L) - 1] = ('hbm', T * md.vp * (4 + 2))                    # fp32 logits in, bf16 d logits out
        self._mse_args = {}
        for t, r in native.items():
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            clean = dict(row_inst=self.row_inst[t], inst_time=self.inst_time, clean_eps=float(md.clean_eps)) if md.model_output_clean else {}
            self._mse_args[t] = capi.make_args('tfx_mse_args', R=r, dl=dl, pred=lt['pred'], ld_pred=dl, flow=lt['flow'], grad_scale=0.0,
                                               dpred=lt['dpred'], ld_d=dlp, acc=self.acc.data_ptr() + 4 * (2 + t), **clean)
            L.append(('tfx_mse_fwd_bwd', self._mse_args[t]))
        self._vel_args = {}
        for t, r in native.items():          # second target on the same prediction; dpred accumulates.  Run only when a teacher is given
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            lt['vel'] = torch.zeros(r, dl, device=self.ps.device, dtype=torch.float32)
            self._vel_args[t] = capi.make_args('tfx_mse_args', R=r, dl=dl, pred=lt['pred'], ld_pred=dl, flow=lt['vel'], grad_scale=0.0,
                                               dpred=lt['dpred'], ld_d=dlp, acc=self.acc.data_ptr() + 4 * (2 + len(md.dim_latents) + t), accumulate=1,
                                               **(dict(row_inst=self.row_inst[t], inst_time=self.inst_time, clean_eps=float(md.clean_eps)) if md.model_output_clean else {}))
            self.vel.append(('tfx_mse_fwd_bwd', self._vel_args[t]))
        self._rec_args = {}
        for t, r in native.items():          # third target on the same prediction: the reconstruction residual, per-row weights filled per step
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            lt['recw'] = torch.zeros(r, device=self.ps.device, dtype=torch.float32)
            self._rec_args[t] = capi.make_args('tfx_mse_args', R=r, dl=dl, pred=lt['pred'], ld_pred=dl, flow=lt['flow'], grad_scale=0.0,
                                               dpred=lt['dpred'], ld_d=dlp, acc=self.acc.data_ptr() + 4 * (2 + 2 * len(md.dim_latents) + t), accumulate=1,
                                               recon_w=lt['recw'], recon_inst=self.row_inst[t], recon_time=self.inst_time, recon_mode=0,
                                               **(dict(row_inst=self.row_inst[t], inst_time=self.inst_time, clean_eps=float(md.clean_eps)) if md.model_output_clean else {}))
            self.rec.append(('tfx_mse_fwd_bwd', self._rec_args[t]))

    def _chain_weight_prefetch(self, L):
        """Decode / prefill plans (round 6): every GEMM launch names the weights of the NEXT GEMM of the list in `prefetch` (tfx.h) - spare blocks of the launch touch
        them, so they wait in the Infinity Cache when their own launch comes.  A decode forward is a chain of ~190 dependent launches of a few dozen blocks that
        stream 25 MB of weights per layer; a GEMM's time is a chain of memory round trips, shorter from the cache than from HBM (tools/decode_cold_probe.py).
        64-row text step 2.32 -> 2.00 ms per forward, 512-row modality evaluation 3.34 -> 3.26, config 5 -1.5 %.  The last GEMM of a pass (logits / flow
        prediction) names the first layer's projection for the next pass.  TFX_DECODE_PREFETCH=0: off (A/B).  Results do not depend on it."""
        if os.environ.get('TFX_DECODE_PREFETCH', '1') == '0':
            return
        nts = [(k, e[1]) for k, e in enumerate(L) if not isinstance(e, Side) and e[0] == 'tfx_gemm_nt']
        first_layer = next((a for k, a in nts if k >= self.fwd_cond[1]), None)
        for (k, a), (_, nxt) in zip(nts, nts[1:] + [(None, first_layer)]):
            if k + 1 in (self.fwd_logits_end, self.fwd_pred_end):         # a pass ends here: the next GEMM that runs is the first layer's
                nxt = first_layer
            if nxt is None or not nxt.B:
                continue
            a.prefetch = nxt.B
            a.prefetch_bytes = int(nxt.N) * int(nxt.ldb) * 2

    def _qknr_separate(self, L, i, li, lkv, gam, plan_kw):
        """QK-RMSNorm + RoPE as its own token-wise launch behind the plain projection, then the attention launch (decode plans; TFX_QKNR=0)"""
        md, T, H, hd, ldq = self.md, self.T, self.md.heads, self.md.hdk, self.md.ldq
        # decode plans: the KV-cache append (k~ | v rows at `cache_pos`, T:1005-1016) rides in the same launch - a decode step is launch-bound
        ck = dict(cache=self.cache[i], ld_cache=2 * hd, cache_pos=self.cache_pos) if self.cache is not None else {}
        self._k(L, 'tfx_qk_norm_rope_fwd', 'tfx_qk_norm_rope_args', T=T, H=H, qkv=self.qkvg[lkv], ld_qkv=ldq, qk=self.qkr[lkv], ld_qk=2 * hd,
                gamma_q=gam('q'), gamma_k=gam('k'), rot_pos=self.rot_pos,
                cos_tab=0, sin_tab=0, q_scale=md.dim_head ** -0.5, norm_scale=md.dim_head ** 0.5, **ck, **plan_kw)
        self._rope_args = getattr(self, '_rope_args', []) + [L[-1][1]]
        L.meta = L.meta or {}
        L.meta[len(L) - 1] = ('hbm', 2 * 2 * T * hd * 2)                # q, k in; q~, k~ out
        self._k(L, 'tfx_attn_fwd' if (self.cache is None or self.tile_attn) else 'tfx_decode_attn', 'tfx_attn_args', **self._attn_kw(i))

    def _clean_q(self, L, t, r):
        """q = W proj of `_clean_flow`: one GEMM over the type's projected token rows"""
        md, d = self.md, self.md.dim
        dl = md.dim_latents[t]; lt = self.lat[t]
        lt['q'] = torch.empty(r, dl, device=self.ps.device, dtype=torch.float32)
        self._nt(L, A=lt['proj'], lda=lt['proj'].shape[1], B=self.ps.shadows[f'outp{t}'], ldb=d, M=r, N=dl, K=d, epi=E['TFX_EPI_F32'], C=lt['q'], ldc=dl)

    def _clean_flow(self, L, t, r):
        """`model_output_clean` (T:1297).  Two conversions exist in the reference:
          latent space (forward_modality, T:2772-2810):   flow = (model_to_latent(embed) - x_t) / max(1 - t, eps)
          model space  (interleaved forward MP:786-792, sample_many T:2446-2456):   flow = model_to_latent((embed - proj) / max(1 - t, eps)),
                        proj = the projected noised tokens the transformer consumed
        model_to_latent is linear without bias, so the model-space form is (W embed - W proj) / max(1 - t, eps): the cancellation of the two
        O(1) model-space vectors is carried out AFTER the projection, between two fp32 GEMM results `pred` and `q = W proj` - never in bf16.
        Both forms run through ONE launch (`tfx_output_to_flow`): `set_clean_mode` points its subtrahend at x / eps (latent) or at q (model)."""
        md = self.md
        dl = md.dim_latents[t]; lt = self.lat[t]
        self._clean_launch[t] = [lt['pred'].data_ptr(), lt['q'].data_ptr() if self.clean_mode == 'model' else lt['x'].data_ptr(), None,
                                 self.row_inst[t].data_ptr(), self.inst_time.data_ptr(), r, dl, float(md.clean_eps)]
        L.append((capi.lib().tfx_output_to_flow, self._clean_launch[t]))

    def set_clean_mode(self, mode: str):
        """'latent' (forward_modality) or 'model' (interleaved forward / sampler): which subtrahend `tfx_output_to_flow` uses, and whether the
        backward adds the gradient paths through `q = W proj` (clean_bwd)"""
        assert mode in ('latent', 'model')
        self.clean_mode = mode
        for t, a in self._clean_launch.items():
            lt = self.lat[t]
            a[1] = lt['q'].data_ptr() if mode == 'model' else lt['x'].data_ptr()
            a[2] = None if mode == 'model' else self.noise_args[t].eps

    def _attn_kw(self, i, bwd=False):
        md, hd, ldq = self.md, self.md.hdk, self.md.ldq
        li, lkv = self._li(i), self._lkv(i)
        kw = dict(q=self.qkr[lkv], k=_p(self.qkr, lkv) + 2 * hd, v=_p(self.qkvg, lkv) + 2 * 2 * hd, ld_q=2 * hd, ld_k=2 * hd, ld_v=ldq,
                  gate=_p(self.qkvg, lkv) + 2 * 3 * hd, ld_gate=ldq, kv_end=self.kv_end, q_start=self.q_start, out=self.og[li], ld_out=hd,
                  lse=self.lse[li], b=self.units[0] if self.units else self.b, h=md.heads, n=self.units[1] if self.units else self.n, softcap=50.0)
        if self.units:
            kw.update(q_row0=self.unit_row0, q_cnt=self.unit_cnt)
        if self.sc_plan is not None:
            kw['sc_plan'] = self.sc_plan[i]
        if self.cache is not None:
            ck = self.cache[i]
            kw.update(k=ck.data_ptr(), v=ck.data_ptr() + 2 * hd, ld_k=2 * hd, ld_v=2 * hd, n_kv=int(ck.shape[1]))
        if bwd:
            dqkvg = self.dqkvg_p[i % len(self.dqkvg_p)]
            kw.update(dout=self.dog, ld_dout=hd, do_eff=self.do_eff, ld_do=hd, delta=self.delta,
                      dgate=dqkvg.data_ptr() + 2 * 3 * hd, ld_dgate=ldq, dq=self.dqk if self.dqk is not None else 0,
                      dk=self.dqk.data_ptr() + 2 * hd if self.dqk is not None else 0,
                      dv=dqkvg.data_ptr() + 2 * 2 * hd, ld_dq=2 * hd, ld_dk=2 * hd, ld_dv=ldq)
        return kw

    def set_rope_tables(self, cos_tab, sin_tab):
        if self.cos_tab is not None and (self.cos_tab.data_ptr() != cos_tab.data_ptr() or self.sin_tab.data_ptr() != sin_tab.data_ptr()):
            self.fwd.invalidate_graphs()                 # captured kernel arguments hold the old table pointers
        self.cos_tab, self.sin_tab = cos_tab, sin_tab
        for a in getattr(self, '_rope_args', []):
            a.cos_tab, a.sin_tab = cos_tab.data_ptr(), sin_tab.data_ptr()
        for a in getattr(self, '_rope_nt_args', []):                      # projections with the fused QK-norm / RoPE epilogue
            a.qk_cos, a.qk_sin = cos_tab.data_ptr(), sin_tab.data_ptr()
        for a in getattr(self, '_rope_attn_args', []):                    # attention backward with the fused QK-norm / RoPE backward
            a.nr_cos, a.nr_sin = cos_tab.data_ptr(), sin_tab.data_ptr()

    def set_segments(self, seg_start, seg_len):
        n_seg = int(seg_start.numel())
        self.seg_start[:n_seg].copy_(seg_start); self.seg_len[:n_seg].copy_(seg_len)
        for a in self._seg_args:
            a.n_seg = n_seg

    def set_rows(self, R_true: dict):
        """latent rows of this step, per type (<= the plan's capacity `R`: training plans are built for row counts rounded up, so that ragged batches
        share them).  Rows past the real ones scatter nowhere (row_tok = -1) and gather row 0 (row_gat); the kernels whose RESULT would see them -
        the noising, the flow losses, the bias column sum - run over the real rows only, and the operands the weight-gradient GEMMs still
        multiply them with (xt, dpred) are zero there."""
        for t, r in R_true.items():
            assert r <= self.R[t]
            for obj, key in [(self.noise_args.get(t), 'R'), (getattr(self, '_mse_args', {}).get(t), 'R'), (getattr(self, '_vel_args', {}).get(t), 'R'),
                             (getattr(self, '_rec_args', {}).get(t), 'R'), *self._rows_dep[t]]:
                if obj is None:
                    continue
                if isinstance(obj, list):
                    obj[key] = r
                else:
                    setattr(obj, key, r)
            lt = self.lat[t]
            if r < self.R[t] and 'xt' in lt:
                lt['xt'][r:].zero_()
                if 'dpred' in lt:
                    lt['dpred'][r:].zero_()
            if self.row_gat is not self.row_tok:
                torch.clamp(self.row_tok[t], min=0, out=self.row_gat[t])

    def set_noise(self, t: int, eps_ptr):
        """noise source of modality type t for this run: a device pointer (training: x_t = t x + (1 - t) eps) or None (no noising)"""
        self.noise_args[t].eps = eps_ptr
        if t in getattr(self, '_clean_launch', {}) and self.clean_mode == 'latent':
            self._clean_launch[t][2] = eps_ptr

    def set_ce_vocab(self, V: int):
        """number of logit columns the cross entropy runs over (the full vocabulary, or the text-only prefix for forward_text)"""
        self._ce_args.V = V

    def set_loss_scales(self, ce_scale: float, mse_scales: dict):
        self._ce_args.grad_scale = ce_scale
        for t, s in mse_scales.items():
            self._mse_args[t].grad_scale = s

    # ------------------------------------------------------------------------------------ backward
    def _build_backward(self):
        ps, md, T, I = self.ps, self.md, self.T, self.I
        d, hd, D, di, dip, ldq, nt3, H = md.dim, md.hdk, md.depth, md.di, md.dip, md.ldq, md.nt3, md.heads
        S = ps.shadows
        L = self.bwd
        pp, gp = ps.ptr, ps.grad_ptr
        lib = capi.lib()
        gmap = ps._maps['geglu']
        # model-space `model_output_clean`: pred = (W embed - W proj) s, so the (already s-scaled) seed also flows, negated, through q = W proj:
        #   d proj = -(dpred W)  ->  latent_to_model weight / bias gradients;   dW += (-dpred)^T proj.   Run before the main list when the mode is 'model'.
        self.clean_bwd = LaunchList()
        if md.model_output_clean:
            C = self.clean_bwd
            for t, r in self.R.items():
                if t in self.ext:
                    continue
                dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
                lt['ndpred'] = torch.zeros(r, dlp, device=self.ps.device, dtype=BF16)
                self._raw(C, lib.tfx_scale_bf16_copy, lt['dpred'].data_ptr(), lt['ndpred'].data_ptr(), r * dlp, -1.0)
                self._tn(C, r, dl, d, A=lt['ndpred'], lda=dlp, a_cols=dlp, B=lt['proj'], ldb=lt['proj'].shape[1], b_cols=d,
                         C=gp(f'model_to_latent_projs.{t}.weight'), ldc=d)
                if dl != d:
                    lt['dpx'] = torch.empty(r, d, device=self.ps.device, dtype=BF16)
                    self._nt(C, algo_k=dl, A=lt['ndpred'], lda=dlp, B=S[f'outp_t{t}'], ldb=dlp, M=r, N=d, K=dlp, epi=E['TFX_EPI_BF16'], C=lt['dpx'], ldc=d)
                    self._tn(C, r, d, dl, A=lt['dpx'], lda=d, a_cols=d, B=lt['xt'], ldb=dlp, b_cols=dlp, C=gp(f'latent_to_model_projs.{t}.weight'), ldc=dl)
                    self._raw(C, lib.tfx_colsum_bf16, lt['dpx'].data_ptr(), d, r, d, None, None, gp(f'latent_to_model_projs.{t}.bias'))
        self._nt(L, algo_k=md.vocab, A=self.dlogits, lda=md.vp, B=S['logits_t'], ldb=md.vp, M=T, N=d, K=md.vp, epi=E['TFX_EPI_BF16'], C=self.dembed, ldc=d)
        self._tn(L, T, md.vocab, d, A=self.dlogits, lda=md.vp, a_cols=md.vp, B=self.embed, ldb=d, b_cols=d, C=gp('to_text_logits.weight'), ldc=d)
        for t, r in self.R.items():
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            if t in self.ext:     # d loss / d (embedding rows handed to the user's decoder), from autograd: dembed[row_tok] += gemb
                self._nt(L, A=lt['gemb'], lda=d, B=S['eye'], ldb=d, M=r, N=d, K=d, epi=E['TFX_EPI_RESID'], C=self.dembed, ldc=d,
                         R=self.dembed, ldr=d, resid_mapped=1, rowmap=self.row_tok[t])
                continue
            self._nt(L, algo_k=dl, A=lt['dpred'], lda=dlp, B=S[f'outp_t{t}'], ldb=dlp, M=r, N=d, K=dlp, epi=E['TFX_EPI_RESID'], C=self.dembed, ldc=d,
                     R=self.dembed, ldr=d, resid_mapped=1, rowmap=self.row_tok[t])
            self._tn(L, r, dl, d, A=lt['dpred'], lda=dlp, a_cols=dlp, B=self.embed, ldb=d, b_cols=d, b_rowmap=self.row_gat[t],
                     C=gp(f'model_to_latent_projs.{t}.weight'), ldc=d)
        self._k(L, 'tfx_rmsnorm_bwd', 'tfx_rmsnorm_args', T=T, d=d, x=self.xres[D], gamma=pp('transformer.norm.gamma'), dy=self.dembed,
                dx=self.gfin, dgamma=gp('transformer.norm.gamma'))
        self._hbm(L, 3)
        src = skip_sources(md)
        pushed = set(src.values())
        g = self.gfin
        side = self.side
        def sync(op, slot):
            if side:
                L.append((op, slot))
        pull = self.pull
        if pull:
            # pull-form AttentionResidual backward: one source record per layer (device array, lowest layer first) - the FINAL gradient of the
            # layer's AttentionResidual output (the backward's last write to that buffer), its saved softmax state, its w = (1 + gamma) pq
            self.dsum = torch.empty(D, T, device=ps.device, dtype=torch.float32)
            self.wtab = torch.empty(2, D, d, device=ps.device, dtype=torch.float32)          # w rows | d w accumulators
            self.nbytes += self.dsum.numel() * 4 + self.wtab.numel() * 4
            SRC = capi.STRUCTS['tfx_attnres_src']
            recs = (SRC * D)()
            for j in range(D):
                pj = f'transformer.layers.{j}.3'
                gj = self.gfin if j == D - 1 else (self.gx[j + 1] if md.has_skip(j + 1) else self.dH[j + 2])
                for k, v in dict(g=gj, save=self.arsave[j], dsum=self.dsum[j], w=self.wtab[0, j], dw=self.wtab[1, j], gamma=pp(f'{pj}.norm_keys.gamma'),
                                 pq=pp(f'{pj}.pseudo_queries'), dgamma=gp(f'{pj}.norm_keys.gamma'), dpq=gp(f'{pj}.pseudo_queries'), L=j + 2).items():
                    setattr(recs[j], k, v.data_ptr() if hasattr(v, 'data_ptr') else v)
            raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(recs), ctypes.sizeof(recs))), dtype=torch.uint8)
            self._src_tab = raw.to(ps.device)
            self._src_size = ctypes.sizeof(SRC)
            self._raw(L, lib.tfx_attnres_prep, self._src_tab.data_ptr(), D, d)
        def pull_launch(l, out_own, add, dh, post):
            """gradient of hidden l from the layers j >= max(l - 1, 0) that mixed it (+ the output side `post` of the wrapper that produced the hidden).
            d w of those layers accumulates inside the launch for few narrow sources; otherwise the launch exports the per-token coefficients and
            one weight-gradient GEMM K1^T . h adds them (rows j0.. of the d w table are contiguous)"""
            j0 = max(l - 1, 0)
            ns = D - j0
            export = ns > 8 or d > 512
            if export and not hasattr(self, 'k1buf'):
                self.k1buf = torch.zeros(T, 32, device=ps.device, dtype=BF16); self.nbytes += self.k1buf.numel() * 2
            a = capi.make_args('tfx_attnres_pull_args', T=T, d=d, l=l, n_src=ns, h=self.hid[l], src=self._src_tab.data_ptr() + j0 * self._src_size,
                               out_own=out_own, out_err=self.arerr[l - 1] if out_own is not None else None, add=add, dh=dh,
                               seg_start=self.seg_start, seg_len=self.seg_len, n_seg=0, k1=self.k1buf if export else None, ld_k1=32)
            self._seg_args.append(a)
            self._keep = getattr(self, '_keep', []) + [a, post]
            self._raw(L, lib.tfx_attnres_pull_bwd, ctypes.addressof(a), ctypes.addressof(post) if post is not None else None)
            self._hbm(L, ns + 6 if post is not None else ns + 4, 12 * ns * T)     # n_src gradient rows + h, out, err in, dh out (+ g, y in, dy out of the wrapper side); saved softmax state
            if export:
                self._tn(L, T, ns, d, A=self.k1buf, lda=32, a_cols=32, B=self.hid[l], ldb=d, b_cols=d, C=self.wtab[1, j0], ldc=d)
        for i in range(D - 1, -1, -1):
            p = f'transformer.layers.{i}'
            x_in = self.xres[i]
            x_a = self.xa[i] if md.has_skip(i) else x_in
            (ta, dta), (tf, dtf) = self._tab(i, 0), self._tab(i, 1)
            par = i % len(self.dag_p)
            dy_f, dy_a, dag, dqkvg = self.dy_f[par], self.dy_a[par], self.dag_p[par], self.dqkvg_p[par]
            if i + 2 <= D - 1:
                sync('tfx_join_wait', i + 2)       # this layer reuses the buffers of layer i+2: its weight gradients must have read them
            G = self.dH[i + 1]
            a_postf = capi.make_args('tfx_adaln_post_args', T=T, d=d, y=self.yf[i], tok_inst=self.tok_inst, table=tf, ld_table=nt3,
                                     layerscale=pp(f'{p}.2.layerscale'), g=G, dy=dy_f, dtable=dtf, dlayerscale=gp(f'{p}.2.layerscale'),
                                     seg_start=self.seg_start, seg_len=self.seg_len, n_seg=0, dbias=gp(f'{p}.2.fn.net.3.bias'))   # ff2 bias gradient = column sums of dy
            self._seg_args.append(a_postf)
            if pull:
                # dH[i+1] = gradient of hidden i + 1 from the AttentionResiduals of layers i .. D-1 (all final), formed ONCE; the feed-forward
                # wrapper's output side rides in the same launch
                pull_launch(i + 1, self.xres[i + 1], None, G, a_postf)
            else:
                g2 = self.dskip[i + 1] if (i + 1) in pushed else None
                self._k(L, 'tfx_attnres_bwd', 'tfx_attnres_args', T=T, d=d, L=i + 2, hiddens=self.hid, stride_h=T * d,
                        gamma=pp(f'{p}.3.norm_keys.gamma'), pq=pp(f'{p}.3.pseudo_queries'), g=g, g2=capi.ptr(g2), dhiddens=self.dH, stride_dh=T * d,
                        first=1 if i == D - 1 else 0, dgamma=gp(f'{p}.3.norm_keys.gamma'), dpq=gp(f'{p}.3.pseudo_queries'))
                # ---- feedforward wrapper
                L.append(('tfx_adaln_post_bwd', a_postf))
            self._nt(L, algo_n=di, A=dy_f, lda=d, B=S[f'ff2_t{i}'], ldb=d, M=T, N=dip, K=d, epi=E['TFX_EPI_GEGLU_BWD'], C=dag, ldc=2 * dip,
                     aux=self.ag[i], ldaux=2 * dip)
            # the weight gradients of this wrapper go to the side stream (dy_f and d[a|g] are final); net.0 carries its bias gradient
            # (column sums of d[a|g]) folded into the same GEMM
            sync('tfx_fork', 2 * i)
            # (round 5 tried net.3's bias gradient as this GEMM's `colsum` to free 8 registers of the pull kernel: the SUM form of the GEMM is 13 us slower per
            # launch (158 vs 145 us, profiles/r05_shapes.txt) - and with its scale row in LDS the pull kernel no longer spills WITH the bias partials)
            ff_grp = [(d, di, dict(A=dy_f, lda=d, a_cols=d, B=self.hm[i], ldb=dip, b_cols=dip, C=gp(f'{p}.2.fn.net.3.weight'), ldc=di)),
                      (2 * dip, d, dict(algo_n=2 * di, A=dag, lda=2 * dip, a_cols=2 * dip, B=self.uf[i], ldb=d, b_cols=d, rowmap=gmap,
                                        C=gp(f'{p}.2.fn.net.0.weight'), ldc=d, colsum=gp(f'{p}.2.fn.net.0.bias')))]
            # (one launch per LAYER - these two waiting for the attention wrapper's products - measured 4.21 -> 3.97 ms on the family and nothing on the overlapped
            #  step, profiles/r05b_ab_tn_layer_group.txt: removed in round 6)
            self._tn_group(L, T, ff_grp, side=side)
            self._nt(L, algo_k=2 * di, A=dag, lda=2 * dip, B=S[f'ff1_t{i}'], ldb=2 * dip, M=T, N=d, K=2 * dip, epi=E['TFX_EPI_BF16'], C=self.du, ldc=d)
            a_pref = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=self.xb[i], tok_inst=self.tok_inst, table=tf, ld_table=nt3,
                                    gamma_text=pp(f'{p}.2.layernorm_gamma'), mean=_p(self.stats, 2, i), rstd=_p(self.stats, 3, i), du=self.du, dx=G,
                                    dtable=dtf, dgamma_text=gp(f'{p}.2.layernorm_gamma'), seg_start=self.seg_start, seg_len=self.seg_len, n_seg=0)
            # ---- attention wrapper
            a_posta = capi.make_args('tfx_adaln_post_args', T=T, d=d, y=self.ya[i], tok_inst=self.tok_inst, table=ta, ld_table=nt3,
                                     layerscale=pp(f'{p}.1.layerscale'), g=G, dy=dy_a, dtable=dta, dlayerscale=gp(f'{p}.1.layerscale'),
                                     seg_start=self.seg_start, seg_len=self.seg_len, n_seg=0)
            self._seg_args += [a_pref, a_posta]
            # input side of the feed-forward wrapper + output side of the attention wrapper: the residual-gradient row is written once, not read back
            self._keep = getattr(self, '_keep', []) + [a_pref, a_posta]
            self._raw(L, lib.tfx_adaln_pre_post_bwd, ctypes.addressof(a_pref), ctypes.addressof(a_posta))
            self._hbm(L, 6)                                             # du, x in, residual gradient in / out, y in, dy out
            self._nt(L, algo_n=md.hd, A=dy_a, lda=d, B=S[f'out_t{i}'], ldb=d, M=T, N=hd, K=d, epi=E['TFX_EPI_BF16'], C=self.dog, ldc=hd)
            gam = (lambda nm: S[f'g{nm}{i}']) if md.dim_head != 64 else (lambda nm: pp(f'{p}.1.fn.{nm}_norm.gamma'))
            if os.environ.get('TFX_ATTN_QKNR', '1') != '0':
                # round 5: the backward of QK-RMSNorm + RoPE rides in the epilogues of the dQ and dK/dV kernels (tfx.h tfx_attn_args.nr_*): d q~ | d k~ are never
                # written out and read back, and the token-wise launch below drops out of the layer
                self._k(L, 'tfx_attn_bwd', 'tfx_attn_args', **self._attn_kw(i, bwd=True), nr_qkv=self.qkvg[i], nr_ld_qkv=ldq, nr_dqkv=dqkvg, nr_ld_dqkv=ldq,
                        nr_gamma_q=gam('q'), nr_gamma_k=gam('k'), nr_rot_pos=self.rot_pos, nr_cos=0, nr_sin=0, nr_q_scale=md.dim_head ** -0.5,
                        nr_norm_scale=md.dim_head ** 0.5, nr_dgamma_q=gp(f'{p}.1.fn.q_norm.gamma'), nr_dgamma_k=gp(f'{p}.1.fn.k_norm.gamma'))
                # (per-block partial rows + a reduction launch instead of the 64 same-address atomics per block measured SLOWER in round 5 - 4.19 vs 4.05 ms -
                # and were removed in round 6)
                self._rope_attn_args = getattr(self, '_rope_attn_args', []) + [L[-1][1]]
            else:
                self._k(L, 'tfx_attn_bwd', 'tfx_attn_args', **self._attn_kw(i, bwd=True))
                self._k(L, 'tfx_qk_norm_rope_bwd', 'tfx_qk_norm_rope_args', T=T, H=H, qkv=self.qkvg[i], ld_qkv=ldq, gamma_q=gam('q'),
                        gamma_k=gam('k'), rot_pos=self.rot_pos, cos_tab=0, sin_tab=0, q_scale=md.dim_head ** -0.5, norm_scale=md.dim_head ** 0.5,
                        dqk=self.dqk, ld_dqk=2 * hd, dqkv=dqkvg, ld_dqkv=ldq, dgamma_q=gp(f'{p}.1.fn.q_norm.gamma'), dgamma_k=gp(f'{p}.1.fn.k_norm.gamma'))
                self._rope_args = getattr(self, '_rope_args', []) + [L[-1][1]]
                L.meta = L.meta or {}
                L.meta[len(L) - 1] = ('hbm', 3 * 2 * T * hd * 2)                # q, k (raw) in; d q~, d k~ in; d q, d k out
            self._nt(L, algo_k=md.nq, A=dqkvg, lda=ldq, B=S[f'qkvg_t{i}'], ldb=ldq, M=T, N=d, K=ldq, epi=E['TFX_EPI_BF16'], C=self.du, ldc=d)
            # (pull form: the gradient a U-Net skip hands to this layer's input joins here, so that G ends up as the TOTAL gradient of xres[i])
            self._k(L, 'tfx_adaln_pre_bwd', 'tfx_adaln_pre_args', T=T, d=d, x=x_a, tok_inst=self.tok_inst, table=ta, ld_table=nt3,
                    gamma_text=pp(f'{p}.1.layernorm_gamma'), mean=_p(self.stats, 0, i), rstd=_p(self.stats, 1, i), du=self.du, dx=G,
                    dtable=dta, dgamma_text=gp(f'{p}.1.layernorm_gamma'), seg_start=self.seg_start, seg_len=self.seg_len, n_seg=0,
                    dx_add=self.dskip[i] if (pull and i in pushed) else None)
            self._hbm(L, 4 + (1 if (pull and i in pushed) else 0))          # du, x in; residual gradient in / out (+ the U-Net skip's share)
            self._seg_args.append(L[-1][1])
            # weight gradients of the attention wrapper (dy_a, d[q|k|v|gates] and G = dH[i+1] are final) on the side stream
            sync('tfx_fork', 2 * i + 1)
            grp = [(d, hd, dict(A=dy_a, lda=d, a_cols=d, B=self.og[i], ldb=hd, b_cols=hd, C=gp(f'{p}.1.fn.to_out.1.weight'), ldc=md.hd,
                                k_group=0 if md.dim_head == 64 else md.dim_head)),
                   (md.nqk, d, dict(algo_n=md.nq, A=dqkvg, lda=ldq, a_cols=ldq, B=self.ua[i], ldb=d, b_cols=d, C=gp(f'{p}.1.fn.to_qk.0.weight'), ldc=d,
                                    rowmap=ps._maps['heads'] if md.dim_head != 64 else None))]
            if md.has_skip(i):
                sk = self.xres[src[i]]
                # d W_skip [d, 2 d] = G^T [x | skip]: two members of the layer's group (as ONE split-B product it measured 117.6 us against 2 x 64 us,
                # profiles/r05b_ab_skip_tn_split.txt: the split-B form was removed in round 6)
                grp.append((d, d, dict(A=G, lda=d, a_cols=d, B=x_in, ldb=d, b_cols=d, C=gp(f'{p}.0.weight'), ldc=2 * d)))
                grp.append((d, d, dict(A=G, lda=d, a_cols=d, B=sk, ldb=d, b_cols=d, C=gp(f'{p}.0.weight', d), ldc=2 * d)))
            self._tn_group(L, T, grp, side=side)
            per = -(-D // self.dp_groups) if self.dp_groups > 0 else D
            if I > 0 and i % per == 0:
                # AdaLN conditioning weights (6d table columns per layer, 63 % of all parameters) of the layer GROUP that ends here: their table
                # gradients are final once the group's backward is.  With the overlapped gradient exchange (dp_groups > 0) the weight / bias
                # gradients are formed group by group, so that the gradient buffer completes back to front and a group's all-reduce can start
                # during the backward; one GEMM per group (>= 768 tiles: no split, every block reduces all I instances).  Without an exchange
                # (dp_groups == 0: one GPU, or one all-reduce after the backward) the "group" is the whole stack: ONE GEMM after layer 0
                hi = min(i + per, D)
                off, cols = i * 2 * 3 * d, (hi - i) * 2 * 3 * d
                w_item = lambda it: L.append(Side(it) if side else it)
                w_item((lib.tfx_cast_block_bf16, (self.dtables.data_ptr() + 4 * off, nt3, self.dtab_bf.data_ptr() + 2 * off, nt3, I, cols)))
                self._tn(L, I, cols, 4 * d, side=side, A=self.dtab_bf.data_ptr() + 2 * off, lda=nt3, a_cols=cols, B=self.cond, ldb=4 * d, b_cols=4 * d,
                         C=gp(f'{p}.1.to_film.weight'), ldc=4 * d)
                w_item((lib.tfx_colsum_f32, (self.dtables.data_ptr() + 4 * off, nt3, I, cols, gp(f'{p}.1.to_film.bias'))))
            sync('tfx_join_record', i)
            if self.dp_groups > 0:
                if i % per == 0:                                     # lowest layer of a group: every gradient of layers >= i is final
                    sync('tfx_join_wait', i)
                    self.bwd_cuts.append((len(L), i, min(i + per, D) - 1))
            if md.has_skip(i):
                st = S[f'skip_t{i}']
                self._nt(L, A=G, lda=d, B=st, ldb=d, M=T, N=d, K=d, epi=E['TFX_EPI_RESID'], C=self.gx[i], ldc=d, R=G, ldr=d)
                self._nt(L, A=G, lda=d, B=st[d:], ldb=d, M=T, N=d, K=d, epi=E['TFX_EPI_BF16'], C=self.dskip[src[i]], ldc=d)
                g = self.gx[i]
            else:
                g = G
        sync('tfx_join', 63)                       # every weight gradient is complete before the list returns
        # ---- gradient wrt the transformer input x0 = hid[0] = xres[0]
        if pull:
            pull_launch(0, None, g, self.dx0, None)               # hidden 0 = the transformer input: every layer mixed it; + the chain gradient g
            self._raw(L, lib.tfx_attnres_finish, self._src_tab.data_ptr(), D, d)
        else:
            self._raw(L, lib.tfx_add_bf16, g.data_ptr(), self.dH[0].data_ptr(), self.dx0.data_ptr(), T * d)
            if 0 in pushed:
                self._raw(L, lib.tfx_add_bf16, self.dx0.data_ptr(), self.dskip[0].data_ptr(), self.dx0.data_ptr(), T * d)
        self._raw(L, lib.tfx_onehot_bf16, self.text_ids.data_ptr(), self.tok_inst.data_ptr(), self.onehot.data_ptr(), T, md.vp)
        self._tn(L, T, md.vocab, d, A=self.onehot, lda=md.vp, a_cols=md.vp, B=self.dx0, ldb=d, b_cols=d, C=gp('text_embed.weight'), ldc=d)
        for t, r in self.R.items():
            dl = md.dim_latents[t]; dlp = pad_to(dl, 64); lt = self.lat[t]
            if dl == d or t in self.ext:
                continue                                   # Identity latent_to_model / the user's encoder: no parameters here (`dx0` rows go back through autograd)
            self._tn(L, r, d, dl, A=self.dx0, lda=d, a_cols=d, a_rowmap=self.row_gat[t], B=lt['xt'], ldb=dlp, b_cols=dlp,
                     C=gp(f'latent_to_model_projs.{t}.weight'), ldc=dl)
            cs = [self.dx0.data_ptr(), d, r, d, None, self.row_tok[t].data_ptr(), gp(f'latent_to_model_projs.{t}.bias')]      # (mutable: the row count follows the step)
            self._rows_dep[t].append((cs, 2))
            L.append((lib.tfx_colsum_bf16, cs))
        if I > 0:
            # (dtab_bf = bf16 table gradients, cast layer by layer above - the join before this point covers the side stream)
            self._nt(L, A=self.dtab_bf, lda=nt3, B=S['ada_t'], ldb=nt3, M=I, N=4 * d, K=nt3, epi=E['TFX_EPI_BF16'], C=self.dcond, ldc=4 * d)
            self._raw(L, lib.tfx_silu_bwd, self.dcond.data_ptr(), self.pre.data_ptr(), self.dpre.data_ptr(), I * 4 * d)
            self._tn(L, I, 4 * d, d + 1, A=self.dpre, lda=4 * d, a_cols=4 * d, B=self.fe, ldb=md.kf, b_cols=md.kf,
                     C=gp('transformer.to_time_cond.1.weight'), ldc=d + 1)
            self._raw(L, lib.tfx_colsum_bf16, self.dpre.data_ptr(), 4 * d, I, 4 * d, None, None, gp('transformer.to_time_cond.1.bias'))

    # ------------------------------------------------------------------------------------ run
    @staticmethod
    def run(launches, stream, lo=0, hi=None, graph=False):
        """replay launches[lo:hi] on `stream`: one `tfx_run_list` call for a LaunchList (the product path), a per-launch loop for a
        plain list (tools that bracket individual launches).  graph=True: replay the captured hipGraph of the range (decode plans);
        graph='auto': training lists - as a graph while nothing a capture freezes has changed (TFX_TRAIN_GRAPH=1, LaunchList.replay_auto)."""
        lib = capi.lib()
        sp = ctypes.c_void_p(stream)
        if isinstance(launches, LaunchList):
            n = len(launches)
            lo = max(0, lo if lo >= 0 else n + lo)
            hi = n if hi is None else min(n, hi if hi >= 0 else n + hi)
            if hi <= lo:
                return
            if graph == 'auto':
                if TRAIN_GRAPH and launches.replay_auto(lo, hi, sp):
                    return
            elif graph:
                rc = lib.tfx_graph_launch(ctypes.c_void_p(launches.graph(lo, hi)), sp)
                if rc != 0:
                    raise capi.TfxError(f'tfx_graph_launch failed with code {rc}')
                return
            arr = launches.native()
            failed = ctypes.c_int32(-1)
            rc = lib.tfx_run_list(ctypes.byref(arr, lo * ctypes.sizeof(_LAUNCH)), hi - lo, sp, ctypes.byref(failed))
            if rc != 0:
                fn = launches[lo + failed.value][0] if failed.value >= 0 else 'tfx_run_list'
                raise capi.TfxError(f'{fn if isinstance(fn, str) else fn.__name__} failed with code {rc}')
            return
        for item in launches[lo:hi]:
            fn, a = item
            if isinstance(fn, str) and fn in _SYNC_OPS:
                continue                                           # one stream, in list order: forks / joins are no-ops
            if isinstance(fn, str):
                rc = getattr(lib, fn)(ctypes.byref(a), sp)
            else:
                rc = fn(*a, sp)
            if rc != 0:
                raise capi.TfxError(f'{fn if isinstance(fn, str) else fn.__name__} failed with code {rc}')
