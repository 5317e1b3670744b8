#!/usr/bin/env python3
"""Generator of the hand-scheduled main loops of the attention kernels (transfusion_pytorch_amd/csrc/attention.hip).

    python tools/gen_attn_loops.py        # writes csrc/attn_fwd_loop_m0.inc, attn_fwd_loop_m1.inc (soft-cap plan modes 0 / 1)

Why a generator (round 6).  The soft-capped softmax costs ~22 VALU clocks per score against 16 MFMA clocks per score, so the forward is
VALU-bound and its floor is the bare vector stream: 88 instructions per 32 x 32 score block and wave (mode 0).  The hipcc-scheduled pipe kernel issues 148 (address
arithmetic of every fragment read, copies between register sets) and waits on every LDS read a few instructions after issuing it (SQ_WAIT_INST_ANY 26 %): 3 x the
floor.  Here the loop over the tiles that need NO mask (every key visible to every row of the block) is ONE asm statement with fixed registers:

  unit u (32 keys) = one PHASE of 8 chunks; chunk c = [MFMA c] + the vector work of scores 2c, 2c+1 of the unit:
      MFMA 0..3   S^T of unit u + 1   (K fragments read one phase earlier, q fragments resident)
      MFMA 4..7   P.V of unit u - 1    (V^T fragments by ds_read_b64_tr_b16 in chunks 0 / 1 of this phase, P packed one phase earlier)
      vector      s (p1 + p3 s^2) [+ p5 s^4] -> v_exp -> row sums -> v_cvt_pk_bf16 : 11 (13) instructions per chunk, exp results consumed one chunk later
      LDS         8 transposed V reads in chunks 0 / 1, the K fragment of k-step ks re-read (for unit u + 2) in chunk ks + 1, right behind the MFMA that consumed it
  every LDS address is a lane VGPR computed once + an immediate (ring slot, key block, 16-key group): the 3-slot K / V rings are unrolled (tiles j % 3 = 0, 1, 2);
  K / V tiles arrive by `buffer_load_dwordx4 ... offen lds` (lane offset VGPR, tile offset SGPR, LDS target in M0), protocol of the C++ loop: at the top of tile j
  `vmcnt(0)` + `s_barrier`, then K(j + 2) and V(j + 1) are requested - so the statement can hand over to the C++ loop (the masked boundary tiles) at any tile.
  `s_waitcnt lgkmcnt(N)` values are computed by the generator from the in-order LDS queue.

Same arithmetic, same per-accumulator MFMA order as the C++ phases (fwd_phase in attention.hip): outputs are bit-identical to attn_fwd_pipe_kernel<false>.
"""
import argparse
import os

B0 = 64                       # first VGPR of the fixed block
SA, SB = B0, B0 + 16          # scores of the unit in flight (raw S^T from the MFMAs, then s2, then exp2) - two sets, alternating
PA, PB = B0 + 32, B0 + 40     # P^T of a unit as MFMA operands: [tt] = 4 packed bf16 pairs per 16 keys
L0, L1 = B0 + 48, B0 + 49     # row sums (even / odd score slots)
VF = B0 + 52                  # 4 V^T fragments (tt, db) of 4 registers
KF = B0 + 68                  # 4 K fragments (ks) of 4 registers
TMP = B0 + 84                 # 4 temporaries
TOP = B0 + 88


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


class Emit:
    def __init__(self):
        self.lines = []
        self.lds_q = []           # outstanding LDS reads, oldest first (names)
    def op(self, s):
        if os.environ.get("GEN_EMPTY") and not s.startswith(("s_sub_u32 %[cnt]", "s_cmp_eq_u32 %[cnt]", "s_cbranch_scc1 L_af_exit", "s_cbranch_scc0 L_af_loop", "s_cbranch_scc1 L_dq_exit", "s_cbranch_scc0 L_dq_loop", "L_af_loop", "L_af_exit", "L_dq_loop", "L_dq_exit")): return       # (timing build: the loop's control flow only)
        if os.environ.get("GEN_NOVALU") and s.startswith(("v_mul", "v_fma", "v_exp", "v_add", "v_cvt", "v_sub", "v_cmp", "v_cndmask")): return      # (timing build)
        if os.environ.get("GEN_NOEXP") and s.startswith("v_exp"): return
        if os.environ.get("GEN_NOMFMA") and s.startswith("v_mfma"): return
        self.lines.append(s)
    def lds_read(self, name, text):
        if os.environ.get("GEN_NOLDS") or os.environ.get("GEN_EMPTY"): return                    # (timing build)
        self.lines.append(text)
        self.lds_q.append(name)
    def need(self, names):
        """wait until the LDS reads `names` have landed (in-order return): allow everything issued after the youngest of them."""
        idx = [i for i, n in enumerate(self.lds_q) if n in names]
        if not idx:
            return
        last = max(idx)
        allow = min(len(self.lds_q) - 1 - last, 15)               # (a 4-bit counter: more than 15 younger reads in flight = wait until 15 are left)
        self.lines.append(f"s_waitcnt lgkmcnt({allow})")
        self.lds_q = self.lds_q[len(self.lds_q) - allow:] if allow else []
    def drain(self):
        if self.lds_q:
            self.lines.append("s_waitcnt lgkmcnt(0)")
            self.lds_q = []


def k_read(e, ks, slot, kb, tag):
    e.lds_read(f"K{ks}", f"ds_read_b128 {vr(KF + 4 * ks, 4)}, %[ka{ks}] offset:{slot * 8192 + kb * 4096}")


def v_reads(slot, kb):
    """the 8 transposed reads of the 4 V^T fragments (tt, db) of key block kb in ring slot `slot`: (name, text)."""
    out = []
    for f in range(4):
        tt, db = f >> 1, f & 1
        off = slot * 8192 + kb * 4096 + tt * 2048
        out.append((f"V{f}a", f"ds_read_b64_tr_b16 {vr(VF + 4 * f, 2)}, %[va{db}] offset:{off}"))
        out.append((f"V{f}", f"ds_read_b64_tr_b16 {vr(VF + 4 * f + 2, 2)}, %[vb{db}] offset:{off}"))
    return out


def phase(e, mode, h, kslot_next, kb_next, vslot_prev, kb_prev, first=False, last_reads=True):
    """one unit.  h: 0 = scores in SA, P -> PA, S of the next unit -> SB, P.V of PB ; 1 = the other way round.
    K fragments of S(u + 1) are in KF (read one phase earlier); this phase re-reads KF for S(u + 2) from (kslot_next, kb_next).
    V^T fragments of P.V(u - 1) come from (vslot_prev, kb_prev).  first: the block's very first unit - no P.V (there is no unit -1)."""
    s_cur, s_nxt = (SA, SB) if h == 0 else (SB, SA)
    p_cur, p_prev = (PA, PB) if h == 0 else (PB, PA)
    vreads = [] if first else v_reads(vslot_prev, kb_prev)
    for c in range(8):
        # ---- the chunk's MFMA
        if c < 4:
            e.need([f"K{c}"])
            cc = "0" if c == 0 else vr(s_nxt, 16)
            e.op(f"v_mfma_f32_32x32x16_bf16 {vr(s_nxt, 16)}, {vr(KF + 4 * c, 4)}, %[qf{c}], {cc}")
        elif not first:
            f = c - 4
            tt, db = f >> 1, f & 1
            e.need([f"V{f}"])
            e.op(f"v_mfma_f32_32x32x16_bf16 %[o{db}], {vr(VF + 4 * f, 4)}, {vr(p_prev + 4 * tt, 4)}, %[o{db}]")
        # ---- vector work of scores a = 2c, b = 2c + 1 ; LDS reads spread between the instructions
        a, b = s_cur + 2 * c, s_cur + 2 * c + 1
        t0, t1 = TMP + 2 * (c & 1), TMP + 2 * (c & 1) + 1
        lds = []
        if c == 0: lds = vreads[0:4]
        if c == 1: lds = vreads[4:8]
        if 1 <= c <= 4 and last_reads: lds = lds + [("K", c - 1)]
        def put_lds():
            if lds:
                x = lds.pop(0)
                if x[0] == "K": k_read(e, x[1], kslot_next, kb_next, None)
                else: e.lds_read(x[0], x[1])
        e.op(f"v_mul_f32 {vr(t0)}, {vr(a)}, {vr(a)}")
        e.op(f"v_mul_f32 {vr(t1)}, {vr(b)}, {vr(b)}")
        put_lds()
        if mode == 0:
            e.op(f"v_fma_f32 {vr(t0)}, {vr(t0)}, %[p3], %[p1]")
            e.op(f"v_fma_f32 {vr(t1)}, {vr(t1)}, %[p3], %[p1]")
            put_lds()
            e.op(f"v_mul_f32 {vr(a)}, {vr(a)}, {vr(t0)}")
            e.op(f"v_mul_f32 {vr(b)}, {vr(b)}, {vr(t1)}")
        else:                      # s (p1 + u (p3 + u p5)) : p5 scalar, p3 / p1 vector constants
            t2, t3 = TMP + 2 * ((c + 1) & 1), TMP + 2 * ((c + 1) & 1) + 1
            e.op(f"v_fma_f32 {vr(t2)}, {vr(t0)}, %[p5], %[p3]")
            e.op(f"v_fma_f32 {vr(t3)}, {vr(t1)}, %[p5], %[p3]")
            put_lds()
            e.op(f"v_fma_f32 {vr(t2)}, {vr(t0)}, {vr(t2)}, %[p1]")
            e.op(f"v_fma_f32 {vr(t3)}, {vr(t1)}, {vr(t3)}, %[p1]")
            e.op(f"v_mul_f32 {vr(a)}, {vr(a)}, {vr(t2)}")
            e.op(f"v_mul_f32 {vr(b)}, {vr(b)}, {vr(t3)}")
        put_lds()
        def addcvt(cq):
            x, y = s_cur + 2 * cq, s_cur + 2 * cq + 1
            e.op(f"v_add_f32 {vr(L0)}, {vr(L0)}, {vr(x)}")
            e.op(f"v_add_f32 {vr(L1)}, {vr(L1)}, {vr(y)}")
            e.op(f"v_cvt_pk_bf16_f32 {vr(p_cur + cq)}, {vr(x)}, {vr(y)}")
        if c >= 1: addcvt(c - 1)
        put_lds()
        e.op(f"v_exp_f32 {vr(a)}, {vr(a)}")
        e.op(f"v_exp_f32 {vr(b)}, {vr(b)}")
        while lds: put_lds()
        if c == 7: addcvt(7)


def tile_top(e, i):
    """top of tile j (j % 3 = i): everything requested so far has landed for every wave; K(j + 2) -> slot (i + 2) % 3, V(j + 1) -> slot (i + 1) % 3."""
    if not os.environ.get("GEN_NOWAIT"): e.op("s_waitcnt vmcnt(0)")       # (timing builds, wrong results: GEN_NOWAIT / GEN_NOBAR / GEN_NODMA)
    if not os.environ.get("GEN_NOBAR"): e.op("s_barrier")
    ks, vs = (i + 2) % 3, (i + 1) % 3
    for base, slot, dv, rs, so in (("%[mk]", ks, "dk", "%[rsk]", "%[sko]"), ("%[mv]", vs, "dv", "%[rsv]", "%[svo]")):
        for jp in range(2):
            e.op(f"s_add_u32 m0, {base}, {slot * 8192 + jp * 1024}")
            e.op("s_nop 0")
            if not os.environ.get("GEN_NODMA"): e.op(f"buffer_load_dwordx4 %[{dv}{jp}], {rs}, {so} offen lds")


def tile(e, mode, i, first=False):
    """tile j, j % 3 = i.  K(j) in K slot i, V(j) in V slot i."""
    tile_top(e, i)
    if first:
        # pipeline fill: S of unit 0 from K(0) first half, then the K fragments of S(1)
        for ks in range(4): k_read(e, ks, 0, 0, None)
        for ks in range(4):
            e.need([f"K{ks}"])
            cc = "0" if ks == 0 else vr(SA, 16)
            e.op(f"v_mfma_f32_32x32x16_bf16 {vr(SA, 16)}, {vr(KF + 4 * ks, 4)}, %[qf{ks}], {cc}")
        for ks in range(4): k_read(e, ks, 0, 1, None)
        e.op("s_nop 15")                                           # MFMA result -> vector read: covered below by MFMA 0 + the nop
    n1 = (i + 1) % 3
    # even unit 2j: S(2j + 1) = K(j) second half (fragments resident); P.V(2j - 1) = V(j - 1) second half; re-read for S(2j + 2) = K(j + 1) first half
    phase(e, mode, 0, n1, 0, (i + 2) % 3, 1, first=first)
    # the tile offsets of the NEXT requests move here, a phase away from the buffer instructions on either side: a VMEM instruction reads its scalar operands some
    # clocks AFTER it issues - with the two s_add right behind the four requests the V pieces went out with the next tile's offset (first GPU trip: V(j + 2) in V(j + 1)'s slot)
    e.op("s_add_u32 %[sko], %[sko], %[stk]")
    e.op("s_add_u32 %[svo], %[svo], %[stv]")
    # odd unit 2j + 1: S(2j + 2); P.V(2j) = V(j) first half; re-read for S(2j + 3) = K(j + 1) second half
    phase(e, mode, 1, n1, 1, i, 0)


def program_fwd(mode):
    e = Emit()
    e.op("s_nop 4")                                                # scalar operands fresh from readfirstlane -> buffer instruction
    tile(e, mode, 0, first=True)
    e.op("s_sub_u32 %[cnt], %[cnt], 1")
    e.op("s_cmp_eq_u32 %[cnt], 0")
    e.op("s_cbranch_scc1 L_af_exit_%=")
    for i in (1, 2):
        tile(e, mode, i)
        e.op("s_sub_u32 %[cnt], %[cnt], 1")
        e.op("s_cmp_eq_u32 %[cnt], 0")
        e.op("s_cbranch_scc1 L_af_exit_%=")
    q_entry = list(e.lds_q)
    e.op("L_af_loop_%=:")
    for i in (0, 1, 2):
        tile(e, mode, i)
        e.op("s_sub_u32 %[cnt], %[cnt], 1")
        e.op("s_cmp_eq_u32 %[cnt], 0")
        if i < 2: e.op("s_cbranch_scc1 L_af_exit_%=")
        else: e.op("s_cbranch_scc0 L_af_loop_%=")
    assert e.lds_q == q_entry, (e.lds_q, q_entry)                   # the LDS queue is the same at the back edge as at the loop's entry
    e.op("L_af_exit_%=:")
    e.lds_q = list(q_entry)
    e.drain()
    e.op("s_nop 15")                                               # MFMA results are read by compiler code from here on
    e.op("s_nop 15")
    return e.lines


# =====================================================================================================================
# backward dQ (attn_bwd_dq_asm_kernel): the WHOLE tile loop - fill, every unit incl. the masked boundary units, drain
# =====================================================================================================================
# Measured (profiles/r06_attn_bwd_what_bounds_it.txt): the hipcc kernels issue ~470 instructions per 32-key unit and wave where the arithmetic needs ~175
# (136 vector, 12 MFMA, 16 LDS reads, waits) - fragment-address arithmetic, register copies, s_nop padding - and run at one instruction per ~2.9 clocks and
# SIMD whatever the mix: the kernels are ISSUE-bound, a software pipeline at the C++ level (attn_bwd_dq_pipe_kernel) changed nothing (226 vs 227 us).
# Unit u = one phase of 8 chunks; chunk c = its MFMAs + the vector work of scores 2c, 2c + 1:
#     MFMAs   S0 dP0 | S1 | dP1 S2 | dP2 | S3 dP3 | dQ0 | dQ1 dQ2 | dQ3      S / dP of unit u + 1 (operands read one phase earlier), dQ += dS K of unit u - 1
#     vector  u = s^2, tanh' = d1 + d3 u, s2 = s (p1 + p3 u) - lse2, P = exp2, dS = P ((dP - delta) tanh') -> bf16 pairs                  17 (21) per chunk
#     LDS     every operand fragment has its own registers and is re-read for the NEXT phase right behind the MFMA that consumed it
# K and V tiles in rings of four (one LDS array: V slot s at +32 KiB), requested two tiles ahead, ONE barrier per tile; the rings are unrolled (tiles j % 4).
DQ0 = 64
D_SA, D_DPA, D_SB, D_DPB = DQ0, DQ0 + 16, DQ0 + 32, DQ0 + 48
D_DSA, D_DSB = DQ0 + 64, DQ0 + 72
D_KF, D_VF, D_TF = DQ0 + 80, DQ0 + 96, DQ0 + 112
D_TMP = DQ0 + 128              # 12 temporaries
D_MT = DQ0 + 140               # mask scratch
D_TOP = DQ0 + 141


def dq_reads(e, kind, idx, slot, kb):
    """(re)read one operand fragment of the NEXT phase: kind K / V = row fragment k-step idx of (slot, kb) ; T = K^T fragment (tt, db) = idx of (slot, kb)."""
    if kind == "K":
        e.lds_read(f"K{idx}", f"ds_read_b128 {vr(D_KF + 4 * idx, 4)}, %[ka{idx}] offset:{slot * 8192 + kb * 4096}")
    elif kind == "V":
        e.lds_read(f"V{idx}", f"ds_read_b128 {vr(D_VF + 4 * idx, 4)}, %[ka{idx}] offset:{32768 + slot * 8192 + kb * 4096}")
    else:
        tt, db = idx >> 1, idx & 1
        off = slot * 8192 + kb * 4096 + tt * 2048
        e.lds_read(f"T{idx}a", f"ds_read_b64_tr_b16 {vr(D_TF + 4 * idx, 2)}, %[ta{db}] offset:{off}")
        e.lds_read(f"T{idx}", f"ds_read_b64_tr_b16 {vr(D_TF + 4 * idx + 2, 2)}, %[tb{db}] offset:{off}")


def dq_mask(e, s_s, s_dp):
    """boundary unit: a masked score gets dP = delta (dS = P x 0 = 0) and the score 0 (P = exp2(-lse2): finite - a large masked score against a small lse would give
    inf x 0).  element r of this lane is key u * 32 + 4 hi + (r & 3) + 8 (r >> 2); visible iff that < kv_end."""
    e.op(f"v_subrev_u32 {vr(D_MT)}, %[su1], %[kmaskp]")           # kv_end - 4 hi + 32 - (u + 1) * 32 = keys of this unit the lane-half sees, counted from its first
    for r in range(16):
        cr = (r & 3) + 8 * (r >> 2)
        e.op(f"v_cmp_lt_i32 vcc, {cr}, {vr(D_MT)}")
        e.op("s_nop 1")
        e.op(f"v_cndmask_b32 {vr(s_dp + r)}, %[dlt], {vr(s_dp + r)}, vcc")
        e.op(f"v_cndmask_b32 {vr(s_s + r)}, 0, {vr(s_s + r)}, vcc")


def dq_phase(e, mode, h, nxt, prv, tag, first=False):
    """h: 0 = vector work on (SA, DPA) -> DSA, S / dP of the next unit -> (SB, DPB), dQ with DSB ; 1 = the other way round.
    nxt = (kslot, vslot, kb) of the unit whose S / dP fragments are read here for the next phase ; prv = (kslot, kb) of the unit whose K^T fragments are read here."""
    s_cur, dp_cur, s_nxt, dp_nxt = (D_SA, D_DPA, D_SB, D_DPB) if h == 0 else (D_SB, D_DPB, D_SA, D_DPA)
    ds_cur, ds_prev = (D_DSA, D_DSB) if h == 0 else (D_DSB, D_DSA)
    # ---- units past the wave's last visible key (u * 32 >= max kv_end of the wave) contribute exact zeros: the wave stops computing - it still runs the tile
    #      tops (requests, barriers).  Its first dead phase owes the dQ of the unit before it (dS is packed, its K^T fragments are on their way); `live` then
    #      drops and every later phase, and the drain, fall through.  (6 of the 72 (wave, unit) pairs of a 128-row block at n = 1024; the kernel is power-bound.)
    e.op("s_cmp_eq_u32 %[live], 0")
    e.op(f"s_cbranch_scc1 L_dq_end_{tag}_%=")
    e.op("s_sub_u32 %[stmp], %[su1], 32")
    e.op("s_cmp_ge_i32 %[stmp], %[kvemax]")
    e.op(f"s_cbranch_scc0 L_dq_live_{tag}_%=")
    e.op("s_mov_b32 %[live], 0")
    if not first:
        e.op("s_waitcnt lgkmcnt(0)")
        for i in range(4):
            tt, db = i >> 1, i & 1
            e.op(f"v_mfma_f32_32x32x16_bf16 %[dq{db}], {vr(D_TF + 4 * i, 4)}, {vr(ds_prev + 4 * tt, 4)}, %[dq{db}]")
    e.op(f"s_branch L_dq_end_{tag}_%=")
    e.op(f"L_dq_live_{tag}_%=:")
    # ---- boundary units: wave-uniform test (u + 1) * 32 > min kv_end of the wave
    e.op("s_cmp_gt_i32 %[su1], %[kvemin]")
    e.op(f"s_cbranch_scc0 L_dq_nomask_{tag}_%=")
    dq_mask(e, s_cur, dp_cur)
    e.op(f"L_dq_nomask_{tag}_%=:")
    sched = [["S0", "P0"], ["S1"], ["P1", "S2"], ["P2"], ["S3", "P3"], ["Q0"], ["Q1", "Q2"], ["Q3"]]
    valu = dq_valu(mode, s_cur, dp_cur, ds_cur)
    for c in range(8):
        pending = []
        for m in sched[c]:
            k, i = m[0], int(m[1])
            if k == "S":
                e.need([f"K{i}"])
                cc = "0" if i == 0 else vr(s_nxt, 16)
                e.op(f"v_mfma_f32_32x32x16_bf16 {vr(s_nxt, 16)}, {vr(D_KF + 4 * i, 4)}, %[qf{i}], {cc}")
                pending.append(("K", i))
            elif k == "P":
                e.need([f"V{i}"])
                cc = "0" if i == 0 else vr(dp_nxt, 16)
                e.op(f"v_mfma_f32_32x32x16_bf16 {vr(dp_nxt, 16)}, {vr(D_VF + 4 * i, 4)}, %[df{i}], {cc}")
                pending.append(("V", i))
            elif not first:
                tt, db = i >> 1, i & 1
                e.need([f"T{i}"])
                e.op(f"v_mfma_f32_32x32x16_bf16 %[dq{db}], {vr(D_TF + 4 * i, 4)}, {vr(ds_prev + 4 * tt, 4)}, %[dq{db}]")
                pending.append(("T", i))
            else:
                pending.append(("T", i))
        def put():
            if pending:
                k, i = pending.pop(0)
                if k == "K": dq_reads(e, "K", i, nxt[0], nxt[2])
                elif k == "V": dq_reads(e, "V", i, nxt[1], nxt[2])
                else: dq_reads(e, "T", i, prv[0], prv[1])
        # vector work of this chunk: its share of the phase's latency-ordered stream (dq_valu), LDS re-reads spread through it
        ops = valu[c]
        k = max(1, len(ops) // (len(pending) + 1)) if pending else len(ops) + 1
        for n, o in enumerate(ops):
            e.op(o)
            if pending and (n + 1) % k == 0: put()
        while pending: put()
    e.op(f"L_dq_end_{tag}_%=:")
    e.op("s_add_u32 %[su1], %[su1], 32")


def dq_valu(mode, s_cur, dp_cur, ds_cur):
    """the 136 (168) vector instructions of a unit, ordered for LATENCY: four scores at a time, stage by stage - a result is read four instructions after it was
    issued, not two - and the products / packing of a group of four ride inside the NEXT group (exp2 results are consumed ~14 instructions later).
    (First GPU trip of the asm kernel: with two scores per chunk in dependency order the VECTOR-ONLY build of the loop ran at 5.4 clocks per instruction and SIMD,
    twice the forward's rate - two waves per SIMD do not cover back-to-back dependent instructions.)  Returns the stream cut into 8 chunks."""
    out = []
    u = [D_TMP + k for k in range(4)]
    th = [D_TMP + 4 + k for k in range(4)]
    x = [D_TMP + 8 + k for k in range(4)]                           # (mode 1 only)
    def tail(g):
        r = [4 * g + k for k in range(4)]
        L = [f"v_mul_f32 {vr(s_cur + q)}, {vr(s_cur + q)}, {vr(dp_cur + q)}" for q in r]
        L += [f"v_cvt_pk_bf16_f32 {vr(ds_cur + 2 * g + h)}, {vr(s_cur + 4 * g + 2 * h)}, {vr(s_cur + 4 * g + 2 * h + 1)}" for h in range(2)]
        return L
    for g in range(4):
        r = [4 * g + k for k in range(4)]
        L = [f"v_mul_f32 {vr(u[k])}, {vr(s_cur + r[k])}, {vr(s_cur + r[k])}" for k in range(4)]
        if mode == 0:
            L += [f"v_fma_f32 {vr(th[k])}, {vr(u[k])}, %[d3], %[d1]" for k in range(4)]
        else:
            L += [f"v_fma_f32 {vr(th[k])}, {vr(u[k])}, %[d5], %[d3]" for k in range(4)]
            L += [f"v_fma_f32 {vr(x[k])}, {vr(u[k])}, %[p5], %[p3]" for k in range(4)]
            L += [f"v_fma_f32 {vr(th[k])}, {vr(u[k])}, {vr(th[k])}, %[d1]" for k in range(4)]
        L += [f"v_sub_f32 {vr(dp_cur + r[k])}, {vr(dp_cur + r[k])}, %[dlt]" for k in range(4)]
        if g > 0: L += tail(g - 1)
        if mode == 0:
            L += [f"v_fma_f32 {vr(u[k])}, {vr(u[k])}, %[p3], %[p1]" for k in range(4)]
        else:
            L += [f"v_fma_f32 {vr(u[k])}, {vr(u[k])}, {vr(x[k])}, %[p1]" for k in range(4)]
        L += [f"v_mul_f32 {vr(dp_cur + r[k])}, {vr(dp_cur + r[k])}, {vr(th[k])}" for k in range(4)]
        L += [f"v_fma_f32 {vr(s_cur + r[k])}, {vr(s_cur + r[k])}, {vr(u[k])}, -%[lse2]" for k in range(4)]
        L += [f"v_exp_f32 {vr(s_cur + r[k])}, {vr(s_cur + r[k])}" for k in range(4)]
        if g == 3: L += tail(3)
        out += L
    n = len(out)
    cuts = [round(n * c / 8) for c in range(9)]
    return [out[cuts[c]:cuts[c + 1]] for c in range(8)]


def dq_tile_top(e, i, tag):
    """top of tile j (j % 4 = i): K(j + 1) / V(j + 1) have landed for every wave; K(j + 2), V(j + 2) -> slot (i + 2) % 4 while rem2 = nt - 2 - j > 0."""
    e.op("s_waitcnt vmcnt(0)")
    e.op("s_barrier")
    e.op("s_cmp_gt_i32 %[rem2], 0")
    e.op(f"s_cbranch_scc0 L_dq_nodma_{tag}_%=")
    slot = (i + 2) % 4
    for base, dv, rs, so in ((slot * 8192, "dk", "%[rsk]", "%[sko]"), (32768 + slot * 8192, "dv", "%[rsv]", "%[svo]")):
        for jp in range(2):
            e.op(f"s_add_u32 m0, %[mk], {base + jp * 1024}")
            e.op("s_nop 0")
            e.op(f"buffer_load_dwordx4 %[{dv}{jp}], {rs}, {so} offen lds")
    e.op(f"L_dq_nodma_{tag}_%=:")
    e.op("s_sub_u32 %[rem2], %[rem2], 1")


def dq_tile(e, mode, i, tag, first=False):
    dq_tile_top(e, i, tag)
    if first:
        for ks in range(4): dq_reads(e, "K", ks, 0, 0); dq_reads(e, "V", ks, 0, 0)
        for ks in range(4):
            e.need([f"K{ks}"])
            e.op(f"v_mfma_f32_32x32x16_bf16 {vr(D_SA, 16)}, {vr(D_KF + 4 * ks, 4)}, %[qf{ks}], " + ("0" if ks == 0 else vr(D_SA, 16)))
            e.need([f"V{ks}"])
            e.op(f"v_mfma_f32_32x32x16_bf16 {vr(D_DPA, 16)}, {vr(D_VF + 4 * ks, 4)}, %[df{ks}], " + ("0" if ks == 0 else vr(D_DPA, 16)))
        for ks in range(4): dq_reads(e, "K", ks, 0, 1); dq_reads(e, "V", ks, 0, 1)      # S / dP of unit 1
        e.op("s_nop 15")
    n1 = (i + 1) % 4
    # even unit 2j: reads here = S / dP fragments of unit 2j + 2 (first halves of K(j + 1), V(j + 1)) and K^T of unit 2j (first half of K(j))
    dq_phase(e, mode, 0, (n1, n1, 0), (i, 0), tag + "a", first=first)
    # the tile offsets of the next requests move a phase away from the buffer instructions (a VMEM instruction reads its scalar operands after it issues)
    e.op("s_add_u32 %[sko], %[sko], %[stk]")
    e.op("s_add_u32 %[svo], %[svo], %[stv]")
    # odd unit 2j + 1: reads = S / dP of unit 2j + 3 (second halves of K(j + 1), V(j + 1)) and K^T of unit 2j + 1 (second half of K(j))
    dq_phase(e, mode, 1, (n1, n1, 1), (i, 1), tag + "b")


def program_dq(mode):
    e = Emit()
    e.op("s_nop 4")
    def tail_check(last):
        e.op("s_sub_u32 %[cnt], %[cnt], 1")
        e.op("s_cmp_eq_u32 %[cnt], 0")
        if last: e.op("s_cbranch_scc0 L_dq_loop_%=")
        else: e.op("s_cbranch_scc1 L_dq_exit_%=")
    dq_tile(e, mode, 0, "p0", first=True)
    tail_check(False)
    for i in (1, 2, 3):
        dq_tile(e, mode, i, f"p{i}")
        tail_check(False)
    q_entry = list(e.lds_q)
    e.op("L_dq_loop_%=:")
    for i in (0, 1, 2, 3):
        dq_tile(e, mode, i, f"l{i}")
        tail_check(i == 3)
    assert e.lds_q == q_entry, (e.lds_q, q_entry)
    e.op("L_dq_exit_%=:")
    e.lds_q = list(q_entry)
    # drain: dQ of the very last unit (DSB; its K^T fragments were read by the last phase) - unless the wave stopped early
    e.op("s_cmp_eq_u32 %[live], 0")
    e.op("s_cbranch_scc1 L_dq_nodrain_%=")
    for f in range(4):
        tt, db = f >> 1, f & 1
        e.need([f"T{f}"])
        e.op(f"v_mfma_f32_32x32x16_bf16 %[dq{db}], {vr(D_TF + 4 * f, 4)}, {vr(D_DSB + 4 * tt, 4)}, %[dq{db}]")
    e.drain()
    e.op("L_dq_nodrain_%=:")
    e.op("s_waitcnt lgkmcnt(0)")
    e.op("s_nop 15")
    e.op("s_nop 15")
    return e.lines


def write_inc(out, L, what, top=None):
    with open(out, "w") as f:
        f.write(f"// GENERATED by tools/gen_attn_loops.py ({what}) - do not edit; the schedule is documented there.\n")
        if top is None: f.write(f"// Fixed registers v{B0}-v{TOP - 1}: scores v{SA}-v{SB + 15}, P^T v{PA}-v{PB + 7}, row sums v{L0}/v{L1}, V^T fragments v{VF}-v{VF + 15}, K fragments v{KF}-v{KF + 15}, temporaries v{TMP}-v{TMP + 3}\n")
        else: f.write(f"// Fixed registers v{DQ0}-v{top - 1} (see the register map in the generator)\n")
        for ln in L:
            f.write('"' + ln + '\\n\\t"\n')
    n_mfma = sum(1 for ln in L if ln.startswith("v_mfma"))
    print(f"wrote {out}: {len(L)} lines, {n_mfma} MFMAs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default=None)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    outdir = a.outdir or os.path.join(here, "..", "transfusion_pytorch_amd", "csrc")
    for mode in (0, 1):
        write_inc(os.path.join(outdir, f"attn_fwd_loop_m{mode}.inc"), program_fwd(mode), f"forward, unmasked tiles, soft-cap plan mode {mode}")
        write_inc(os.path.join(outdir, f"attn_dq_loop_m{mode}.inc"), program_dq(mode), f"backward dQ, whole tile loop, soft-cap plan mode {mode}", D_TOP)
    with open(os.path.join(outdir, "attn_asm_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_attn_loops.py - the fixed registers of the generated attention loops, as clobber lists\n")
        f.write("#define TFX_DQ_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(DQ0, D_TOP)) + "\n")


if __name__ == "__main__":
    main()
