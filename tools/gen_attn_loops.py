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
        if os.environ.get("GEN_EMPTY") and not s.startswith(("s_sub_u32 %[cnt]", "s_cmp", "s_cbranch", "L_af")): return       # (timing build: the loop's control flow only)
        if os.environ.get("GEN_NOVALU") and s.startswith(("v_mul", "v_fma", "v_exp", "v_add", "v_cvt")): return      # (timing build)
        if os.environ.get("GEN_NOMFMA") and s.startswith("v_mfma"): return
        self.lines.append(s)
    def lds_read(self, name, text):
        if os.environ.get("GEN_NOLDS") or os.environ.get("GEN_EMPTY"): return                    # (timing build)
        self.lines.append(text)
        self.lds_q.append(name)
        assert len(self.lds_q) <= 15, "lgkmcnt is a 4-bit counter"
    def need(self, names):
        """wait until the LDS reads `names` have landed (in-order return): allow everything issued after the youngest of them."""
        idx = [i for i, n in enumerate(self.lds_q) if n in names]
        if not idx:
            return
        last = max(idx)
        allow = len(self.lds_q) - 1 - last
        self.lines.append(f"s_waitcnt lgkmcnt({allow})")
        self.lds_q = self.lds_q[last + 1:]
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


def write_inc(out, L, what):
    with open(out, "w") as f:
        f.write(f"// GENERATED by tools/gen_attn_loops.py ({what}) - do not edit; the schedule is documented there.\n")
        f.write(f"// Fixed registers v{B0}-v{TOP - 1}: scores v{SA}-v{SB + 15}, P^T v{PA}-v{PB + 7}, row sums v{L0}/v{L1}, V^T fragments v{VF}-v{VF + 15}, K fragments v{KF}-v{KF + 15}, temporaries v{TMP}-v{TMP + 3}\n")
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


if __name__ == "__main__":
    main()
