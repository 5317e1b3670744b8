#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of `gemm_nt_ow_kernel` (transfusion_pytorch_amd/csrc/gemm.hip).

Writes transfusion_pytorch_amd/csrc/gemm_nt_ow_loop.inc: ONE inline-asm statement (prologue, steady-state K-tile, the two tail K-tiles)
whose register operands are bound by the C++ side (`OW_ASM_OPERANDS` in gemm.hip, same numbering as below).

Why a generator: the loop is 64 MFMAs per K-tile with every LDS read, LDS-DMA issue, scalar update, counted wait and barrier placed by hand
in the 32-clock gaps between the MFMAs of a lone wave (one wave per SIMD, 128 x 128 per wave, 256 accumulator AGPRs): hipcc's scheduler
re-serialises such a loop (LAB_NOTEBOOK round 4: two hipcc-scheduled kernels of this shape landed on the ping-pong kernel's speed).

Schedule of one K-tile (gap g = the slot behind MFMA g; P = the LDS buffer of K-tile kt, Q = the other one):
   g 0..7    read the A fragments of the K-tile's second half (k-steps 2, 3) from P
   g 10/11   lgkmcnt(0) ; s_barrier                                                          -> nobody reads P's A tile any more
   g 12..33  the 8 A pieces of K-tile kt+2 into P by LDS-DMA, one per three gaps (A first: it comes from HBM, B = weights from L2)
   g 12..19  read the B fragments of the second half ; g 22/23 lgkmcnt(0) ; s_barrier        -> nobody reads P any more
   g 25..39  (odd) the 8 fragment-read addresses move from P to Q
   g 36..57  the 8 B pieces, one per three gaps
   g 43/44   vmcnt(<pieces issued so far>) ; s_barrier                                       -> K-tile kt+1 has landed in Q, for every wave
   g 45..60  read the fragments of K-tile kt+1's first half (k-steps 0, 1) from Q
   g 58..62  lane offsets, DMA target, K-tile counter ; lgkmcnt(0)
(`--nosplit`: one barrier behind all sixteen second-half reads, DMA pieces from g 20 on - the first form, 2-3 % slower on the long-K shapes.)
Fragments of a whole K-tile live in registers (2 x 16 x 4 VGPRs), which is what frees P for the DMA of K-tile kt+2 a quarter into K-tile kt:
two K-tiles stay in flight with two 64 KiB LDS buffers.
"""
import argparse
import os

# defaults = the schedule that measured best in the power-limited steady state (gpurun_out/ow20.txt, ow21.txt: the split form -2.6 % on K >= 1408 shapes, -3 % on
# 1544 x 512 against one barrier behind all sixteen second-half reads; `--nosplit` restores that form)
OPT = argparse.Namespace(nodma=False, noread=False, reads0_per_gap=1, reads1_per_gap=1, wait1=10, bar1=11, dma_start=20, dma_step=3, n_before=12,
                         late_start=46, late_step=3, wait2=43, bar2=44, reads0_start=45, split=True, wait1b=22, bar1b=23, nobar=False, nosalu=False, nosplit=False)

# ---- operand numbers (must match OW_ASM_OPERANDS in gemm.hip) ----
def ACC(i, j):            # accumulator of row block i (A), column block j (B): acc[j >> 1][i][j & 1]
    return (j >> 1) * 8 + i * 2 + (j & 1)
def FA(i, ks):            # A fragment, row block i, k-step ks
    return 16 + (ks >> 1) * 16 + (ks & 1) * 4 + i
def FB(j, ks):
    return 16 + (ks >> 1) * 16 + 8 + (ks & 1) * 4 + j
def RA(ks): return 48 + ks
def RB(ks): return 52 + ks
VOA, VOB, SM, CNT, SO, DELTA, RSA, RSB, STA, STB = 56, 57, 58, 59, 60, 61, 62, 63, 64, 65

VOAN, VOBN = 66, 67       # (persistent kernel only) lane offsets of the NEXT tile's first K-tile

def mfma(m, zero=False):
    half, r = divmod(m, 32)
    ksl, r = divmod(r, 16)
    j, i = divmod(r, 4)
    ks = half * 2 + ksl
    a = ACC(i, j)
    c = "0" if (zero and ks == 0) else f"%{a}"            # a tile's first k-step starts its accumulators from the inline constant: no zeroing pass
    return f"v_mfma_f32_32x32x16_bf16 %{a}, %{FB(j, ks)}, %{FA(i, ks)}, {c}"

def frag_reads(half):
    """16 reads of one half K-tile, in order of first use (B0, A0..A3, B1..B3 of each k-step)."""
    out = []
    for ksl in range(2):
        ks = half * 2 + ksl
        out.append(f"ds_read_b128 %{FB(0, ks)}, %{RB(ks)}")
        for i in range(4):
            out.append(f"ds_read_b128 %{FA(i, ks)}, %{RA(ks)}" + (f" offset:{i * 4096}" if i else ""))
        for j in range(1, 4):
            out.append(f"ds_read_b128 %{FB(j, ks)}, %{RB(ks)} offset:{j * 4096}")
    return out

def dma(d):
    """(prep instructions, issue instruction) of DMA piece d of a K-tile: 0..7 = A pieces, 8..15 = B pieces."""
    op, jp = divmod(d, 8)
    vo, rs, st = (VOA, RSA, STA) if op == 0 else (VOB, RSB, STB)
    prep = []
    if jp == 0:
        prep.append(f"s_mov_b32 m0, %{SM}" if op == 0 else f"s_add_u32 m0, %{SM}, 0x8000")
        so = "0"
    else:
        prep.append("s_add_u32 m0, m0, 0x800")
        prep.append(f"s_mov_b32 %{SO}, %{st}" if jp == 1 else f"s_add_u32 %{SO}, %{SO}, %{st}")
        so = f"%{SO}"
    return prep, f"buffer_load_dwordx4 %{vo}, %{rs}, {so} offen lds"

def toggles():
    return [f"v_add_u32 %{RA(ks)}, %{DELTA}, %{RA(ks)}" for ks in range(4)] + [f"v_add_u32 %{RB(ks)}, %{DELTA}, %{RB(ks)}" for ks in range(4)]

def dma_gaps():
    """gap of each of the 16 DMA pieces of a steady K-tile."""
    o = OPT
    if o.split:          # A pieces behind the first barrier (A fragments read), B pieces behind the second (B fragments read)
        g, out = o.bar1 + 1, []
        for d in range(16):
            if d == 8: g = max(g, o.bar1b + 1)
            out.append(g); g += o.dma_step
        return out
    return [o.dma_start + o.dma_step * d if d < o.n_before else o.late_start + o.late_step * (d - o.n_before) for d in range(16)]

def body(kind, zero=False, loop="L_ow_steady_%=", nof0=False):
    """kind: 'steady' (DMA of K-tile kt+2, reads of kt+1), 't1' (K-tile nk-2: no DMA), 't2' (last K-tile: second-half reads only).
    zero: the K-tile is a tile's first (accumulators start from 0); loop: label the steady body branches back to while more than 2 K-tiles are left (None: straight-line);
    nof0: certify the next K-tile (wait + barrier) but leave the reads of its first fragments to the caller (persistent kernel: a tile's last K-tile)."""
    o = OPT
    fill = [[] for _ in range(64)]
    timing_steady = kind == 'steady'
    def put_reads(reads, g0, per_gap):
        for q, r in enumerate(reads):
            if not (o.noread and timing_steady): fill[g0 + q // per_gap].append(r)
    r1 = frag_reads(1)
    if o.split and kind == 'steady':
        ra = [r for r in r1 if any(f"%{FA(i, ks)}," in r for i in range(4) for ks in (2, 3))]
        rb = [r for r in r1 if r not in ra]
        assert len(ra) == 8 and len(rb) == 8
        put_reads(ra, 0, o.reads1_per_gap)
        fill[o.wait1].append("s_waitcnt lgkmcnt(0)")         # A fragments landed
        fill[o.bar1].append("s_barrier")                     # every wave has read P's A tile for the last time
        put_reads(rb, o.bar1 + 1, o.reads1_per_gap)
        fill[o.wait1b].append("s_waitcnt lgkmcnt(0)")        # B fragments landed (needed from MFMA 32 on)
        fill[o.bar1b].append("s_barrier")
    else:
        put_reads(r1, 0, o.reads1_per_gap)
        fill[o.wait1 if not o.split else 18].append("s_waitcnt lgkmcnt(0)")      # ALL second-half fragments landed (needed from MFMA 32 on; the reads end at gap 15)
    tog0 = (o.bar1b if o.split else o.bar1) + 2
    if kind == 'steady':
        if not o.split: fill[o.bar1].append("s_barrier")     # every wave has read P for the last time
        gaps = dma_gaps()
        assert all(b > a for a, b in zip(gaps, gaps[1:])) and gaps[-1] <= 57, gaps
        for d, g in enumerate(gaps):
            prep, issue = dma(d)
            fill[g - 1] = prep + fill[g - 1] if g - 1 == o.bar1 else fill[g - 1] + prep      # (the first m0 set-up goes ahead of the barrier)
            if not o.nodma: fill[g].append(issue)
        for q, tg in enumerate(toggles()):
            fill[tog0 + 2 * q].append(tg)
        assert tog0 + 14 < o.reads0_start
        n_before = sum(1 for g in gaps if g <= o.wait2)
        fill[o.wait2].append(f"s_waitcnt vmcnt({0 if o.nodma else n_before})")
        fill[o.bar2].append("s_barrier")
        if not nof0: put_reads(frag_reads(0), o.reads0_start, o.reads0_per_gap)
        u = gaps[-1] + 1
        fill[u].append(f"v_add_u32 %{VOA}, 0x80, %{VOA}")
        fill[u + 1].append(f"v_add_u32 %{VOB}, 0x80, %{VOB}")
        fill[u + 2].append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
        fill[u + 3].append(f"s_sub_u32 %{DELTA}, 0, %{DELTA}")
        fill[u + 4].append(f"s_sub_u32 %{CNT}, %{CNT}, 1")
        if not nof0: fill[62].append("s_waitcnt lgkmcnt(0)")
        if loop:
            fill[63].append(f"s_cmp_gt_u32 %{CNT}, 2")
            fill[63].append(f"s_cbranch_scc1 {loop}")
    elif kind == 't1':
        for q, tg in enumerate(toggles()):
            fill[tog0 + 2 * q].append(tg)
        fill[o.wait2].append("s_waitcnt vmcnt(0)")
        fill[o.bar2].append("s_barrier")
        put_reads(frag_reads(0), o.reads0_start, o.reads0_per_gap)
        fill[62].append("s_waitcnt lgkmcnt(0)")
    if timing_steady and (o.nobar or o.nosalu):              # timing builds (wrong results): what the barriers / the scalar + address instructions cost
        keep = lambda x: not ((o.nobar and x.startswith("s_barrier")) or
                              (o.nosalu and (x.startswith("s_add_u32 m0") or x.startswith("s_mov_b32") or x.startswith(f"s_add_u32 %{SO}") or x.startswith("v_add_u32"))))
        fill = [[x for x in f if keep(x)] for f in fill]
    lines = []
    for m in range(64):
        lines.append(mfma(m, zero))
        lines += fill[m]
    return lines

def prologue():
    L = []
    def tile():
        for d in range(16):
            prep, issue = dma(d)
            L.extend(prep)
            if d in (0, 8): L.append("s_nop 0")
            L.append(issue)
        L.append(f"v_add_u32 %{VOA}, 0x80, %{VOA}")
        L.append(f"v_add_u32 %{VOB}, 0x80, %{VOB}")
    tile()                                                   # K-tile 0 -> buffer 0
    L.append(f"s_cmp_lt_u32 %{CNT}, 2")
    L.append("s_cbranch_scc1 L_ow_one_%=")
    L.append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
    tile()                                                   # K-tile 1 -> buffer 1
    L.append(f"s_sub_u32 %{SM}, %{SM}, %{DELTA}")
    L.append("s_waitcnt vmcnt(16)")
    L.append("s_branch L_ow_go_%=")
    L.append("L_ow_one_%=:")
    L.append("s_waitcnt vmcnt(0)")
    L.append("L_ow_go_%=:")
    L.append("s_barrier")
    L.extend(frag_reads(0))
    if OPT.noread: L.extend(frag_reads(1))                   # (timing build: the steady loop reads nothing, keep real data in every fragment)
    L.append("s_waitcnt lgkmcnt(0)")
    return L

def program():
    L = prologue()
    L.append(f"s_cmp_le_u32 %{CNT}, 2")
    L.append("s_cbranch_scc1 L_ow_t1_%=")
    L.append("L_ow_steady_%=:")
    L += body('steady')
    L.append("L_ow_t1_%=:")
    L.append(f"s_cmp_eq_u32 %{CNT}, 1")
    L.append("s_cbranch_scc1 L_ow_t2_%=")
    L += body('t1')
    L.append("L_ow_t2_%=:")
    L += body('t2')
    L.append("s_nop 15")                                     # the compiler does not know the block ends in MFMAs: cover the XDL-write -> VALU-read wait states
    L.append("s_nop 15")
    return L

def program_p_pro():
    """persistent kernel, once per block: K-tiles 0 and 1 of the block's first tile, first-half fragments of K-tile 0 (needs >= 3 K-tiles per tile)."""
    L = []
    def tile():
        for d in range(16):
            prep, issue = dma(d)
            L.extend(prep)
            if d in (0, 8): L.append("s_nop 0")
            L.append(issue)
        L.append(f"v_add_u32 %{VOA}, 0x80, %{VOA}")
        L.append(f"v_add_u32 %{VOB}, 0x80, %{VOB}")
    tile()
    L.append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
    tile()
    L.append(f"s_sub_u32 %{SM}, %{SM}, %{DELTA}")
    L.append("s_waitcnt vmcnt(16)")
    L.append("s_barrier")
    return L

def program_p_main(has_next):
    """persistent kernel, once per tile.  On entry: the tile's K-tile 0 certified in its buffer (every wave's pieces landed, barrier passed), its K-tile 1 in
    flight, lane offsets at K-tile 2.  has_next: the K-tile stream runs on into the block's next tile - the last two K-tiles fetch the next tile's K-tiles 0 / 1
    (lane offsets VOAN / VOBN) and the last one certifies K-tile 0, so the next tile's operands travel while this tile's epilogue runs; on exit the entry state
    holds for that tile.  (The first fragments are read at the entry, not by the previous tile's last K-tile: 64 registers kept alive across the epilogue spill.)"""
    L = frag_reads(0) + ["s_waitcnt lgkmcnt(0)"]
    L += body('steady', zero=True, loop=None)
    L.append(f"s_cmp_le_u32 %{CNT}, 2")
    L.append("s_cbranch_scc1 L_owp_tail_%=")
    L.append("L_owp_loop_%=:")
    L += body('steady', loop="L_owp_loop_%=")
    L.append("L_owp_tail_%=:")
    if has_next:
        L.append(f"v_mov_b32 %{VOA}, %{VOAN}")
        L.append(f"v_mov_b32 %{VOB}, %{VOBN}")
        L += body('steady', loop=None)
        L += body('steady', loop=None, nof0=True)
    else:
        L += body('t1')
        L += body('t2')
    L.append("s_nop 15")
    L.append("s_nop 15")
    return L

def write_inc(out, L, what):
    with open(out, "w") as f:
        f.write(f"// GENERATED by tools/gen_nt_ow_loop.py ({what}) - do not edit; the schedule is documented there.\n")
        f.write("// Operands: %0-15 accumulators (AGPR) | %16-47 fragments | %48-55 fragment-read addresses | %56/57 DMA lane offsets A/B |\n")
        f.write("// %58 LDS address of the wave's first A piece in the DMA target buffer | %59 K-tiles left | %60 scratch SGPR | %61 +-64 KiB |\n")
        f.write("// %62/63 buffer resources A/B | %64/65 bytes between two pieces (16 rows) of A/B | %66/67 lane offsets of the next tile (persistent kernel)\n")
        for ln in L:
            f.write('"' + ln + '\\n\\t"\n')
    n_mfma = sum(1 for ln in L if ln.startswith("v_mfma"))
    print(f"wrote {out}: {len(L)} lines, {n_mfma} MFMAs")

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    for k, v in vars(OPT).items():
        if isinstance(v, bool): ap.add_argument("--" + k.replace("_", "-"), action="store_true", default=v)
        else: ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    a = ap.parse_args()
    for k in vars(OPT): setattr(OPT, k, getattr(a, k))
    if OPT.nosplit: OPT.split, OPT.wait1, OPT.bar1, OPT.dma_step = False, 18, 19, 2
    here = os.path.dirname(os.path.abspath(__file__))
    out = a.out or os.path.join(here, "..", "transfusion_pytorch_amd", "csrc", "gemm_nt_ow_loop.inc")
    write_inc(out, program(), "one tile per block")
    base = out[:-len("_loop.inc")] if out.endswith("_loop.inc") else out + "."
    write_inc(base + "p_pro.inc", program_p_pro(), "persistent: block prologue")
    write_inc(base + "p_next.inc", program_p_main(True), "persistent: tile with a successor")
    write_inc(base + "p_last.inc", program_p_main(False), "persistent: a block's last tile")

if __name__ == "__main__":
    main()
