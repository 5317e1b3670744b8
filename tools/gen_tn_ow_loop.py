#!/usr/bin/env python3
"""Generator of the hand-scheduled row loop of `gemm_tn_ow_kernel` (transfusion_pytorch_amd/csrc/gemm.hip): the weight-gradient product
C[N, K] += A[M, N]^T B[M, K] on the one-wave-per-SIMD plan of tools/gen_nt_ow_loop.py (4 waves x 128 x 128, 256 accumulator AGPRs, the fragments of a
whole 64-row step in registers, two 64 KiB LDS buffers fed by LDS-DMA with two steps in flight, two barriers per step).

What differs from the NT loop: both operands are contracted over their ROWS, so a fragment (32 columns x 16 rows in MFMA layout) is two
`ds_read_b64_tr_b16` (rows +0..3 / +4..7 of the lane's half of the k-step) - 64 reads per step instead of 32 - and a DMA piece is 4 rows x 256 bytes of a
128-column sub-slab ([64 rows][128 columns], 16-byte chunk index XOR ((row & 3) << 2): the layout of gemm_tn_wide_kernel).  The two halves of a fragment
are sub-registers of one MFMA operand, which inline-asm operands cannot express: the fragments live in FIXED registers v[64:191] (clobbered).

Writes transfusion_pytorch_amd/csrc/gemm_tn_ow_loop.inc.  Operands (OW_TN_OPERANDS in gemm.hip):
  %0-15 accumulators acc[i][j] (AGPR; i = block of 32 A columns = output rows, j = block of 32 B columns) | %16-19 / %20-23 fragment-read addresses of
  the A / B column blocks | %24 / %25 DMA lane offsets A / B | %26 LDS address of the wave's first A piece in the DMA target buffer | %27 steps left |
  %28 scratch SGPR | %29 +-64 KiB | %30 / %31 buffer resources | %32 / %33 bytes between two pieces (4 rows) | %34 / %35 bytes between two steps (64 rows)
gemm_tn_ow_sum01.inc / _sum23.inc (the waves of the first K columns also form the bias gradient, A^T x ones: the wk = 0 wave for its A blocks 0, 1, the wk = 1 wave -
which holds the same A fragments - for blocks 2, 3: 8 more MFMAs per step each) have two more outputs, the column-sum accumulators (VGPR) %30-31, which moves the
inputs to %32-37, and one more input, the ones operand %38.
"""
import argparse
import os

def ACC(i, j): return i * 4 + j
def RA(i): return 16 + i
def RB(j): return 20 + j
VOA, VOB, SM, CNT, SO, DELTA, RSA, RSB, STA, STB, KSA, KSB = 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35
SUMS, ONES = 36, 40
SUM = False
SUM_BLOCKS = (0, 1, 2, 3)          # the A blocks whose column sums this loop form accumulates (gemm_tn_ow_sum01.inc: 0, 1 - the wk = 0 wave; sum23: 2, 3 - the wk = 1 wave)
F0 = 64                                # first fragment register
def FA(i, ks): return F0 + (ks >> 1) * 64 + (ks & 1) * 16 + i * 4          # v[FA : FA + 3]; halves of the step: ks 0,1 -> v64-127, ks 2,3 -> v128-191
def FB(j, ks): return F0 + (ks >> 1) * 64 + 32 + (ks & 1) * 16 + j * 4
def vr(a, n): return f"v[{a}:{a + n - 1}]"

# defaults: the split form of the NT loop (A sub-slabs released and re-filled before the B fragments are read; DMA pieces three gaps apart): +3 ... 4.6 % on the larger
# products, neutral on the small ones (gpurun_out/ow22.txt); `--split 0 --wait1 18 --bar1 19 --dma-step 2` = one barrier behind all second-half reads
OPT = argparse.Namespace(wait1=10, bar1=11, dma_start=20, dma_step=3, n_before=12, late_start=46, late_step=3, wait2=43, bar2=44, reads0_start=45, split=1, wait1b=22, bar1b=23)

def mfma(m, zero=False):
    half, r = divmod(m, 32)
    ksl, r = divmod(r, 16)
    i, j = divmod(r, 4)
    ks = half * 2 + ksl
    a = ACC(i, j)
    c = "0" if (zero and ks == 0) else f"%{a}"
    return f"v_mfma_f32_32x32x16_bf16 %{a}, {vr(FA(i, ks), 4)}, {vr(FB(j, ks), 4)}, {c}"

def frag_reads(half):
    """32 reads of one half step (k-steps 2 half, 2 half + 1), in order of first use (A0, B0..B3, A1..A3 of each k-step)."""
    out = []
    def rd(reg, addr, ks):
        off = (ks & 3) * 4096                                  # 16 rows of 256 bytes
        out.append(f"ds_read_b64_tr_b16 {vr(reg, 2)}, %{addr}" + (f" offset:{off}" if off else ""))
        out.append(f"ds_read_b64_tr_b16 {vr(reg + 2, 2)}, %{addr} offset:{off + 1024}")
    for ksl in range(2):
        ks = half * 2 + ksl
        rd(FA(0, ks), RA(0), ks)
        for j in range(4): rd(FB(j, ks), RB(j), ks)
        for i in range(1, 4): rd(FA(i, ks), RA(i), ks)
    return out

def dma(d):
    op, jp = divmod(d, 8)
    vo, rs, st = (VOA, RSA, STA) if op == 0 else (VOB, RSB, STB)
    prep = []
    if jp == 0:
        prep.append(f"s_mov_b32 m0, %{SM}" if op == 0 else f"s_add_u32 m0, %{SM}, 0x8000")
        so = "0"
    else:
        prep.append("s_add_u32 m0, m0, 0x400")
        prep.append(f"s_mov_b32 %{SO}, %{st}" if jp == 1 else f"s_add_u32 %{SO}, %{SO}, %{st}")
        so = f"%{SO}"
    return prep, f"buffer_load_dwordx4 %{vo}, %{rs}, {so} offen lds"

def toggles():
    return [f"v_add_u32 %{RA(i)}, %{DELTA}, %{RA(i)}" for i in range(4)] + [f"v_add_u32 %{RB(j)}, %{DELTA}, %{RB(j)}" for j in range(4)]

def body(kind, zero=False, loop=None):
    o = OPT
    fill = [[] for _ in range(64)]
    def put_reads(reads, g0):
        for q, r in enumerate(reads): fill[g0 + q // 2].append(r)
    r1 = frag_reads(1)
    split = o.split and kind == 'steady'
    if split:            # (the NT loop's split form: the A sub-slabs are released - and re-filled - before the B fragments are read)
        ra = [r for r in r1 if any(f"%{RA(i)}" in r.split(",")[1] for i in range(4))]
        rb = [r for r in r1 if r not in ra]
        assert len(ra) == 16 and len(rb) == 16
        put_reads(ra, 0)
        fill[o.wait1].append("s_waitcnt lgkmcnt(0)")
        fill[o.bar1].append("s_barrier")
        put_reads(rb, o.bar1 + 1)
        fill[o.wait1b].append("s_waitcnt lgkmcnt(0)")
        fill[o.bar1b].append("s_barrier")
    else:
        put_reads(r1, 0)
        fill[18 if o.split else o.wait1].append("s_waitcnt lgkmcnt(0)")
    tog0 = (o.bar1b if o.split else o.bar1) + 2
    if kind == 'steady':
        if not split: fill[o.bar1].append("s_barrier")
        if split:
            g, gaps = o.bar1 + 1, []
            for d in range(16):
                if d == 8: g = max(g, o.bar1b + 1)
                gaps.append(g); g += o.dma_step
        else:
            gaps = [o.dma_start + o.dma_step * d if d < o.n_before else o.late_start + o.late_step * (d - o.n_before) for d in range(16)]
        assert all(b > a for a, b in zip(gaps, gaps[1:])) and gaps[-1] <= 57, gaps
        for d, g in enumerate(gaps):
            prep, issue = dma(d)
            fill[g - 1] = prep + fill[g - 1] if g - 1 == o.bar1 else fill[g - 1] + prep
            fill[g].append(issue)
        for q, tg in enumerate(toggles()):
            fill[tog0 + 2 * q].append(tg)
        assert tog0 + 14 < o.reads0_start
        n_before = sum(1 for g in gaps if g <= o.wait2)
        fill[o.wait2].append(f"s_waitcnt vmcnt({n_before})")
        fill[o.bar2].append("s_barrier")
        put_reads(frag_reads(0), o.reads0_start)
        u = gaps[-1] + 1
        fill[u].append(f"v_add_u32 %{VOA}, %{KSA}, %{VOA}")
        fill[u + 1].append(f"v_add_u32 %{VOB}, %{KSB}, %{VOB}")
        fill[u + 2].append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
        fill[u + 3].append(f"s_sub_u32 %{DELTA}, 0, %{DELTA}")
        fill[u + 4].append(f"s_sub_u32 %{CNT}, %{CNT}, 1")
        fill[62].append("s_waitcnt lgkmcnt(0)")
        if loop:
            fill[63].append(f"s_cmp_gt_u32 %{CNT}, 2")
            fill[63].append(f"s_cbranch_scc1 {loop}")
    elif kind == 't1':
        for q, tg in enumerate(toggles()):
            fill[tog0 + 2 * q].append(tg)
        fill[o.wait2].append("s_waitcnt vmcnt(0)")
        fill[o.bar2].append("s_barrier")
        put_reads(frag_reads(0), o.reads0_start)
        fill[62].append("s_waitcnt lgkmcnt(0)")
    lines = []
    for m in range(64):
        lines.append(mfma(m, zero))
        if SUM and m % 4 == 3:                                   # behind the four MFMAs of (k-step, A block i): the block's column sums (AHEAD of the gap's fillers: the last gap holds the loop branch)
            half, r = divmod(m, 32); ksl, r = divmod(r, 16); i = r // 4; ks = half * 2 + ksl
            if i in SUM_BLOCKS:
                acc = SUMS + SUM_BLOCKS.index(i)
                c = "0" if (zero and ks == 0) else f"%{acc}"
                lines.append(f"v_mfma_f32_32x32x16_bf16 %{acc}, {vr(FA(i, ks), 4)}, %{ONES}, {c}")
        lines += fill[m]
    return lines

def program():
    """>= 3 steps: prologue (steps 0 and 1 fetched, first fragments read), first step (accumulators from the inline zero), steady loop, two tail steps."""
    L = []
    def tile():
        for d in range(16):
            prep, issue = dma(d)
            L.extend(prep)
            if d in (0, 8): L.append("s_nop 0")
            L.append(issue)
        L.append(f"v_add_u32 %{VOA}, %{KSA}, %{VOA}")
        L.append(f"v_add_u32 %{VOB}, %{KSB}, %{VOB}")
    tile()
    L.append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
    tile()
    L.append(f"s_sub_u32 %{SM}, %{SM}, %{DELTA}")
    L.append("s_waitcnt vmcnt(16)")
    L.append("s_barrier")
    L.extend(frag_reads(0))
    L.append("s_waitcnt lgkmcnt(0)")
    L += body('steady', zero=True)
    L.append(f"s_cmp_le_u32 %{CNT}, 2")
    L.append("s_cbranch_scc1 L_tnow_tail_%=")
    L.append("L_tnow_loop_%=:")
    L += body('steady', loop="L_tnow_loop_%=")
    L.append("L_tnow_tail_%=:")
    L += body('t1')
    L += body('t2')
    L.append("s_nop 15")
    L.append("s_nop 15")
    return L

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    for k, v in vars(OPT).items(): ap.add_argument("--" + k.replace("_", "-"), type=int, default=v)
    a = ap.parse_args()
    for k in vars(OPT): setattr(OPT, k, getattr(a, k))
    here = os.path.dirname(os.path.abspath(__file__))
    out = a.out or os.path.join(here, "..", "transfusion_pytorch_amd", "csrc", "gemm_tn_ow_loop.inc")
    global SUM, RSA, RSB, STA, STB, KSA, KSB, SUMS, ONES, SUM_BLOCKS
    # the bias gradient is shared by the two waves that hold the same A fragments (wk = 0 sums blocks 0, 1; wk = 1 blocks 2, 3): 8 more MFMAs per step on each
    # instead of 16 on one of them (two column-sum accumulators %30, %31; inputs %32-37, ones %38)
    for SUM, SUM_BLOCKS, path in ((False, (), out), (True, (0, 1), out.replace("_loop.inc", "_sum01.inc")), (True, (2, 3), out.replace("_loop.inc", "_sum23.inc"))):
        if SUM: SUMS, RSA, RSB, STA, STB, KSA, KSB, ONES = 30, 32, 33, 34, 35, 36, 37, 38
        L = program()
        with open(path, "w") as f:
            f.write("// GENERATED by tools/gen_tn_ow_loop.py - do not edit; operands and schedule are documented there.  Fragments: fixed registers v[64:191].\n")
            for ln in L:
                f.write('"' + ln + '\\n\\t"\n')
        print(f"wrote {path}: {len(L)} lines, {sum(1 for x in L if x.startswith('v_mfma'))} MFMAs, {sum(1 for x in L if x.startswith('ds_read'))} reads")

if __name__ == "__main__":
    main()
