#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of `gemm_nt_ow_kernel` (transfusion_pytorch_amd/csrc/gemm.hip).

Writes transfusion_pytorch_amd/csrc/gemm_nt_ow_loop.inc: ONE inline-asm statement (prologue, steady-state K-tile, the two tail K-tiles)
whose register operands are bound by the C++ side (`OW_ASM_OPERANDS` in gemm.hip, same numbering as below).

Why a generator: the loop is 64 MFMAs per K-tile with every LDS read, LDS-DMA issue, scalar update, counted wait and barrier placed by hand
in the 32-clock gaps between the MFMAs of a lone wave (one wave per SIMD, 128 x 128 per wave, 256 accumulator AGPRs): hipcc's scheduler
re-serialises such a loop (LAB_NOTEBOOK round 4: two hipcc-scheduled kernels of this shape landed on the ping-pong kernel's speed).

Schedule of one K-tile (gap g = the slot behind MFMA g; P = the LDS buffer of K-tile kt, Q = the other one):
   g 0..15   read the fragments of the K-tile's second half (k-steps 2, 3) from P            -> every read of P is issued
   g 18/19   lgkmcnt(0) ; s_barrier                                                          -> nobody reads P any more
   g 20..42  12 LDS-DMA pieces of K-tile kt+2 into P, one per two gaps (A pieces first: they come from HBM, B = weights from L2)
   g 21..35  (odd) the 8 fragment-read addresses move from P to Q
   g 43/44   vmcnt(12) ; s_barrier                                                           -> K-tile kt+1 has landed in Q, for every wave
   g 45..60  read the fragments of K-tile kt+1's first half (k-steps 0, 1) from Q ; the last 4 DMA pieces at g 46 / 49 / 52 / 55
   g 56..60  lane offsets, DMA target, K-tile counter
   g 62      lgkmcnt(0)
Fragments of a whole K-tile live in registers (2 x 16 x 4 VGPRs), which is what frees P for the DMA of K-tile kt+2 a quarter into K-tile kt:
two K-tiles stay in flight with two 64 KiB LDS buffers.
"""
import os

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

def mfma(m):
    half, r = divmod(m, 32)
    ksl, r = divmod(r, 16)
    j, i = divmod(r, 4)
    ks = half * 2 + ksl
    a = ACC(i, j)
    return f"v_mfma_f32_32x32x16_bf16 %{a}, %{FB(j, ks)}, %{FA(i, ks)}, %{a}"

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

def body(kind):
    """kind: 'steady' (DMA of K-tile kt+2, reads of kt+1), 't1' (K-tile nk-2: no DMA), 't2' (last K-tile: second-half reads only)."""
    fill = [[] for _ in range(64)]
    for g, r in enumerate(frag_reads(1)):
        fill[g].append(r)
    fill[18].append("s_waitcnt lgkmcnt(0)")                 # second-half fragments landed (needed from MFMA 32 on)
    if kind == 'steady':
        fill[19].append("s_barrier")                        # every wave has read P for the last time
        for d in range(16):
            g = 20 + 2 * d if d < 12 else 46 + 3 * (d - 12)
            prep, issue = dma(d)
            fill[g - 1] = prep + fill[g - 1] if g - 1 == 19 else fill[g - 1] + prep      # (g 19: the m0 set-up ahead of the barrier)
            fill[g].append(issue)
        for q, tg in enumerate(toggles()):
            fill[21 + 2 * q].append(tg)
        fill[43].append("s_waitcnt vmcnt(12)")
        fill[44].append("s_barrier")
        for q, r in enumerate(frag_reads(0)):
            fill[45 + q].append(r)
        fill[56].append(f"v_add_u32 %{VOA}, 0x80, %{VOA}")
        fill[57].append(f"v_add_u32 %{VOB}, 0x80, %{VOB}")
        fill[58].append(f"s_add_u32 %{SM}, %{SM}, %{DELTA}")
        fill[59].append(f"s_sub_u32 %{DELTA}, 0, %{DELTA}")
        fill[60].append(f"s_sub_u32 %{CNT}, %{CNT}, 1")
        fill[62].append("s_waitcnt lgkmcnt(0)")
        fill[63].append(f"s_cmp_gt_u32 %{CNT}, 2")
        fill[63].append("s_cbranch_scc1 L_ow_steady_%=")
    elif kind == 't1':
        for q, tg in enumerate(toggles()):
            fill[21 + 2 * q].append(tg)
        fill[43].append("s_waitcnt vmcnt(0)")
        fill[44].append("s_barrier")
        for q, r in enumerate(frag_reads(0)):
            fill[45 + q].append(r)
        fill[62].append("s_waitcnt lgkmcnt(0)")
    lines = []
    for m in range(64):
        lines.append(mfma(m))
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

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "transfusion_pytorch_amd", "csrc", "gemm_nt_ow_loop.inc")
    L = program()
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_nt_ow_loop.py - do not edit; the schedule is documented there.\n")
        f.write("// Operands: %0-15 accumulators (AGPR) | %16-47 fragments | %48-55 fragment-read addresses | %56/57 DMA lane offsets A/B |\n")
        f.write("// %58 LDS address of the wave's first A piece in the DMA target buffer | %59 K-tiles left | %60 scratch SGPR | %61 +-64 KiB |\n")
        f.write("// %62/63 buffer resources A/B | %64/65 bytes between two pieces (16 rows) of A/B\n")
        for ln in L:
            f.write('"' + ln + '\\n\\t"\n')
    n_mfma = sum(1 for ln in L if ln.startswith("v_mfma"))
    print(f"wrote {out}: {len(L)} lines, {n_mfma} MFMAs")

if __name__ == "__main__":
    main()
