"""Per-kernel parity: every HIP kernel (through the C ABI) vs a plain PyTorch fp32 reference of the same op.
Tolerances: bf16 outputs rel-Frobenius <= 1e-2 (bf16 eps = 3.9e-3); fp32 outputs of bf16 MFMA GEMMs <= 5e-3."""
import ctypes
import math

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from transfusion_pytorch_amd import capi  # noqa: E402

DEV = 'cuda'
BF = torch.bfloat16


def stream():
    return torch.cuda.current_stream().cuda_stream


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def check(name, got, ref, tol):
    assert torch.isfinite(got.float()).all(), f'{name}: non-finite output'
    e = relerr(got, ref)
    print(f'{name}: rel err {e:.3e} (tol {tol})')
    assert e <= tol, f'{name}: rel err {e} > {tol}'


def rnd(*shape, scale=1.0, dtype=BF):
    return (torch.randn(*shape, device=DEV) * scale).to(dtype)


def gemm_nt(**kw):
    a = capi.make_args('tfx_gemm_nt_args', **kw)
    capi.call('tfx_gemm_nt', a, stream())


# ---------------------------------------------------------------------------------------------- GEMM NT
# (33000, 1032, 128) and (65536, 512, 192): >= 512 tiles of 256x256 -> the ping-pong kernel (ragged M and N tiles in the first)
@pytest.mark.parametrize('M,N,K', [(300, 200, 128), (128, 128, 64), (1000, 1544, 512), (4096, 512, 1408), (77, 390, 192), (33000, 1032, 128), (65536, 512, 192),
                                   (64, 1544, 512), (1, 512, 64), (37, 2816, 1408), (512, 520, 1024), (256, 128, 768),    # M <= 512: the skinny deep-ring kernel
                                   (640, 5504, 1024), (2000, 388, 512), (640, 1024, 2752), (640, 1544, 1024), (3000, 1024, 256),
                                   # the training step's own shapes at bench size (b 64 x 1024 tokens, dim512): qkvg / out / ff2 forward, dX of ff1 / qkvg
                                   (65536, 1544, 512), (65536, 512, 512), (65536, 512, 1408), (65536, 512, 2816), (65536, 512, 1600)])   # <= 256 tiles of 128 x 128: the 4-slot mid kernel; 640 rows: split-K decode kernel up to 1024 rows
def test_gemm_nt_bf16_bias(M, N, K):
    torch.manual_seed(0)
    A, B = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    Np = (N + 3) // 4 * 4
    bias = torch.randn(Np, device=DEV)
    C = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    gemm_nt(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=N, bias=bias)
    ref = A.float() @ B.float().T + bias[:N]
    check(f'gemm_nt bf16 {M}x{N}x{K}', C, ref, 6e-3)
    C32 = torch.full((M, N), float('nan'), device=DEV)
    gemm_nt(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_F32'], C=C32, ldc=N, bias=bias)
    check(f'gemm_nt f32 {M}x{N}x{K}', C32, ref, 1e-5 * 50)


@pytest.mark.parametrize('M,N,K', [(64, 1024, 1024), (300, 1544, 512), (640, 5504, 1024), (4096, 512, 384), (37, 2816, 1408)])   # decode / skinny / mid / 128 x 128 kernels
def test_gemm_nt_prefetch_of_the_next_weights_changes_nothing(M, N, K):
    """round 6 (tfx_gemm_nt_args.prefetch): spare blocks of a small-M launch touch `prefetch_bytes` of another buffer (the next GEMM's weights).  They
    must leave the launch's own tiles alone (its block -> tile map ignores them), write nothing, and stay inside the span - ragged byte counts, a span shorter
    than one block's share, and a span that needs the 512-block cap."""
    torch.manual_seed(0)
    A, B = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn((N + 3) // 4 * 4, device=DEV)
    kw = dict(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], ldc=N, bias=bias)
    C0 = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
    gemm_nt(C=C0, **kw)
    for nbytes in (100, 65536, 3 * 65536 + 4, 40 * 2 ** 20):
        W = torch.full((nbytes // 2 + 64,), 1.0, device=DEV, dtype=BF)          # the guard elements behind the span are read back below
        C1 = torch.full((M, N), float('nan'), device=DEV, dtype=BF)
        gemm_nt(C=C1, prefetch=W, prefetch_bytes=nbytes, **kw)
        torch.cuda.synchronize()
        assert torch.equal(C0.view(torch.int16), C1.view(torch.int16)), (M, N, K, nbytes)
        assert bool((W == 1.0).all())


def test_gemm_nt_one_wave_kernels_bit_identical_to_ping_pong():
    """round 5: the one-wave-per-SIMD NT kernels (gemm_nt_ow_kernel: one tile per block; gemm_nt_owp_kernel: persistent, K-tile stream across tile boundaries;
    hand-scheduled inline-asm K loops from tools/gen_nt_ow_loop.py) keep the ping-pong kernel's LDS layout, per-accumulator k order and epilogues: the same bits.
    The library reads TFX_NT_OW once per process, so the cases run in two child processes (0 = ping-pong, 1 = the default: bf16 outputs and long-K fp32 outputs on
    the new kernels; TFX_NT_PP_MIN=1 sends the few-tile shapes to the 256 x 256 family) and the hashes of every output must agree."""
    import subprocess, sys
    outs = []
    for mode in ('0', '1'):
        env = dict(os.environ, TFX_NT_OW=mode, TFX_NT_PP_MIN='1')
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), '_nt_hash_child.py')], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith('CASE')])
    assert len(outs[0]) == 13 and len(outs[0]) == len(outs[1])
    for a, b in zip(*outs):
        print(a)
        assert a == b, f'ping-pong: {a}\none-wave:  {b}'


def test_gemm_nt_asymmetric_identity():
    # transpose-detecting check: A = I, asymmetric B
    M = N = K = 128
    A = torch.eye(M, device=DEV).to(BF)
    B = (torch.arange(N * K, device=DEV).reshape(N, K) % 251).float().to(BF)
    C = torch.zeros(M, N, device=DEV)
    gemm_nt(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_F32'], C=C, ldc=N)
    assert torch.equal(C, B.float().T)


@pytest.mark.parametrize('M', [333, 1500, 66000])      # 66000 rows x 512 columns: 516 tiles of 256x256 -> ping-pong kernel (split-A, RESID); 1500: the mid kernel
def test_gemm_nt_split_a_rowmaps_resid(M):
    torch.manual_seed(1)
    N, K1, K2 = (256 if M < 1000 else 512), 128, 192
    A1, A2 = rnd(M, K1), rnd(M, K2)
    B = rnd(N, K1 + K2, scale=(K1 + K2) ** -0.5)
    R = rnd(M, N)
    C = torch.zeros(M, N, device=DEV, dtype=BF)
    gemm_nt(A=A1, lda=K1, A2=A2, lda2=K2, K1=K1, B=B, ldb=K1 + K2, M=M, N=N, K=K1 + K2,
            epi=capi.ENUMS['TFX_EPI_RESID'], C=C, ldc=N, R=R, ldr=N)
    ref = torch.cat([A1, A2], 1).float() @ B.float().T + R.float()
    check('gemm_nt split-A + resid', C, ref, 6e-3)
    # gather A rows + scatter C rows (negative = drop) + residual read at the scattered row
    Msrc, T = 500, M + 367
    Asrc = rnd(Msrc, K1)
    amap = torch.randint(0, Msrc, (M,), device=DEV, dtype=torch.int32)
    omap = torch.randperm(T, device=DEV)[:M].to(torch.int32)
    omap[::17] = -1
    Bs = rnd(N, K1, scale=K1 ** -0.5)
    Cbig = rnd(T, N)
    Cref = Cbig.float().clone()
    gemm_nt(A=Asrc, lda=K1, a_rowmap=amap, B=Bs, ldb=K1, M=M, N=N, K=K1, epi=capi.ENUMS['TFX_EPI_RESID'],
            C=Cbig, ldc=N, R=Cbig, ldr=N, resid_mapped=1, rowmap=omap)
    prod = Asrc[amap.long()].float() @ Bs.float().T
    keep = omap >= 0
    Cref[omap[keep].long()] += prod[keep]
    check('gemm_nt gather/scatter/resid_mapped', Cbig, Cref, 6e-3)


def test_gemm_nt_silu():
    torch.manual_seed(2)
    M, N, K = 200, 512, 576
    A, B = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn(N, device=DEV)
    C = torch.zeros(M, N, device=DEV, dtype=BF); C2 = torch.zeros(M, N, device=DEV, dtype=BF)
    gemm_nt(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_SILU'], C=C, ldc=N, C2=C2, ldc2=N, bias=bias)
    pre = A.float() @ B.float().T + bias
    check('gemm_nt silu pre', C2, pre, 6e-3)
    check('gemm_nt silu act', C, F.silu(pre), 6e-3)


def gelu_grad(g):
    """gelu'(g) = Phi(g) + g phi(g) (erf form, F.gelu's derivative)"""
    return 0.5 * (1 + torch.erf(g * 2 ** -0.5)) + g * torch.exp(-0.5 * g * g) * (2 * torch.pi) ** -0.5


def geglu_saved_ref(a, g, is_gate, feat):
    """what the GEGLU forward saves for its backward (round 5; csrc/tfx_common.h geglu_uvh): u = gelu(g) in the value slots, v = a gelu'(g) in the gate
    slots of the interleaved buffer"""
    return torch.where(is_gate[None], (a * gelu_grad(g))[:, feat], F.gelu(g)[:, feat])


def geglu_bwd_refs(dh, ag, a, g, is_gate):
    """(d[a|g] from the SAVED bf16 [u|v]: da = dh u, dg = dh v - the epilogue's own arithmetic;  d[a|g] by autograd from the fp32 pre-activations)"""
    from_saved = torch.zeros_like(ag, dtype=torch.float32)
    from_saved[:, ~is_gate] = dh * ag.float()[:, ~is_gate]; from_saved[:, is_gate] = dh * ag.float()[:, is_gate]
    a_, g_ = a.detach().clone().requires_grad_(True), g.detach().clone().requires_grad_(True)
    (a_ * F.gelu(g_)).backward(dh)
    auto = torch.zeros_like(from_saved)
    auto[:, ~is_gate] = a_.grad; auto[:, is_gate] = g_.grad
    return from_saved, auto


def geglu_perm(dip):
    """physical column c of the interleaved layout -> (is_gate, feature)."""
    c = torch.arange(2 * dip)
    blk, within = c // 64, c % 64
    is_gate = within >= 32
    feat = blk * 32 + within % 32
    return is_gate, feat


@pytest.mark.parametrize('M', [300, 130, 640, 70000])  # 70000 rows: the ping-pong 256x256 kernel and its staged epilogues; 130: the split-K decode kernel (K >= 256); 640: the mid kernel
def test_gemm_nt_geglu_fwd_bwd(M):
    torch.manual_seed(3)
    d, dip = (512 if M in (130, 640) else 128), (192 if M < 600 else 2752 if M == 640 else 1024)
    u = rnd(M, d)
    Wa, Wg = rnd(dip, d, scale=d ** -0.5), rnd(dip, d, scale=d ** -0.5)
    ba, bg = torch.randn(dip, device=DEV), torch.randn(dip, device=DEV)
    is_gate, feat = geglu_perm(dip)
    is_gate, feat = is_gate.to(DEV), feat.to(DEV)
    Wphys = torch.where(is_gate[:, None], Wg[feat], Wa[feat]).contiguous()
    bphys = torch.where(is_gate, bg[feat], ba[feat]).contiguous()
    ag = torch.zeros(M, 2 * dip, device=DEV, dtype=BF); hm = torch.zeros(M, dip, device=DEV, dtype=BF)
    gemm_nt(A=u, lda=d, B=Wphys, ldb=d, M=M, N=2 * dip, K=d, epi=capi.ENUMS['TFX_EPI_GEGLU'], C=ag, ldc=2 * dip,
            C2=hm, ldc2=dip, bias=bphys)
    a = u.float() @ Wa.float().T + ba
    g = u.float() @ Wg.float().T + bg
    check('geglu saved [gelu(g) | a gelu\'(g)] (interleaved)', ag, geglu_saved_ref(a, g, is_gate, feat), 6e-3)
    check('geglu hidden', hm, a * F.gelu(g), 8e-3)
    # backward epilogue: dh = dy @ W2t^T (here: plain GEMM against random B), d[a|g] from saved ag
    K2 = 320 if M in (130, 640) else 128
    dy, W2t = rnd(M, K2), rnd(dip, K2, scale=K2 ** -0.5)
    dag = torch.zeros(M, 2 * dip, device=DEV, dtype=BF)
    gemm_nt(A=dy, lda=K2, B=W2t, ldb=K2, M=M, N=dip, K=K2, epi=capi.ENUMS['TFX_EPI_GEGLU_BWD'], C=dag, ldc=2 * dip,
            aux=ag, ldaux=2 * dip)
    dh = dy.float() @ W2t.float().T
    from_saved, auto = geglu_bwd_refs(dh, ag, a, g, is_gate)
    check('geglu backward epilogue (from the saved [u|v])', dag, from_saved, 6e-3)
    check('geglu backward epilogue (autograd of a gelu(g), fp32 pre-activations)', dag, auto, 1e-2)


def test_gemm_nt_bench_shapes_resid_skip_geglu():
    """the fused epilogues of the ping-pong kernel at the bench's own shapes (M = 65536 tokens, dim 512): attention out-projection
    + residual (RESID, K = 512), U-Net skip projection (split-A RESID, K = 2 x 512), GEGLU forward (N = 2 x 1408 interleaved, bias) and GEGLU backward
    (N = 1408, K = 512, saved [a|g]) - against fp32 torch.matmul on the same bf16 operands."""
    torch.manual_seed(11)
    M, d, dip = 65536, 512, 1408
    A, B, R = rnd(M, d), rnd(d, d, scale=d ** -0.5), rnd(M, d)
    C = torch.full((M, d), float('nan'), device=DEV, dtype=BF)
    gemm_nt(A=A, lda=d, B=B, ldb=d, M=M, N=d, K=d, epi=capi.ENUMS['TFX_EPI_RESID'], C=C, ldc=d, R=R, ldr=d)
    check('bench out-proj + residual', C, A.float() @ B.float().T + R.float(), 6e-3)
    A2, B2 = rnd(M, d), rnd(d, 2 * d, scale=(2 * d) ** -0.5)
    C.fill_(float('nan'))
    gemm_nt(A=A, lda=d, A2=A2, lda2=d, K1=d, B=B2, ldb=2 * d, M=M, N=d, K=2 * d, epi=capi.ENUMS['TFX_EPI_RESID'], C=C, ldc=d, R=A, ldr=d)
    check('bench skip projection (split-A) + residual', C, torch.cat([A, A2], 1).float() @ B2.float().T + A.float(), 6e-3)
    del A2, B2, R
    Wa, Wg = rnd(dip, d, scale=d ** -0.5), rnd(dip, d, scale=d ** -0.5)
    ba, bg = torch.randn(dip, device=DEV), torch.randn(dip, device=DEV)
    is_gate, feat = geglu_perm(dip)
    is_gate, feat = is_gate.to(DEV), feat.to(DEV)
    Wphys = torch.where(is_gate[:, None], Wg[feat], Wa[feat]).contiguous()
    bphys = torch.where(is_gate, bg[feat], ba[feat]).contiguous()
    ag = torch.full((M, 2 * dip), float('nan'), device=DEV, dtype=BF); hm = torch.full((M, dip), float('nan'), device=DEV, dtype=BF)
    gemm_nt(A=A, lda=d, B=Wphys, ldb=d, M=M, N=2 * dip, K=d, epi=capi.ENUMS['TFX_EPI_GEGLU'], C=ag, ldc=2 * dip, C2=hm, ldc2=dip, bias=bphys)
    a = A.float() @ Wa.float().T + ba
    g = A.float() @ Wg.float().T + bg
    check('bench geglu saved [u|v]', ag, geglu_saved_ref(a, g, is_gate, feat), 6e-3)
    check('bench geglu hidden', hm, a * F.gelu(g), 8e-3)
    dy, W2t = rnd(M, d), rnd(dip, d, scale=d ** -0.5)
    dag = torch.full((M, 2 * dip), float('nan'), device=DEV, dtype=BF)
    gemm_nt(A=dy, lda=d, B=W2t, ldb=d, M=M, N=dip, K=d, epi=capi.ENUMS['TFX_EPI_GEGLU_BWD'], C=dag, ldc=2 * dip, aux=ag, ldaux=2 * dip)
    dh = dy.float() @ W2t.float().T
    from_saved, auto = geglu_bwd_refs(dh, ag, a, g, is_gate)
    del a, g
    check('bench geglu backward epilogue (from the saved [u|v])', dag, from_saved, 6e-3)
    check('bench geglu backward epilogue (autograd, fp32 pre-activations)', dag, auto, 1e-2)


def test_geglu_table_range_ends():
    """The ping-pong kernel's GEGLU forward reads gelu AND its derivative from a 32 KiB LDS grid (second-order Taylor on a 2^-7 grid over [-8, 8), linear
    extrapolation beyond: csrc/gemm.hip geglu_uvh_grid) and saves u = gelu(g), v = a gelu'(g) for the backward, whose epilogue is da = dh u, dg = dh v
    (round 5; ADVICE r4: the round-4 backward table clamped |g| >= 8 and the test ran on a shape that never took the table kernel).  Gate pre-activations
    pinned (through the bias) far outside, at the ends of and deep inside the grid - saturated, clamped and tiny - against fp32 torch, per pinned column,
    on a shape that DOES run the ping-pong kernel (asserted through tfx_gemm_nt_plan)."""
    import ctypes
    torch.manual_seed(12)
    M, d, dip = 65536 + 200, 128, 1408
    u = rnd(M, d)
    Wa, Wg = rnd(dip, d, scale=d ** -0.5), rnd(dip, d, scale=1e-3 * d ** -0.5)
    ba = torch.randn(dip, device=DEV)
    pins = torch.tensor([-40., -9., -8.01, -7.99, -3., -1e-2, -2e-4, -1e-5, 0., 3e-6, 1.3e-4, 6e-3, 0.5, 2.5, 7.97, 8.03, 11., 60.], device=DEV)
    bg = pins[torch.arange(dip, device=DEV) % pins.numel()].contiguous()
    is_gate, feat = geglu_perm(dip)
    is_gate, feat = is_gate.to(DEV), feat.to(DEV)
    Wphys = torch.where(is_gate[:, None], Wg[feat], Wa[feat]).contiguous()
    bphys = torch.where(is_gate, bg[feat], ba[feat]).contiguous()
    ag = torch.zeros(M, 2 * dip, device=DEV, dtype=BF); hm = torch.zeros(M, dip, device=DEV, dtype=BF)
    kw = dict(A=u, lda=d, B=Wphys, ldb=d, M=M, N=2 * dip, K=d, epi=capi.ENUMS['TFX_EPI_GEGLU'], C=ag, ldc=2 * dip, C2=hm, ldc2=dip, bias=bphys)
    kind, grid = ctypes.c_int32(-1), ctypes.c_int32(-1)
    capi.lib().tfx_gemm_nt_plan(ctypes.byref(capi.make_args('tfx_gemm_nt_args', **kw)), ctypes.byref(kind), ctypes.byref(grid))
    assert kind.value == 3, 'the shape must run on the ping-pong kernel (the one with the LDS grid)'
    gemm_nt(**kw)
    a = u.float() @ Wa.float().T + ba
    g = u.float() @ Wg.float().T + bg
    saved = torch.zeros(M, 2 * dip, device=DEV)
    saved[:, ~is_gate] = F.gelu(g); saved[:, is_gate] = a * gelu_grad(g)
    for nm, got, ref in (('h', hm.float(), a * F.gelu(g)), ('saved u = gelu(g)', ag.float()[:, ~is_gate], F.gelu(g)), ('saved v = a gelu\'(g)', ag.float()[:, is_gate], a * gelu_grad(g))):
        for c in range(pins.numel()):                            # per pinned column: elementwise, relative to the column's own scale
            cols = torch.arange(c, dip, pins.numel(), device=DEV)
            e = (got[:, cols] - ref[:, cols]).abs().max().item()
            sc = ref[:, cols].abs().max().item()
            assert e <= 8e-3 * sc + 1e-7, f'forward {nm}, gate ~ {pins[c].item():g}: max err {e:.3e} at scale {sc:.3e}'
    del saved
    dy, W2t = rnd(M, d), rnd(dip, d, scale=d ** -0.5)
    dag = torch.zeros(M, 2 * dip, device=DEV, dtype=BF)
    kwb = dict(A=dy, lda=d, B=W2t, ldb=d, M=M, N=dip, K=d, epi=capi.ENUMS['TFX_EPI_GEGLU_BWD'], C=dag, ldc=2 * dip, aux=ag, ldaux=2 * dip)
    capi.lib().tfx_gemm_nt_plan(ctypes.byref(capi.make_args('tfx_gemm_nt_args', **kwb)), ctypes.byref(kind), ctypes.byref(grid))
    assert kind.value == 3
    gemm_nt(**kwb)
    dh = dy.float() @ W2t.float().T
    _, auto = geglu_bwd_refs(dh, ag, a, g, is_gate)
    da, dg = dag.float()[:, ~is_gate], dag.float()[:, is_gate]
    for c in range(pins.numel()):
        cols = torch.arange(c, dip, pins.numel(), device=DEV)
        for nm, got, ref in (('da', da, auto[:, ~is_gate]), ('dg', dg, auto[:, is_gate])):
            e = (got[:, cols] - ref[:, cols]).abs().max().item()
            sc = ref[:, cols].abs().max().item()
            assert e <= 1.2e-2 * sc + 1e-7, f'backward {nm}, gate ~ {pins[c].item():g}: max err {e:.3e} at scale {sc:.3e}'   # two bf16 roundings (saved value, result)


# ---------------------------------------------------------------------------------------------- GEMM TN
@pytest.mark.parametrize('M,N,K,splits', [(1000, 200, 136, 1), (1000, 200, 136, 4), (4096, 1544, 512, 8), (100, 64, 64, 3),
                                          (4096, 1544, 512, 0), (8192, 512, 1408, 0), (4096, 300, 700, 16), (2048, 2816, 512, 1), (64, 130, 260, 0), (16384, 32, 512, 0),
                                          (2112, 512, 512, 0), (2112, 512, 512, 8),      # chunks rounded up to 64 rows: the last of 8 splits starts past M
                                          # the weight gradients of the training step at bench size, library-chosen splits (11 / 17 / 20 / 16 row chunks, tfx_gemm_tn_plan)
                                          (65536, 2816, 512, 0), (65536, 1544, 512, 0), (65536, 512, 1408, 0), (65536, 512, 512, 0)])
def test_gemm_tn(M, N, K, splits):
    torch.manual_seed(4)
    lda = (N + 7) // 8 * 8 + 8
    ldb = (K + 7) // 8 * 8
    A = rnd(M, lda, scale=0.5); B = rnd(M, ldb, scale=0.5)
    kv = K - 3
    rowmap = torch.arange(N, device=DEV, dtype=torch.int32).flip(0).contiguous()
    rowmap[5] = -1
    C = torch.full((N, K), 1.0, device=DEV)
    a = capi.make_args('tfx_gemm_tn_args', A=A, lda=lda, a_cols=(N + 7) // 8 * 8, B=B, ldb=ldb, b_cols=ldb, M=M, N=N, K=K,
                       C=C, ldc=K, rowmap=rowmap, k_valid=kv, splits=splits, accumulate=1, alpha=0.5)
    capi.call('tfx_gemm_tn', a, stream())
    prod = 0.5 * (A[:, :N].float().T @ B[:, :K].float())
    ref = torch.full((N, K), 1.0, device=DEV)
    keep = rowmap >= 0
    ref[rowmap[keep].long(), :kv] += prod[keep][:, :kv]
    check(f'gemm_tn {M}x{N}x{K} splits={splits}', C, ref, 5e-3)


@pytest.mark.parametrize('M,N,K,splits,kg', [(4096, 1544, 512, 8, 0), (1000, 200, 136, 4, 0), (2048, 128, 256, 8, 8), (2048, 384, 192, 16, 32),
                                             (8192, 2816, 512, 0, 0), (4096, 520, 320, 0, 40), (65536, 2816, 512, 0, 0)])   # last: net.0 weight + bias gradient at bench size
def test_gemm_tn_folded_bias_gradient_and_head_compaction(M, N, K, splits, kg):
    """`colsum`: the bias gradient (column sums of A through the row map) rides on the weight-gradient GEMM - LDS-DMA kernel (M % 64 == 0)
    and the register-staged fallback; `k_group`: per-head padded product columns are compacted into the unpadded gradient."""
    torch.manual_seed(6)
    lda = (N + 7) // 8 * 8
    A = rnd(M, lda, scale=0.5); B = rnd(M, K, scale=0.5)
    rowmap = torch.randperm(N, device=DEV).to(torch.int32)
    rowmap[3] = -1
    Kout = K if not kg else K // 64 * kg
    C = torch.zeros(N, Kout, device=DEV); bias = torch.full((N,), 2.0, device=DEV)
    a = capi.make_args('tfx_gemm_tn_args', A=A, lda=lda, a_cols=lda, B=B, ldb=K, b_cols=K, M=M, N=N, K=K, C=C, ldc=Kout, rowmap=rowmap, k_valid=K,
                       splits=splits, accumulate=1, alpha=1.0, colsum=bias, k_group=kg)
    capi.call('tfx_gemm_tn', a, stream())
    prod = A[:, :N].float().T @ B.float()
    if kg:
        prod = prod.view(N, K // 64, 64)[:, :, :kg].reshape(N, Kout)
    keep = rowmap >= 0
    ref = torch.zeros(N, Kout, device=DEV); ref[rowmap[keep].long()] = prod[keep]
    bref = torch.full((N,), 2.0, device=DEV); bref[rowmap[keep].long()] += A[:, :N].float().sum(0)[keep]
    check(f'gemm_tn colsum/k_group weight grad {M}x{N}x{K}', C, ref, 5e-3)
    check(f'gemm_tn colsum/k_group bias grad {M}x{N}x{K}', bias, bref, 5e-3)


@pytest.mark.parametrize('M,shapes,groupable', [(8192, [(512, 1408), (2816, 512)], True), (4096, [(512, 512), (1544, 512), (512, 512), (512, 512)], True),
                                                 (1000, [(200, 136), (136, 200)], False), (16384, [(512, 512), (512, 1024)], True), (8192, [(512, 512), (512, 1024)], False)])
def test_gemm_tn_grouped_launch(M, shapes, groupable):
    """round 5: `group_next` - weight-gradient products over the same M rows as ONE launch (a transformer layer's FeedForward pair; to_out + to_qk/v/gates + the skip
    projection).  The group's tiles share one grid of the one-wave kernel (plan kind 3, tiles = the sum); chains the kernel does not take run product by product inside
    the library.  Each member against fp32 torch, with a row map + folded bias gradient on the second member."""
    torch.manual_seed(9)
    structs, keep, refs = [], [], []
    for idx, (N, K) in enumerate(shapes):
        lda = (N + 7) // 8 * 8; ldb = (K + 7) // 8 * 8
        A = rnd(M, lda, scale=0.5); C = torch.zeros(N, K, device=DEV)
        kw = dict(A=A, lda=lda, a_cols=lda, M=M, N=N, K=K, C=C, ldc=K, k_valid=K, splits=0, accumulate=1, alpha=1.0)
        B = rnd(M, ldb, scale=0.5); kw.update(B=B, ldb=ldb, b_cols=ldb); Bfull = B[:, :K]; keep.append(B)
        prod = A[:, :N].float().T @ Bfull.float()
        if idx == 1:
            rowmap = torch.randperm(N, device=DEV).to(torch.int32); bias = torch.zeros(N, device=DEV)
            kw.update(rowmap=rowmap, colsum=bias)
            ref = torch.zeros_like(prod); ref[rowmap.long()] = prod
            bref = torch.zeros(N, device=DEV); bref[rowmap.long()] = A[:, :N].float().sum(0)
            refs.append((C, ref, bias, bref)); keep += [rowmap, bias]
        else:
            refs.append((C, prod, None, None))
        keep += [A, C]
        structs.append(capi.make_args('tfx_gemm_tn_args', **kw))
    for a, b in zip(structs, structs[1:]):
        a.group_next = ctypes.addressof(b)
    out = [ctypes.c_int32(-9) for _ in range(4)]
    assert capi.lib().tfx_gemm_tn_plan(ctypes.byref(structs[0]), *[ctypes.byref(o) for o in out]) == 0
    tiles = sum(((N + 255) // 256) * ((K + 255) // 256) for N, K in shapes)
    if groupable:
        assert out[0].value == 3 and out[1].value == tiles, [o.value for o in out]
    capi.call('tfx_gemm_tn', structs[0], stream())
    for i, (C, ref, bias, bref) in enumerate(refs):
        check(f'grouped gemm_tn member {i} {M}x{shapes[i]}', C, ref, 5e-3)
        if bias is not None:
            check(f'grouped gemm_tn member {i} bias grad', bias, bref, 5e-3)


# ---------------------------------------------------------------------------------------------- attention
def make_kv_end(b, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    kv_end = torch.arange(1, n + 1).repeat(b, 1)
    q_start = torch.arange(n).repeat(b, 1)
    for bi in range(b):
        pos = 3
        while pos < n - 2:
            L = int(torch.randint(1, 9, (1,), generator=g))
            L = min(L, n - pos)
            kv_end[bi, pos:pos + L] = pos + L
            q_start[bi, pos:pos + L] = pos
            pos += L + int(torch.randint(1, 30, (1,), generator=g))
    return kv_end.to(torch.int32), q_start.to(torch.int32)


def attn_ref(q, k, v, gate, kv_end, cap):
    # q,k,v (b,h,n,64) fp32 ; gate (b,h,n)
    n = q.shape[2]
    sim = torch.einsum('bhid,bhjd->bhij', q, k)
    sim = torch.tanh(sim / cap) * cap
    mask = torch.arange(n, device=q.device)[None, None, :] < kv_end[:, :, None]
    sim = sim.masked_fill(~mask[:, None], -torch.finfo(torch.float32).max)
    p = sim.softmax(-1)
    o = torch.einsum('bhij,bhjd->bhid', p, v)
    return o * gate.sigmoid()[..., None]


# (qs, ks): scales of q~ / k~.  (0.35, 2.5): |s / cap| stays in the polynomial branches of the soft-cap (the training regime: QK-RMSNorm bounds
# the scores).  (1.2, 3.0): |s / cap| reaches ~1 - the degree-9 polynomial and, for the waves holding an outlier, the wave-uniform exact
# exp2/rcp branch (attention.hip softcap16).  (3.0, 6.0): scores of +-150 and more - every wave takes the exact branch, tanh saturates
# (1 - tanh^2 -> 0 in the backward), the fixed-reference softmax sees exponents up to cap * log2e = 72.
@pytest.mark.parametrize('b,h,n,qs,ks', [(2, 2, 200, 0.35, 2.5), (1, 3, 128, 0.35, 2.5), (2, 1, 333, 0.35, 2.5), (1, 8, 1024, 0.35, 2.5),
                                         (2, 2, 200, 1.2, 3.0), (1, 2, 333, 3.0, 6.0), (1, 8, 1024, 1.2, 3.0)])
def test_attention_fwd_bwd(b, h, n, qs, ks):
    torch.manual_seed(5)
    T, HD = b * n, h * 64
    ldq, ldv = 2 * HD, 3 * HD + 8
    qk = rnd(T, ldq, scale=1.0)                                   # q | k, token-major
    qk[:, :HD] *= qs                                              # q~ carries the 1/8 scale
    qk[:, HD:] *= ks
    qkv = rnd(T, ldv)                                             # (unused q,k cols) | v | gates
    kv_end, q_start = make_kv_end(b, n)
    kv_end, q_start = kv_end.to(DEV), q_start.to(DEV)
    out = torch.zeros(T, HD, device=DEV, dtype=BF)
    lse = torch.zeros(b, h, n, device=DEV)
    dout = rnd(T, HD)
    do_eff = torch.zeros(T, HD, device=DEV, dtype=BF)
    delta = torch.zeros(b, h, n, device=DEV)
    dqk = torch.zeros(T, ldq, device=DEV, dtype=BF)
    dqkv = torch.zeros(T, ldv, device=DEV, dtype=BF)
    gate_view = qkv[:, 3 * HD:]
    a = capi.make_args(
        'tfx_attn_args', q=qk, k=qk[:, HD:], v=qkv[:, 2 * HD:], ld_q=ldq, ld_k=ldq, ld_v=ldv,
        gate=gate_view, ld_gate=ldv, kv_end=kv_end, q_start=q_start, out=out, ld_out=HD, lse=lse, b=b, h=h, n=n, softcap=50.0,
        dout=dout, ld_dout=HD, do_eff=do_eff, ld_do=HD, delta=delta, dgate=dqkv[:, 3 * HD:], ld_dgate=ldv,
        dq=dqk, dk=dqk[:, HD:], dv=dqkv[:, 2 * HD:], ld_dq=ldq, ld_dk=ldq, ld_dv=ldv)
    capi.call('tfx_attn_fwd', a, stream())
    capi.call('tfx_attn_bwd', a, stream())
    torch.cuda.synchronize()

    def heads(x):  # (T, HD) -> (b,h,n,64)
        return x.float().reshape(b, n, h, 64).transpose(1, 2)
    q = heads(qk[:, :HD]).requires_grad_(True)
    k = heads(qk[:, HD:]).requires_grad_(True)
    v = heads(qkv[:, 2 * HD:3 * HD]).requires_grad_(True)
    g = gate_view[:, :h].float().reshape(b, n, h).transpose(1, 2).requires_grad_(True)
    ref = attn_ref(q, k, v, g, kv_end.long(), 50.0)
    ref.backward(heads(dout))
    check(f'attn fwd b{b} h{h} n{n}', heads(out), ref, 8e-3)
    check('attn dq', heads(dqk[:, :HD]), q.grad, 2e-2)
    check('attn dk', heads(dqk[:, HD:]), k.grad, 2e-2)
    check('attn dv', heads(dqkv[:, 2 * HD:3 * HD]), v.grad, 2e-2)
    check('attn dgate', dqkv[:, 3 * HD:3 * HD + h].float().reshape(b, n, h).transpose(1, 2), g.grad, 2e-2)
    sim = torch.einsum('bhid,bhjd->bhij', q.detach(), k.detach()).abs() / 50.
    print(f'  |s / cap|: median {sim.median():.3f} max {sim.max():.3f}; share above the polynomial bound 0.45: {(sim > 0.45).float().mean():.4f}')
    if qs > 1.:
        assert (sim > 0.45).any(), 'this case must reach the exact-tanh branch'


@pytest.mark.parametrize('b,h,n', [(2, 2, 200), (1, 3, 128), (3, 1, 333), (1, 8, 1024), (2, 2, 64)])
def test_attention_bwd_with_fused_qk_norm_rope_bwd(b, h, n):
    """round 5: the backward of QK-RMSNorm + RoPE in the epilogues of the dQ and dK/dV kernels (tfx_attn_args.nr_*) against the two launches it replaces
    (tfx_attn_bwd writing d q~ | d k~, then tfx_qk_norm_rope_bwd): same d q | d k (raw) up to one bf16 rounding of the same arithmetic, same gain
    gradients, d v / d gate untouched; n = 200 / 333 / 64 leave wave blocks partly or wholly past a sample's end (no read or write there: the
    output buffer is poisoned)."""
    torch.manual_seed(23)
    T, HD = b * n, h * 64
    ld = 3 * HD + 8
    qkv = rnd(T, ld, scale=1.0)                                   # raw q | k | v | gates
    gq = (torch.rand(64, device=DEV) * 2 - 1) * 0.2; gk = (torch.rand(64, device=DEV) * 2 - 1) * 0.2
    pos = torch.randint(0, 50, (T,), device=DEV, dtype=torch.int32)
    ang = torch.arange(50, device=DEV)[:, None] * (10000. ** (-torch.arange(32, device=DEV) / 32.))[None, :]
    cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
    qk = torch.zeros(T, 2 * HD, device=DEV, dtype=BF)
    fa = capi.make_args('tfx_qk_norm_rope_args', T=T, H=h, qkv=qkv, ld_qkv=ld, qk=qk, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk,
                        rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t, q_scale=0.125)
    capi.call('tfx_qk_norm_rope_fwd', fa, stream())
    kv_end, q_start = make_kv_end(b, n)
    kv_end, q_start = kv_end.to(DEV), q_start.to(DEV)
    out = torch.zeros(T, HD, device=DEV, dtype=BF); lse = torch.zeros(b, h, n, device=DEV)
    dout = rnd(T, HD)
    res = []
    for fused in (False, True):
        do_eff = torch.zeros(T, HD, device=DEV, dtype=BF); delta = torch.zeros(b, h, n, device=DEV)
        dqk = torch.full((T, 2 * HD), float('nan'), device=DEV, dtype=BF)
        dqkv = torch.full((T, ld), float('nan'), device=DEV, dtype=BF)
        dgq, dgk = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
        kw = dict(q=qk, k=qk[:, HD:], v=qkv[:, 2 * HD:], ld_q=2 * HD, ld_k=2 * HD, ld_v=ld, gate=qkv[:, 3 * HD:], ld_gate=ld, kv_end=kv_end, q_start=q_start,
                  out=out, ld_out=HD, lse=lse, b=b, h=h, n=n, softcap=50.0, dout=dout, ld_dout=HD, do_eff=do_eff, ld_do=HD, delta=delta,
                  dgate=dqkv[:, 3 * HD:], ld_dgate=ld, dq=dqk, dk=dqk[:, HD:], dv=dqkv[:, 2 * HD:], ld_dq=2 * HD, ld_dk=2 * HD, ld_dv=ld)
        if fused:
            kw.update(nr_qkv=qkv, nr_ld_qkv=ld, nr_dqkv=dqkv, nr_ld_dqkv=ld, nr_gamma_q=gq, nr_gamma_k=gk, nr_rot_pos=pos, nr_cos=cos_t, nr_sin=sin_t,
                      nr_q_scale=0.125, nr_norm_scale=8.0, nr_dgamma_q=dgq, nr_dgamma_k=dgk)
        a = capi.make_args('tfx_attn_args', **kw)
        if not fused:
            capi.call('tfx_attn_fwd', a, stream())
        capi.call('tfx_attn_bwd', a, stream())
        if not fused:
            ba = capi.make_args('tfx_qk_norm_rope_args', T=T, H=h, qkv=qkv, ld_qkv=ld, gamma_q=gq, gamma_k=gk, rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t,
                                q_scale=0.125, norm_scale=8.0, dqk=dqk, ld_dqk=2 * HD, dqkv=dqkv, ld_dqkv=ld, dgamma_q=dgq, dgamma_k=dgk)
            capi.call('tfx_qk_norm_rope_bwd', ba, stream())
        else:
            assert torch.isnan(dqk.float()).all(), 'the fused form must not write d q~ / d k~'
        torch.cuda.synchronize()
        res.append((dqkv.clone(), dgq.clone(), dgk.clone()))
    (d0, gq0, gk0) = res[0]
    for tag, (d1, gq1, gk1) in zip(('atomics',), res[1:]):
        assert torch.isfinite(d1[:, :3 * HD + h].float()).all(), 'every d q | d k | d v | d gate element must have been written'
        check(f'fused ({tag}): d q (raw)', d1[:, :HD], d0[:, :HD].float(), 2e-4)               # measured 0 ... 1e-5: a handful of one-ulp bf16 flips
        check(f'fused ({tag}): d k (raw)', d1[:, HD:2 * HD], d0[:, HD:2 * HD].float(), 2e-4)       # measured 0 ... 2.4e-5
        assert torch.equal(d1[:, 2 * HD:3 * HD + h], d0[:, 2 * HD:3 * HD + h]), 'd v / d gate must not change'
        check(f'fused ({tag}): d gamma_q', gq1, gq0, 2e-5)                                       # measured 1e-7 ... 3e-7 (summation order)
        check(f'fused ({tag}): d gamma_k', gk1, gk0, 2e-5)
    # ---- and DIRECTLY against fp32 torch autograd through norm -> RoPE -> attention (the fused outputs had only been compared with the unfused HIP form)
    x = qkv[:, :2 * HD].float().reshape(T, 2, h, 64).requires_grad_(True)
    gqr, gkr = gq.clone().requires_grad_(True), gk.clone().requires_grad_(True)
    y = F.normalize(x, dim=-1) * 8 * (torch.stack([gqr, gkr])[None, :, None, :] + 1)
    angp = ang[pos.long()].repeat_interleave(2, dim=-1)[:, None, None, :]
    y2 = y.reshape(T, 2, h, 32, 2)
    rot = torch.stack((-y2[..., 1], y2[..., 0]), -1).reshape(T, 2, h, 64)
    qkr = (y * angp.cos() + rot * angp.sin()) * torch.tensor([0.125, 1.0], device=DEV)[None, :, None, None]
    hd4 = lambda t: t.reshape(b, n, h, 64).transpose(1, 2)
    vr = qkv[:, 2 * HD:3 * HD].float().requires_grad_(True)
    gr = qkv[:, 3 * HD:3 * HD + h].float().requires_grad_(True)
    ref = attn_ref(hd4(qkr[:, 0]), hd4(qkr[:, 1]), hd4(vr.reshape(T, h, 64)), gr.reshape(b, n, h).transpose(1, 2), kv_end.long(), 50.0)
    ref.backward(hd4(dout.float().reshape(T, h, 64)))
    d1, gq1, gk1 = res[1]
    check('fused vs fp32 autograd: d q (raw)', d1[:, :HD].reshape(T, h, 64), x.grad[:, 0], 2.5e-2)
    check('fused vs fp32 autograd: d k (raw)', d1[:, HD:2 * HD].reshape(T, h, 64), x.grad[:, 1], 2.5e-2)
    check('fused vs fp32 autograd: d v', d1[:, 2 * HD:3 * HD], vr.grad, 2e-2)
    check('fused vs fp32 autograd: d gamma_q', gq1, gqr.grad, 2e-2)
    check('fused vs fp32 autograd: d gamma_k', gk1, gkr.grad, 2e-2)


@pytest.mark.parametrize('gscale,want_mode', [(0.04, 0), (0.22, 1), (1.0, 2)])
def test_attention_softcap_plan_from_qk_norm_bound(gscale, want_mode):
    """round 4: the soft-cap polynomial's degree is a property of the LAYER - QK-RMSNorm bounds |q~ . k~| by
    B = 8 max|1 + gamma_q| max|1 + gamma_k| - so tfx_qk_norm_rope_fwd writes a plan (mode, Chebyshev-economised coefficients, derivative
    coefficients, B) and the attention kernels drop the per-score |s| maximum, the wave vote and the degree branch.  Checks: the mode thresholds,
    B really bounds the scores of ADVERSARIAL inputs (keys aligned with queries, so the scores reach the bound), forward and backward with the plan
    equal the fp32 reference as closely as without it."""
    torch.manual_seed(21)
    b, h, n = 2, 2, 384
    T, HD = b * n, h * 64
    ld = 3 * HD + 8
    qkv = rnd(T, ld, scale=1.0)
    # adversarial: every 7th key vector parallel to its own query AFTER the gains (so that q~ . k~ reaches norm_scale^2 q_scale max.. for some pairs)
    gq = (torch.rand(64, device=DEV) * 2 - 1) * gscale; gk = (torch.rand(64, device=DEV) * 2 - 1) * gscale
    gq[3] = gscale; gk[3] = gscale                                 # the maxima sit on the same coordinate ...
    qkv[::7, :HD] = 0; qkv[::7, 3:HD:64] = 2.0                     # ... and these tokens' q and k are that coordinate's unit vector
    qkv[::7, HD:2 * HD] = 0; qkv[::7, HD + 3:2 * HD:64] = 2.0
    pos = torch.zeros(T, device=DEV, dtype=torch.int32)           # no rotation: aligned pairs stay aligned
    cos_t, sin_t = torch.ones(8, 32, device=DEV), torch.zeros(8, 32, device=DEV)
    qk = torch.zeros(T, 2 * HD, device=DEV, dtype=BF)
    plan = torch.full((8,), float('nan'), device=DEV)
    a = capi.make_args('tfx_qk_norm_rope_args', T=T, H=h, qkv=qkv, ld_qkv=ld, qk=qk, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk,
                       rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t, q_scale=0.125, sc_plan=plan, softcap=50.0)
    capi.call('tfx_qk_norm_rope_fwd', a, stream())
    torch.cuda.synchronize()
    pl = plan.cpu()
    B = 1.02 * 8 * (1 + gscale) ** 2
    assert int(pl[0]) == want_mode and abs(float(pl[7]) - B) <= 1e-4 * B, (pl, B)

    def heads(x):
        return x.float().reshape(b, n, h, 64).transpose(1, 2)
    smax = torch.einsum('bhid,bhjd->bhij', heads(qk[:, :HD]), heads(qk[:, HD:])).abs().max().item()
    print(f'  bound B = {float(pl[7]):.3f}, largest |score| = {smax:.3f}, mode {int(pl[0])}')
    assert smax <= float(pl[7]) and smax >= 0.9 * 8 * (1 + gscale) ** 2       # the bound holds and is nearly attained
    kv_end, q_start = make_kv_end(b, n)
    kv_end, q_start = kv_end.to(DEV), q_start.to(DEV)
    vg = rnd(T, HD + 8)
    dout = rnd(T, HD)
    res = []
    for use_plan in (False, True):
        out = torch.zeros(T, HD, device=DEV, dtype=BF); lse = torch.zeros(b, h, n, device=DEV)
        do_eff = torch.zeros(T, HD, device=DEV, dtype=BF); delta = torch.zeros(b, h, n, device=DEV)
        dqk = torch.zeros(T, 2 * HD, device=DEV, dtype=BF); dvg = torch.zeros(T, HD + 8, device=DEV, dtype=BF)
        aa = capi.make_args('tfx_attn_args', q=qk, k=qk[:, HD:], v=vg, ld_q=2 * HD, ld_k=2 * HD, ld_v=HD + 8, gate=vg[:, HD:], ld_gate=HD + 8,
                            kv_end=kv_end, q_start=q_start, out=out, ld_out=HD, lse=lse, b=b, h=h, n=n, softcap=50.0, dout=dout, ld_dout=HD,
                            do_eff=do_eff, ld_do=HD, delta=delta, dgate=dvg[:, HD:], ld_dgate=HD + 8, dq=dqk, dk=dqk[:, HD:], dv=dvg, ld_dq=2 * HD,
                            ld_dk=2 * HD, ld_dv=HD + 8, sc_plan=plan if use_plan else None)
        capi.call('tfx_attn_fwd', aa, stream()); capi.call('tfx_attn_bwd', aa, stream())
        torch.cuda.synchronize()
        res.append((out.clone(), dqk.clone(), dvg.clone()))
    q = heads(qk[:, :HD]).requires_grad_(True); k = heads(qk[:, HD:]).requires_grad_(True)
    v = heads(vg[:, :HD]).requires_grad_(True); g = vg[:, HD:HD + h].float().reshape(b, n, h).transpose(1, 2).requires_grad_(True)
    ref = attn_ref(q, k, v, g, kv_end.long(), 50.0)
    ref.backward(heads(dout))
    errs = []
    for out, dqk, dvg in res:
        errs.append((relerr(heads(out), ref), relerr(heads(dqk[:, :HD]), q.grad), relerr(heads(dqk[:, HD:]), k.grad), relerr(heads(dvg[:, :HD]), v.grad)))
    print('  rel err (out, dq, dk, dv) without plan:', ['%.2e' % e for e in errs[0]], ' with plan:', ['%.2e' % e for e in errs[1]])
    for e0, e1, tol in zip(errs[0], errs[1], (8e-3, 2e-2, 2e-2, 2e-2)):
        assert e1 <= tol and e1 <= 1.25 * e0 + 1e-4


@pytest.mark.parametrize('entry', ['tfx_attn_fwd', 'tfx_decode_attn'])     # the forward kernel with cache addressing / the decode entry (<= 2 rows per sample: one block per (sample, head), no matrix cores)
@pytest.mark.parametrize('b,h,lq,n_kv', [(3, 2, 1, 300), (2, 4, 4, 200), (5, 1, 6, 64), (2, 8, 16, 1100), (64, 8, 1, 333), (7, 3, 2, 77), (128, 8, 5, 330), (3, 2, 8, 100), (2, 2, 3, 9), (2, 2, 7, 40)])
def test_attention_fwd_against_kv_cache(b, h, lq, n_kv, entry):
    """decode-time call (engine.Plan(cache=...)): `lq` new query rows per sample against keys / values that live in a LONGER per-sample
    cache buffer (`n_kv` > n rows, token-major [b, n_kv, 2 * h * 64] = k~ | v), each query row with its own visible length `kv_end`
    (own prefix + the block being decoded; rows of finished samples see a short prefix).  Reference: masked softmax in fp32."""
    torch.manual_seed(11)
    HD = h * 64
    T = b * lq
    q = rnd(T, HD, scale=0.35)
    cache = rnd(b * n_kv, 2 * HD)
    cache[:, :HD] *= 2.5
    gates = rnd(T, 8 + h)
    g = torch.Generator().manual_seed(3)
    kv_end = torch.stack([torch.randint(1, n_kv + 1, (1,), generator=g).expand(lq) for _ in range(b)]).clone()
    if b > 2:                                                     # rows of one sample with DIFFERENT visible lengths (a text row next to a block, mixed decode steps)
        kv_end[1] = torch.randint(1, n_kv + 1, (lq,), generator=g)
    kv_end[0] = n_kv                                              # a full cache
    kv_end[-1] = 1                                                # a single visible key
    kv_end = kv_end.to(torch.int32).to(DEV)
    out = torch.full((T, HD), float('nan'), device=DEV, dtype=BF)
    lse = torch.zeros(b, h, lq, device=DEV)
    a = capi.make_args('tfx_attn_args', q=q, k=cache, v=cache[:, HD:], ld_q=HD, ld_k=2 * HD, ld_v=2 * HD, gate=gates[:, 8:], ld_gate=8 + h,
                       kv_end=kv_end, q_start=torch.zeros(T, dtype=torch.int32, device=DEV), out=out, ld_out=HD, lse=lse, b=b, h=h, n=lq,
                       n_kv=n_kv, softcap=50.0)
    capi.call(entry, a, stream())
    torch.cuda.synchronize()
    qh = q.float().reshape(b, lq, h, 64).transpose(1, 2)
    kh = cache[:, :HD].float().reshape(b, n_kv, h, 64).transpose(1, 2)
    vh = cache[:, HD:].float().reshape(b, n_kv, h, 64).transpose(1, 2)
    sim = torch.tanh(torch.einsum('bhid,bhjd->bhij', qh, kh) / 50.) * 50.
    mask = torch.arange(n_kv, device=DEV)[None, None, :] < kv_end.long()[:, :, None]
    sim = sim.masked_fill(~mask[:, None], -torch.finfo(torch.float32).max)
    ref = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), vh) * gates[:, 8:8 + h].float().reshape(b, lq, h).transpose(1, 2).sigmoid()[..., None]
    check(f'attn fwd vs cache b{b} h{h} lq{lq} n_kv{n_kv}', out.float().reshape(b, lq, h, 64).transpose(1, 2), ref, 8e-3)
    ref_lse = torch.logsumexp(sim, dim=-1)
    assert (lse - ref_lse).abs().max() <= 2e-2


# ---------------------------------------------------------------------------------------------- token-wise
def tok_setup(T, d, I, seed=0):
    torch.manual_seed(seed)
    tok_inst = torch.full((T,), -1, dtype=torch.int32)
    idx = torch.randperm(T)[: T // 3]
    tok_inst[idx] = torch.randint(0, I, (len(idx),), dtype=torch.int32)
    ld = 3 * d + 8
    table = torch.randn(I, ld, device=DEV) * 0.5
    return tok_inst.to(DEV), table, ld


@pytest.mark.parametrize('T,d', [(1000, 512), (333, 64), (257, 768), (130, 1024)])
def test_adaln_pre_post(T, d):
    I = 37
    tok_inst, table, ld = tok_setup(T, d, I)
    x = rnd(T, d, scale=2.0); gt = torch.randn(d, device=DEV) * 0.3
    u = torch.zeros(T, d, device=DEV, dtype=BF)
    mean = torch.zeros(T, device=DEV); rstd = torch.zeros(T, device=DEV)
    du = rnd(T, d)
    dx = rnd(T, d); dx0 = dx.float().clone()
    dtable = torch.zeros_like(table); dgt = torch.zeros(d, device=DEV)
    a = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=x, u=u, tok_inst=tok_inst, table=table, ld_table=ld, gamma_text=gt,
                       mean=mean, rstd=rstd, du=du, dx=dx, dtable=dtable, dgamma_text=dgt)
    capi.call('tfx_adaln_pre_fwd', a, stream())
    capi.call('tfx_adaln_pre_bwd', a, stream())
    xr = x.float().requires_grad_(True); tr = table.clone().requires_grad_(True); gr = gt.clone().requires_grad_(True)
    im = (tok_inst >= 0)[:, None]
    ii = tok_inst.clamp(min=0).long()
    xh = F.layer_norm(xr, (d,))
    ref = torch.where(im, xh * (tr[ii, :d] + 1) + tr[ii, d:2 * d], xh * (gr + 1))
    ref.backward(du.float())
    check(f'adaln_pre fwd d{d}', u, ref, 6e-3)
    check('adaln_pre dx', dx.float() - dx0, xr.grad, 1.5e-2)
    check('adaln_pre dtable', dtable, tr.grad, 5e-3)
    check('adaln_pre dgamma_text', dgt, gr.grad, 5e-3)
    # post
    y = rnd(T, d); ls = torch.randn(d, device=DEV) * 0.3
    out = torch.zeros(T, d, device=DEV, dtype=BF); g = rnd(T, d); dy = torch.zeros(T, d, device=DEV, dtype=BF)
    dtable2 = torch.zeros_like(table); dls = torch.zeros(d, device=DEV)
    a = capi.make_args('tfx_adaln_post_args', T=T, d=d, x=x, y=y, out=out, tok_inst=tok_inst, table=table, ld_table=ld,
                       layerscale=ls, g=g, dy=dy, dtable=dtable2, dlayerscale=dls)
    capi.call('tfx_adaln_post_fwd', a, stream())
    capi.call('tfx_adaln_post_bwd', a, stream())
    yr = y.float().requires_grad_(True); tr = table.clone().requires_grad_(True); lr = ls.clone().requires_grad_(True)
    ref = x.float() + torch.where(im, yr * tr[ii, 2 * d:3 * d].sigmoid(), yr * (lr + 1))
    ref.backward(g.float())
    check('adaln_post fwd', out, ref, 6e-3)
    check('adaln_post dy', dy, yr.grad, 6e-3)
    check('adaln_post dtable', dtable2, tr.grad, 5e-3)
    check('adaln_post dlayerscale', dls, lr.grad, 5e-3)


@pytest.mark.parametrize('T,d', [(777, 512), (100, 128), (130, 1024)])
def test_rmsnorm(T, d):
    torch.manual_seed(0)
    x = rnd(T, d, scale=3.0); gm = torch.randn(d, device=DEV) * 0.3
    y = torch.zeros(T, d, device=DEV, dtype=BF); dy = rnd(T, d); dx = torch.zeros(T, d, device=DEV, dtype=BF); dg = torch.zeros(d, device=DEV)
    a = capi.make_args('tfx_rmsnorm_args', T=T, d=d, x=x, y=y, gamma=gm, dy=dy, dx=dx, dgamma=dg)
    capi.call('tfx_rmsnorm_fwd', a, stream()); capi.call('tfx_rmsnorm_bwd', a, stream())
    xr = x.float().requires_grad_(True); gr = gm.clone().requires_grad_(True)
    ref = F.normalize(xr, dim=-1) * d ** 0.5 * (gr + 1)
    ref.backward(dy.float())
    check('rmsnorm fwd', y, ref, 6e-3); check('rmsnorm dx', dx, xr.grad, 1e-2); check('rmsnorm dgamma', dg, gr.grad, 5e-3)


@pytest.mark.parametrize('T,d,L', [(500, 512, 5), (300, 64, 3), (100, 1024, 9), (64, 256, 1)])
def test_attnres(T, d, L):
    torch.manual_seed(0)
    H = rnd(L, T, d, scale=2.0); gm = torch.randn(d, device=DEV) * 0.3; pq = torch.randn(d, device=DEV) * 0.5
    out = torch.zeros(T, d, device=DEV, dtype=BF)
    g = rnd(T, d); g2 = rnd(T, d)
    dH = rnd(L, T, d); dH0 = dH.float().clone()
    dgm = torch.zeros(d, device=DEV); dpq = torch.zeros(d, device=DEV)
    a = capi.make_args('tfx_attnres_args', T=T, d=d, L=L, hiddens=H, stride_h=T * d, gamma=gm, pq=pq, out=out, g=g, g2=g2,
                       dhiddens=dH, stride_dh=T * d, first=0, dgamma=dgm, dpq=dpq)
    capi.call('tfx_attnres_fwd', a, stream()); capi.call('tfx_attnres_bwd', a, stream())
    Hr = H.float().requires_grad_(True); gr = gm.clone().requires_grad_(True); pr = pq.clone().requires_grad_(True)
    keys = F.normalize(Hr, dim=-1) * d ** 0.5 * (gr + 1)
    sim = torch.einsum('ltd,d->tl', keys, pr) * d ** -0.5
    ref = torch.einsum('tl,ltd->td', sim.softmax(-1), Hr)
    ref.backward(g.float() + g2.float())
    check(f'attnres fwd L{L} d{d}', out, ref, 6e-3)
    check('attnres dH (accumulate)', dH.float() - dH0, Hr.grad, 2e-2)
    check('attnres dgamma', dgm, gr.grad, 1e-2); check('attnres dpq', dpq, pr.grad, 1e-2)
    dH2 = torch.full_like(dH, float('nan'))
    a = capi.make_args('tfx_attnres_args', T=T, d=d, L=L, hiddens=H, stride_h=T * d, gamma=gm, pq=pq, out=out, g=g,
                       dhiddens=dH2, stride_dh=T * d, first=1, dgamma=dgm, dpq=dpq)
    capi.call('tfx_attnres_bwd', a, stream())
    Hr.grad = None
    keys = F.normalize(Hr, dim=-1) * d ** 0.5 * (gm + 1)
    sim = torch.einsum('ltd,d->tl', keys, pq) * d ** -0.5
    torch.einsum('tl,ltd->td', sim.softmax(-1), Hr).backward(g.float())
    check('attnres dH (first=store)', dH2, Hr.grad, 1e-2)


@pytest.mark.parametrize('T,H', [(500, 2), (1000, 8)])
def test_qk_norm_rope(T, H):
    torch.manual_seed(0)
    HD = H * 64; ld = 3 * HD + 8
    qkv = rnd(T, ld, scale=1.5)
    gq = torch.randn(64, device=DEV) * 0.3; gk = torch.randn(64, device=DEV) * 0.3
    pos = torch.randint(0, 900, (T,), device=DEV, dtype=torch.int32)
    freqs = 1. / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(1024).float()[:, None] * freqs[None]
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    qk = torch.zeros(T, 2 * HD, device=DEV, dtype=BF)
    dqk = rnd(T, 2 * HD); dqkv = torch.zeros(T, ld, device=DEV, dtype=BF)
    dgq = torch.zeros(64, device=DEV); dgk = torch.zeros(64, device=DEV)
    a = capi.make_args('tfx_qk_norm_rope_args', T=T, H=H, qkv=qkv, ld_qkv=ld, qk=qk, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk,
                       rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t, q_scale=0.125, dqk=dqk, ld_dqk=2 * HD, dqkv=dqkv, ld_dqkv=ld,
                       dgamma_q=dgq, dgamma_k=dgk)
    capi.call('tfx_qk_norm_rope_fwd', a, stream()); capi.call('tfx_qk_norm_rope_bwd', a, stream())
    x = qkv[:, :2 * HD].float().reshape(T, 2, H, 64).requires_grad_(True)
    gqr = gq.clone().requires_grad_(True); gkr = gk.clone().requires_grad_(True)
    gam = torch.stack([gqr, gkr])[None, :, None, :]
    y = F.normalize(x, dim=-1) * 8 * (gam + 1)
    angp = ang.to(DEV)[pos.long()].repeat_interleave(2, dim=-1)[:, None, None, :]
    y2 = y.reshape(T, 2, H, 32, 2)
    rot = torch.stack((-y2[..., 1], y2[..., 0]), -1).reshape(T, 2, H, 64)
    out = y * angp.cos() + rot * angp.sin()
    out = out * torch.tensor([0.125, 1.0], device=DEV)[None, :, None, None]
    out.backward(dqk.float().reshape(T, 2, H, 64))
    check('qk_norm_rope fwd', qk.reshape(T, 2, H, 64), out, 6e-3)
    check('qk_norm_rope dx', dqkv[:, :2 * HD].reshape(T, 2, H, 64), x.grad, 1.2e-2)
    check('qk_norm_rope dgamma_q', dgq, gqr.grad, 1e-2); check('qk_norm_rope dgamma_k', dgk, gkr.grad, 1e-2)


@pytest.mark.parametrize('T,H', [(66000, 8), (65536, 8), (1000, 2)])     # two shapes of the 256 x 256 kernel (ragged last row tile / the bench's own) and one it does not take
def test_gemm_nt_fused_qk_norm_rope_epilogue_is_bit_identical(T, H):
    """round 4, SURVEY K4 (T:946-965): TFX_EPI_QKV_NORM_ROPE - the [q | k | v | gates] projection whose epilogue also norms and rotates q, k - against
    the plain projection followed by tfx_qk_norm_rope_fwd: the raw projection, q~ | k~ and the soft-cap plan must be IDENTICAL bit for bit (the
    fused epilogue runs the token-wise kernel's arithmetic on the same bf16-rounded values)."""
    torch.manual_seed(31)
    d, HD = 512, H * 64
    N = 3 * HD + H; ldq = (N + 63) // 64 * 64
    u, W = rnd(T, d), rnd(N, d, scale=d ** -0.5)
    gq = torch.randn(64, device=DEV) * 0.2; gk = torch.randn(64, device=DEV) * 0.2
    pos = torch.randint(0, 1000, (T,), device=DEV, dtype=torch.int32)
    freqs = 1. / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(1024).float()[:, None] * freqs[None]
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    # reference: two launches
    C0 = torch.full((T, ldq), float('nan'), device=DEV, dtype=BF); qk0 = torch.full((T, 2 * HD), float('nan'), device=DEV, dtype=BF)
    plan0 = torch.full((8,), float('nan'), device=DEV)
    gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_BF16'], C=C0, ldc=ldq)
    a = capi.make_args('tfx_qk_norm_rope_args', T=T, H=H, qkv=C0, ld_qkv=ldq, qk=qk0, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk, rot_pos=pos, cos_tab=cos_t,
                       sin_tab=sin_t, q_scale=0.125, norm_scale=8.0, sc_plan=plan0, softcap=50.0)
    capi.call('tfx_qk_norm_rope_fwd', a, stream())
    # fused
    C1 = torch.full((T, ldq), float('nan'), device=DEV, dtype=BF); qk1 = torch.full((T, 2 * HD), float('nan'), device=DEV, dtype=BF)
    plan1 = torch.full((8,), float('nan'), device=DEV)
    gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_QKV_NORM_ROPE'], C=C1, ldc=ldq, C2=qk1, ldc2=2 * HD, qk_heads=H,
            qk_gamma_q=gq, qk_gamma_k=gk, qk_rot_pos=pos, qk_cos=cos_t, qk_sin=sin_t, qk_q_scale=0.125, qk_norm_scale=8.0, qk_plan=plan1, qk_softcap=50.0)
    torch.cuda.synchronize()
    assert torch.isfinite(qk1.float()).all() and torch.isfinite(C1[:, :N].float()).all()
    assert torch.equal(C1[:, :N], C0[:, :N]), 'raw projection'
    assert torch.equal(qk1, qk0), f'q~ | k~ differ in {(qk1 != qk0).sum().item()} elements'
    assert torch.equal(plan1, plan0)
    kind, grid = ctypes.c_int32(-1), ctypes.c_int32(-1)
    pa = capi.make_args('tfx_gemm_nt_args', A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_QKV_NORM_ROPE'], C=C0, ldc=ldq)
    capi.lib().tfx_gemm_nt_plan(ctypes.byref(pa), ctypes.byref(kind), ctypes.byref(grid))
    assert (kind.value == 3) == (T > 60000)                        # the large shapes really took the ping-pong kernel (fused), the small one the two launches


@pytest.mark.parametrize('T', [640, 600, 130])
def test_decode_step_qkv_projection_with_fused_norm_rope_and_cache_append_is_bit_identical(T):
    """decode steps: the [q | k | v | gates] projection on the decode-step GEMM kernel (M <= 1024) with TFX_EPI_QKV_NORM_ROPE + `qk_cache` - q~ | k~, the raw
    projection AND this step's k~ / v rows in the KV cache (rows `qk_cache_pos`, -1 = skip) - against the plain projection followed by
    tfx_qk_norm_rope_fwd with its `cache` arguments: everything identical bit for bit, cache rows that are not addressed untouched."""
    torch.manual_seed(41)
    d, H = 1024, 8
    HD = H * 64
    N = 3 * HD + H; ldq = (N + 63) // 64 * 64
    u, W = rnd(T, d), rnd(N, d, scale=d ** -0.5)
    gq = torch.randn(64, device=DEV) * 0.2; gk = torch.randn(64, device=DEV) * 0.2
    pos = torch.randint(0, 1000, (T,), device=DEV, dtype=torch.int32)
    freqs = 1. / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(1024).float()[:, None] * freqs[None]
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    rows_cache = 4 * T
    cpos = torch.randperm(rows_cache, device=DEV)[:T].to(torch.int32)
    cpos[torch.rand(T, device=DEV) < 0.2] = -1
    kind, grid = ctypes.c_int32(-1), ctypes.c_int32(-1)
    pa = capi.make_args('tfx_gemm_nt_args', A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_BF16'], C=torch.empty(T, ldq, device=DEV, dtype=BF), ldc=ldq)
    capi.lib().tfx_gemm_nt_plan(ctypes.byref(pa), ctypes.byref(kind), ctypes.byref(grid))
    assert kind.value == 5, 'the shape must run on the decode-step kernel'
    outs = []
    for fused in (False, True):
        C = torch.full((T, ldq), float('nan'), device=DEV, dtype=BF); qk = torch.full((T, 2 * HD), float('nan'), device=DEV, dtype=BF)
        cache = torch.full((rows_cache, 2 * HD), 7.0, device=DEV, dtype=BF); plan = torch.full((8,), float('nan'), device=DEV)
        if fused:
            gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_QKV_NORM_ROPE'], C=C, ldc=ldq, C2=qk, ldc2=2 * HD, qk_heads=H, qk_gamma_q=gq,
                    qk_gamma_k=gk, qk_rot_pos=pos, qk_cos=cos_t, qk_sin=sin_t, qk_q_scale=0.125, qk_norm_scale=8.0, qk_plan=plan, qk_softcap=50.0,
                    qk_cache=cache, qk_ld_cache=2 * HD, qk_cache_pos=cpos)
        else:
            gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=ldq)
            a = capi.make_args('tfx_qk_norm_rope_args', T=T, H=H, qkv=C, ld_qkv=ldq, qk=qk, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk, rot_pos=pos, cos_tab=cos_t,
                               sin_tab=sin_t, q_scale=0.125, norm_scale=8.0, sc_plan=plan, softcap=50.0, cache=cache, ld_cache=2 * HD, cache_pos=cpos)
            capi.call('tfx_qk_norm_rope_fwd', a, stream())
        torch.cuda.synchronize()
        outs.append((C[:, :N].clone(), qk, cache, plan))
    (C0, qk0, ca0, pl0), (C1, qk1, ca1, pl1) = outs
    assert torch.isfinite(qk1.float()).all() and torch.isfinite(C1.float()).all()
    assert torch.equal(C1, C0), 'raw projection'
    assert torch.equal(qk1, qk0), f'q~ | k~ differ in {(qk1 != qk0).sum().item()} elements'
    assert torch.equal(ca1, ca0), f'cache differs in {(ca1 != ca0).sum().item()} elements'
    assert torch.equal(pl1, pl0)
    live = cpos[cpos >= 0].long()
    assert (ca1[live] != 7.0).any() and torch.equal(ca1[live][:, HD:], C1[(cpos >= 0)][:, 2 * HD:3 * HD])       # v rows are the raw projection's


def test_embed_noise_fourier():
    torch.manual_seed(0)
    T, d, V = 700, 512, 390
    ids = torch.randint(-1, V, (T,), device=DEV, dtype=torch.int32)
    tok_inst = torch.where(torch.rand(T, device=DEV) < 0.3, torch.zeros(T, device=DEV), -torch.ones(T, device=DEV)).to(torch.int32)
    table = rnd(V, d)
    x = torch.zeros(T, d, device=DEV, dtype=BF); dx = rnd(T, d); dtab = torch.zeros(V, d, device=DEV)
    a = capi.make_args('tfx_embed_args', T=T, d=d, text_ids=ids, tok_inst=tok_inst, table=table, x=x, dx=dx, dtable=dtab)
    capi.call('tfx_embed_fwd', a, stream()); capi.call('tfx_embed_bwd', a, stream())
    text = tok_inst < 0
    ref = torch.where(text[:, None], table[ids.clamp(min=0).long()].float(), torch.zeros(T, d, device=DEV))
    check('embed fwd', x, ref, 1e-6)
    dref = torch.zeros(V, d, device=DEV).index_add_(0, ids.clamp(min=0).long()[text], dx.float()[text])
    check('embed bwd', dtab, dref, 1e-5)
    # noise mix
    R, dl, ldx = 300, 48, 64
    xs = torch.randn(R, dl, device=DEV); eps = torch.randn(R, dl, device=DEV)
    I = 40
    row_inst = torch.randint(0, I, (R,), device=DEV, dtype=torch.int32); times = torch.rand(I, device=DEV)
    xt = torch.full((R, ldx), float('nan'), device=DEV, dtype=BF); flow = torch.zeros(R, dl, device=DEV)
    a = capi.make_args('tfx_noise_mix_args', R=R, dl=dl, x=xs, eps=eps, row_inst=row_inst, inst_time=times, xt=xt, ld_xt=ldx, flow=flow)
    capi.call('tfx_noise_mix', a, stream())
    tt = times[row_inst.long()][:, None]
    check('noise_mix xt', xt[:, :dl], xs * tt + eps * (1 - tt), 4e-3)
    assert (xt[:, dl:] == 0).all()
    check('noise_mix flow', flow, xs - eps, 1e-6)
    # fourier
    half, ldf = 32, 128
    w = torch.randn(half, device=DEV); out = torch.full((I, ldf), float('nan'), device=DEV, dtype=BF)
    a = capi.make_args('tfx_fourier_args', I=I, half=half, times=times, w=w, out=out, ld=ldf)
    capi.call('tfx_fourier', a, stream())
    fr = times[:, None] * w[None] * 2 * math.pi
    check('fourier', out[:, :2 * half + 1], torch.cat([times[:, None], fr.sin(), fr.cos()], -1), 4e-3)
    assert (out[:, 2 * half + 1:] == 0).all()


@pytest.mark.parametrize('T,V,ld', [(1000, 390, 448), (777, 259, 264), (300, 700, 704)])      # rows of <= 512 logits stay in registers; wider ones take three passes
def test_losses(T, V, ld):
    torch.manual_seed(0)
    logits = torch.randn(T, ld, device=DEV) * 2
    labels = torch.randint(0, V, (T,), device=DEV, dtype=torch.int32)
    labels[::3] = -1
    dlog = torch.full((T, ld), float('nan'), device=DEV, dtype=BF); acc = torch.zeros(4, device=DEV)
    scale = 1.0 / 1234
    a = capi.make_args('tfx_ce_args', T=T, V=V, logits=logits, ld=ld, labels=labels, grad_scale=scale, dlogits=dlog, ld_d=ld, acc=acc)
    capi.call('tfx_ce_fwd_bwd', a, stream())
    lr = logits[:, :V].clone().requires_grad_(True)
    ce = F.cross_entropy(lr, labels.long(), ignore_index=-1, reduction='sum')
    (ce * scale).backward()
    assert abs(acc[0].item() - ce.item()) < 1e-3 * ce.item()
    assert acc[1].item() == (labels >= 0).sum().item()
    check('ce dlogits', dlog[:, :V], lr.grad, 6e-3)
    assert (dlog[:, V:] == 0).all()
    R, dl, ldp, ldd = 500, 48, 48, 64
    pred = torch.randn(R, ldp, device=DEV); flow = torch.randn(R, dl, device=DEV)
    dp = torch.full((R, ldd), float('nan'), device=DEV, dtype=BF); acc = torch.zeros(4, device=DEV)
    a = capi.make_args('tfx_mse_args', R=R, dl=dl, pred=pred, ld_pred=ldp, flow=flow, grad_scale=0.37, dpred=dp, ld_d=ldd, acc=acc)
    capi.call('tfx_mse_fwd_bwd', a, stream())
    assert abs(acc[0].item() - ((pred - flow) ** 2).sum().item()) < 1e-3 * acc[0].item()
    check('mse dpred', dp[:, :dl], (pred - flow) * 0.37, 4e-3)
    assert (dp[:, dl:] == 0).all()


def test_cast_batch_plain_and_transposed_jobs_in_one_launch():
    """`tfx_cast_batch`: every bf16 shadow of the parameter set in one launch - plain jobs (row map, column padding) and transposed jobs
    (dst[c][r] = src[map(r)][c]) on sources that start at odd float offsets of one flat buffer (16-byte loads only where a row is aligned)."""
    import ctypes
    torch.manual_seed(2)
    flat = torch.randn(400000, device=DEV)
    specs = [(3, 200, 70, True, True), (1003, 130, 64, False, True), (20000, 512, 136, True, False), (100001, 64, 200, False, False),
             (150002, 300, 129, True, True), (250000, 256, 256, False, True)]            # (offset, Rs, Cs, row map?, transposed?)
    J = capi.STRUCTS['tfx_cast_job']
    arr = (J * len(specs))()
    keep, first = [], 0
    for i, (off, Rs, Cs, use_map, tr) in enumerate(specs):
        src = flat[off:off + Rs * Cs].view(Rs, Cs)
        n_r = Rs + 24                                                    # logical rows: some map to nothing / past the source
        rowmap = torch.randint(-1, Rs + 5, (n_r,), device=DEV, dtype=torch.int32) if use_map else None
        if tr:
            Rd, ldd = (Cs + 15) // 8 * 8, (n_r + 7) // 8 * 8             # dst rows = padded Cs, dst columns = padded logical rows
            nb = ((ldd + 63) // 64) * ((Rd + 63) // 64)
        else:
            Rd, ldd = n_r, (Cs + 15) // 8 * 8
            nb = (Rd * ldd + 2047) // 2048
        dst = torch.full((Rd, ldd), float('nan'), device=DEV, dtype=BF)
        for k, v in dict(src=src.data_ptr(), ld_src=Cs, Rs=Rs, Cs=Cs, rowmap=capi.ptr(rowmap), dst=dst.data_ptr(), ld_dst=ldd, Rd=Rd,
                         Cd=n_r if tr else ldd, transposed=int(tr), first_block=first).items():
            setattr(arr[i], k, v)
        first += nb
        ref = torch.zeros(n_r, Cs, device=DEV)
        rows = rowmap.long() if use_map else torch.arange(n_r, device=DEV)
        ok = (rows >= 0) & (rows < Rs)
        ref[ok] = src[rows[ok]]
        full = torch.zeros(Rd, ldd, device=DEV)
        if tr:
            full[:Cs, :n_r] = ref.T
        else:
            full[:, :Cs] = ref
        keep.append((dst, full, rowmap, tr))
    tab = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8).to(DEV)
    capi.check(capi.lib().tfx_cast_batch(tab.data_ptr(), len(specs), first, stream()), 'tfx_cast_batch')
    for i, (dst, full, _, tr) in enumerate(keep):
        assert not torch.isnan(dst.float()).any(), f'job {i}: unwritten elements'
        assert torch.equal(dst, full.to(BF)), f'job {i} ({"transposed" if tr else "plain"})'


def test_param_plumbing():
    torch.manual_seed(0)
    Rs, Cs = 200, 70
    src = torch.randn(Rs, Cs, device=DEV)
    Rd, Cd, ldd = 256, 128, 128
    rowmap = torch.randint(-1, Rs, (Rd,), device=DEV, dtype=torch.int32)
    dst = torch.full((Rd, ldd), float('nan'), device=DEV, dtype=BF)
    a = capi.make_args('tfx_cast_args', src=src, ld_src=Cs, Rs=Rs, Cs=Cs, rowmap=rowmap, dst=dst, ld_dst=ldd, Rd=Rd, Cd=Cd)
    capi.call('tfx_cast_rows', a, stream())
    ref = torch.zeros(Rd, ldd, device=DEV)
    keep = rowmap >= 0
    ref[keep, :Cs] = src[rowmap[keep].long()]
    check('cast_rows', dst, ref, 4e-3)
    # transposed: dst[c][r] = src[map(r)][c]; dst rows = padded Cs (128), cols = padded #r (256)
    dstT = torch.full((128, 256), float('nan'), device=DEV, dtype=BF)
    a = capi.make_args('tfx_cast_args', src=src, ld_src=Cs, Rs=Rs, Cs=Cs, rowmap=rowmap, dst=dstT, ld_dst=256, Rd=128, Cd=Rd)
    capi.call('tfx_cast_rows_t', a, stream())
    check('cast_rows_t', dstT, ref[:, :128].T, 4e-3)
    # colsum
    X = rnd(1000, 136); out = torch.ones(136, device=DEV)
    capi.check(capi.lib().tfx_colsum_bf16(X.data_ptr(), 136, 1000, 130, None, None, out.data_ptr(), stream()), 'colsum')
    ref = torch.ones(136, device=DEV); ref[:130] += X.float()[:, :130].sum(0)
    check('colsum_bf16', out, ref, 1e-4)
    # adam + clip
    n = 100003
    p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    pr = p.clone().requires_grad_(True); opt = torch.optim.Adam([pr], lr=3e-4)
    ss = torch.zeros(1, device=DEV)
    for step in (1, 2, 3):
        g = torch.randn(n, device=DEV) * 0.01
        ss.zero_()
        capi.check(capi.lib().tfx_sumsq(g.data_ptr(), n, ss.data_ptr(), stream()), 'sumsq')
        a = capi.make_args('tfx_adam_args', p=p, g=g, m=m, v=v, n=n, lr=3e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                           max_norm=0.5, grad_scale=1.0, step=step, sumsq=ss)
        capi.call('tfx_adam_step', a, stream())
        pr.grad = g.clone(); torch.nn.utils.clip_grad_norm_([pr], 0.5); opt.step()
    check('adam+clip', p, pr.detach(), 1e-6)


@pytest.mark.parametrize('T,d', [(1000, 512), (300, 1024)])
def test_adaln_bwd_segment_mode(T, d):
    """segment mode (one wave owns an instance: register reduction + plain stores) == per-token atomics mode."""
    import numpy as np
    from transfusion_pytorch_amd.packing import token_segments
    torch.manual_seed(0)
    I = 23
    # contiguous instances of random lengths inside 2 sample rows
    tok = np.full((2, T // 2), -1, dtype=np.int32)
    g = 0
    for bi in range(2):
        pos = 2
        while pos < T // 2 - 12 and g < I:
            L = int(np.random.RandomState(g).randint(1, 12))
            tok[bi, pos:pos + L] = g; g += 1
            pos += L + int(np.random.RandomState(100 + g).randint(0, 20))
    seg_start, seg_len = token_segments(tok)
    assert seg_len.sum() == tok.size
    tok_inst = torch.from_numpy(tok.reshape(-1)).to(DEV)
    T = tok.size
    ld = 3 * d + 8
    table = torch.randn(I, ld, device=DEV) * 0.5
    x = rnd(T, d, scale=2.0); gt = torch.randn(d, device=DEV) * 0.3; du = rnd(T, d)
    u = torch.zeros(T, d, device=DEV, dtype=BF); mean = torch.zeros(T, device=DEV); rstd = torch.zeros(T, device=DEV)
    ss, sl = torch.from_numpy(seg_start).to(DEV), torch.from_numpy(seg_len).to(DEV)
    outs = []
    for seg in (False, True):
        dx = torch.zeros(T, d, device=DEV, dtype=BF); dtable = torch.zeros_like(table); dgt = torch.zeros(d, device=DEV)
        kw = dict(seg_start=ss, seg_len=sl, n_seg=len(seg_start)) if seg else {}
        a = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=x, u=u, tok_inst=tok_inst, table=table, ld_table=ld, gamma_text=gt,
                           mean=mean, rstd=rstd, du=du, dx=dx, dtable=dtable, dgamma_text=dgt, **kw)
        capi.call('tfx_adaln_pre_fwd', a, stream()); capi.call('tfx_adaln_pre_bwd', a, stream())
        y = rnd(T, d, scale=1.0) if not outs else y_keep
        y_keep = y
        ls = gt; gg = du; dy = torch.zeros(T, d, device=DEV, dtype=BF); dt2 = torch.zeros_like(table); dls = torch.zeros(d, device=DEV)
        a2 = capi.make_args('tfx_adaln_post_args', T=T, d=d, x=x, y=y, out=u, tok_inst=tok_inst, table=table, ld_table=ld,
                            layerscale=ls, g=gg, dy=dy, dtable=dt2, dlayerscale=dls, **kw)
        capi.call('tfx_adaln_post_bwd', a2, stream())
        outs.append((dx.float(), dtable[:, :2 * d].clone(), dgt.clone(), dy.float(), dt2[:, 2 * d:3 * d].clone(), dls.clone()))
    names = ['pre dx', 'pre dtable', 'pre dgamma_text', 'post dy', 'post dtable', 'post dlayerscale']
    for nm, a_, b_ in zip(names, outs[1], outs[0]):
        check(f'segment-mode {nm}', a_, b_, 2e-3 if 'dx' in nm or 'dy' in nm else 1e-5)


def _segmented_tokens(T, I, seed=0):
    """two sample rows with contiguous instances of random lengths; returns tok_inst (device), segment arrays (device), T"""
    import numpy as np
    from transfusion_pytorch_amd.packing import token_segments
    tok = np.full((2, T // 2), -1, dtype=np.int32)
    g = 0
    for bi in range(2):
        pos = 2
        while pos < T // 2 - 12 and g < I:
            L = int(np.random.RandomState(seed + g).randint(1, 12))
            tok[bi, pos:pos + L] = g; g += 1
            pos += L + int(np.random.RandomState(seed + 100 + g).randint(0, 20))
    seg_start, seg_len = token_segments(tok)
    return torch.from_numpy(tok.reshape(-1)).to(DEV), torch.from_numpy(seg_start).to(DEV), torch.from_numpy(seg_len).to(DEV), tok.size


@pytest.mark.parametrize('T,d', [(1000, 512), (300, 1024), (200, 64)])
def test_adaln_pre_post_bwd_fused_equals_two_launches(T, d):
    """tfx_adaln_pre_post_bwd == tfx_adaln_pre_bwd followed by tfx_adaln_post_bwd (segment mode): rows and stored table gradients bit for bit,
    atomically accumulated vectors to summation order; `dx_add` of the input-side kernel in both modes."""
    torch.manual_seed(0)
    I = 23
    tok_inst, ss, sl, T = _segmented_tokens(T, I)
    ld = 6 * d + 8
    table = torch.randn(I, ld, device=DEV) * 0.5
    x = rnd(T, d, scale=2.0); gt = torch.randn(d, device=DEV) * 0.3; du = rnd(T, d); y = rnd(T, d); ls = torch.randn(d, device=DEV) * 0.3
    u = torch.zeros(T, d, device=DEV, dtype=BF); mean = torch.zeros(T, device=DEV); rstd = torch.zeros(T, device=DEV)
    dx0 = rnd(T, d)
    seg = dict(seg_start=ss, seg_len=sl, n_seg=int(ss.numel()))
    outs = []
    for fused in (False, True):
        dx = dx0.clone(); dtable = torch.zeros_like(table); dgt = torch.zeros(d, device=DEV)
        dy = torch.zeros(T, d, device=DEV, dtype=BF); dls = torch.zeros(d, device=DEV)
        a = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=x, u=u, tok_inst=tok_inst, table=table, ld_table=ld, gamma_text=gt,
                           mean=mean, rstd=rstd, du=du, dx=dx, dtable=dtable, dgamma_text=dgt, **seg)
        b = capi.make_args('tfx_adaln_post_args', T=T, d=d, y=y, tok_inst=tok_inst, table=table.data_ptr() + 4 * 3 * d, ld_table=ld,
                           layerscale=ls, g=dx, dy=dy, dtable=dtable.data_ptr() + 4 * 3 * d, dlayerscale=dls, **seg)
        capi.call('tfx_adaln_pre_fwd', a, stream())
        if fused:
            import ctypes
            capi.check(capi.lib().tfx_adaln_pre_post_bwd(ctypes.byref(a), ctypes.byref(b), stream()), 'tfx_adaln_pre_post_bwd')
        else:
            capi.call('tfx_adaln_pre_bwd', a, stream()); capi.call('tfx_adaln_post_bwd', b, stream())
        outs.append((dx.clone(), dy.clone(), dtable.clone(), dgt.clone(), dls.clone()))
    assert torch.equal(outs[0][0], outs[1][0]), 'dx differs'
    assert torch.equal(outs[0][1], outs[1][1]), 'dy differs'
    assert torch.equal(outs[0][2], outs[1][2]), 'dtable differs'
    check('fused dgamma_text', outs[1][3], outs[0][3], 1e-5); check('fused dlayerscale', outs[1][4], outs[0][4], 1e-5)
    # dx_add: one more addend of dx, segment mode and per-token mode
    add = rnd(T, d)
    for kw in (seg, {}):
        res = []
        for use_add in (False, True):
            dx = dx0.clone()
            dtable = torch.zeros_like(table); dgt = torch.zeros(d, device=DEV)
            a = capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=x, u=u, tok_inst=tok_inst, table=table, ld_table=ld, gamma_text=gt,
                               mean=mean, rstd=rstd, du=du, dx=dx, dtable=dtable, dgamma_text=dgt, dx_add=add if use_add else None, **kw)
            capi.call('tfx_adaln_pre_bwd', a, stream())
            res.append(dx.float())
        check('adaln_pre_bwd dx_add', res[1], res[0] + add.float(), 6e-3)


@pytest.mark.parametrize('T,d,D,segs,post', [(600, 512, 8, True, True), (300, 512, 8, False, True), (300, 1024, 6, True, True), (200, 512, 12, True, False),
                                             (200, 256, 3, True, True), (150, 768, 9, False, False)])
def test_attnres_pull_backward_stack(T, d, D, segs, post):
    """A stack of D AttentionResiduals (layer j mixes hiddens 0 .. j+1, T:807-829) differentiated in PULL form (tfx_attnres_pull_bwd: the gradient
    of hidden l gathered once from every layer that mixed it) against torch autograd of the same stack; the forward kernels supply the saved
    softmax state.  Covers the register (n_src <= 8, d <= 512) and the LDS accumulation forms, segments / one token per wave, and the fused
    output side of the feed-forward wrapper against a separate tfx_adaln_post_bwd on the same row."""
    import ctypes
    torch.manual_seed(1)
    I = 19
    tok_inst, ss, sl, T = _segmented_tokens(T, I, seed=3)
    seg = dict(seg_start=ss, seg_len=sl, n_seg=int(ss.numel())) if segs else {}
    H = rnd(D + 1, T, d, scale=2.0)
    gm = torch.randn(D, d, device=DEV) * 0.3; pq = torch.randn(D, d, device=DEV) * 0.5
    G = rnd(D, T, d)                                              # gradient wrt the output of layer j's AttentionResidual
    extra = rnd(T, d)                                             # direct gradient of hidden 0
    outs = torch.zeros(D, T, d, device=DEV, dtype=BF); errs = torch.zeros(D, T, d, device=DEV, dtype=BF)
    saves = [torch.zeros(T, j + 2, 4, device=DEV) for j in range(D)]
    for j in range(D):
        a = capi.make_args('tfx_attnres_args', T=T, d=d, L=j + 2, hiddens=H, stride_h=T * d, gamma=gm[j], pq=pq[j], out=outs[j], save=saves[j], err=errs[j])
        capi.call('tfx_attnres_fwd', a, stream())
    # reference
    Hr = H.float().requires_grad_(True); gr = gm.clone().requires_grad_(True); pr = pq.clone().requires_grad_(True)
    loss = (Hr[0] * extra.float()).sum()
    for j in range(D):
        hs = Hr[:j + 2]
        keys = F.normalize(hs, dim=-1) * d ** 0.5 * (gr[j] + 1)
        sim = torch.einsum('ltd,d->tl', keys, pr[j]) * d ** -0.5
        o = torch.einsum('tl,ltd->td', sim.softmax(-1), hs)
        if j == D - 1:
            check('attnres fwd (with save)', outs[j], o, 6e-3)
            aw = sim.softmax(-1)
            check('saved softmax weights', saves[j][:, :, 0], aw, 1e-4)
        loss = loss + (o * G[j].float()).sum()
    loss.backward()
    # native
    dsum = torch.zeros(D, T, device=DEV); wtab = torch.zeros(2, D, d, device=DEV)
    dgm = torch.zeros(D, d, device=DEV); dpq = torch.zeros(D, d, device=DEV)
    SRC = capi.STRUCTS['tfx_attnres_src']
    recs = (SRC * D)()
    for j in range(D):
        for k, v in dict(g=G[j], save=saves[j], dsum=dsum[j], w=wtab[0, j], dw=wtab[1, j], gamma=gm[j], pq=pq[j], dgamma=dgm[j], dpq=dpq[j]).items():
            setattr(recs[j], k, v.data_ptr())
        recs[j].L = j + 2
    tab = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(recs), ctypes.sizeof(recs))), dtype=torch.uint8).to(DEV)
    lib = capi.lib()
    capi.check(lib.tfx_attnres_prep(tab.data_ptr(), D, d, stream()), 'prep')
    dH = torch.full((D + 1, T, d), float('nan'), device=DEV, dtype=BF)
    ld = 3 * d + 8
    table = torch.randn(I, ld, device=DEV) * 0.5; ls = torch.randn(d, device=DEV) * 0.3; y = rnd(T, d)
    k1 = torch.full((T, 32), float('nan'), device=DEV, dtype=BF)
    for l in range(D, -1, -1):
        j0 = max(l - 1, 0)
        export = (D - j0) > 8 or d > 512          # d w = K1^T . h by one weight-gradient GEMM behind the launch (include/tfx.h tfx_attnres_pull_args.k1)
        a = capi.make_args('tfx_attnres_pull_args', T=T, d=d, l=l, n_src=D - j0, h=H[l], src=tab.data_ptr() + j0 * ctypes.sizeof(SRC),
                           out_own=outs[l - 1] if l >= 1 else None, out_err=errs[l - 1] if l >= 1 else None, add=extra if l == 0 else None, dh=dH[l],
                           k1=k1 if export else None, ld_k1=32, **seg)
        if post and l == D:
            dy = torch.zeros(T, d, device=DEV, dtype=BF); dt = torch.zeros_like(table); dls = torch.zeros(d, device=DEV); db = torch.zeros(d, device=DEV)
            b = capi.make_args('tfx_adaln_post_args', T=T, d=d, y=y, tok_inst=tok_inst, table=table, ld_table=ld, layerscale=ls, g=dH[l], dy=dy,
                               dtable=dt, dlayerscale=dls, dbias=db, **seg)
            capi.check(lib.tfx_attnres_pull_bwd(ctypes.byref(a), ctypes.byref(b), stream()), 'pull+post')
            dy2 = torch.zeros(T, d, device=DEV, dtype=BF); dt2 = torch.zeros_like(table); dls2 = torch.zeros(d, device=DEV); db2 = torch.zeros(d, device=DEV)
            b2 = capi.make_args('tfx_adaln_post_args', T=T, d=d, y=y, tok_inst=tok_inst, table=table, ld_table=ld, layerscale=ls, g=dH[l], dy=dy2,
                                dtable=dt2, dlayerscale=dls2, dbias=db2, **seg)
            capi.call('tfx_adaln_post_bwd', b2, stream())
            assert torch.equal(dy, dy2), 'fused output side: dy differs'
            check('fused output side dtable', dt, dt2, 1e-5); check('fused dlayerscale', dls, dls2, 1e-5); check('fused dbias', db, db2, 1e-5)
        else:
            capi.check(lib.tfx_attnres_pull_bwd(ctypes.byref(a), None, stream()), 'pull')
        if export:
            tn = capi.make_args('tfx_gemm_tn_args', A=k1, lda=32, a_cols=32, B=H[l], ldb=d, b_cols=d, M=T, N=D - j0, K=d, C=wtab[1, j0], ldc=d,
                                k_valid=d, splits=8, accumulate=1, alpha=1.0)
            capi.call('tfx_gemm_tn', tn, stream())
    capi.check(lib.tfx_attnres_finish(tab.data_ptr(), D, d, stream()), 'finish')
    torch.cuda.synchronize()
    for l in range(D + 1):
        check(f'pull dH[{l}] (D={D}, d={d})', dH[l], Hr.grad[l], 2e-2)
    check('pull dgamma', dgm, gr.grad, 2e-2); check('pull dpq', dpq, pr.grad, 2e-2)
    for j in (0, D - 1):
        check(f'pull dgamma layer {j}', dgm[j], gr.grad[j], 3e-2); check(f'pull dpq layer {j}', dpq[j], pr.grad[j], 3e-2)


# ---------------------------------------------------------------------------------------------- decode-side kernels
@pytest.mark.parametrize('B,V,ld', [(1, 390, 392), (7, 392, 392), (64, 390, 448), (5, 70, 72)])
def test_sample_tokens_greedy_and_min_p(B, V, ld):
    """tfx_sample_tokens vs the reference's sample_text_token / min_p_filter semantics (T:591-605) in torch: greedy = argmax (first index on
    ties); with a temperature the draw is the inverse CDF of softmax(logits / T) restricted to p >= min_p * p_max at the given uniform."""
    import ctypes
    torch.manual_seed(4)
    logits = torch.randn(B, ld, device=DEV) * 3.
    logits[:, V:] = 1e9                                            # pad columns must never be chosen
    if B > 1:
        logits[1, 5] = logits[1, 9] = logits[1, :V].max() + 1.      # a tie: the first index wins
    out = torch.full((B,), -7, dtype=torch.int32, device=DEV)
    lib = capi.lib()
    capi.check(lib.tfx_sample_tokens(logits.data_ptr(), ld, B, V, 0., 0.1, None, None, out.data_ptr(), ctypes.c_void_p(stream())), 'greedy')
    assert torch.equal(out.long().cpu(), logits[:, :V].argmax(-1).cpu())
    for T, min_p in ((1.0, 0.1), (0.7, 0.3), (2.0, 0.0)):
        u = torch.rand(B, device=DEV)
        capi.check(lib.tfx_sample_tokens(logits.data_ptr(), ld, B, V, T, min_p, u.data_ptr(), None, out.data_ptr(), ctypes.c_void_p(stream())), 'min-p')
        p = (logits[:, :V].double() / T).softmax(-1)
        keep = p >= min_p * p.amax(-1, keepdim=True)
        q = torch.where(keep, p, torch.zeros_like(p))
        cdf = q.cumsum(-1)
        target = u.double()[:, None] * cdf[:, -1:]
        ref = (cdf > target).float().argmax(-1)
        got = out.long()
        assert bool(keep.gather(1, got[:, None]).all()), 'a filtered token was drawn'
        # fp32 vs fp64 prefix sums may disagree exactly at a CDF boundary: accept the neighbouring survivor there
        bad = (got != ref).nonzero().flatten().tolist()
        for r in bad:
            lo, hi = sorted((int(got[r]), int(ref[r])))
            assert float(q[r, lo + 1:hi].sum()) == 0. and abs(float(cdf[r, lo] - target[r])) < 1e-5 * float(cdf[r, -1]), (r, int(got[r]), int(ref[r]))
    # active mask: untouched rows keep their value
    act = torch.zeros(B, dtype=torch.int32, device=DEV); act[0] = 1
    out.fill_(-7)
    capi.check(lib.tfx_sample_tokens(logits.data_ptr(), ld, B, V, 0., 0.1, None, act.data_ptr(), out.data_ptr(), ctypes.c_void_p(stream())), 'active')
    assert int(out[0]) == int(logits[0, :V].argmax()) and bool((out[1:] == -7).all())


def test_ode_axpy_with_guidance():
    import ctypes
    torch.manual_seed(6)
    y, fc, fu = (torch.randn(5, 7, 33, device=DEV) for _ in range(3))
    out = torch.empty_like(y)
    lib = capi.lib()
    capi.check(lib.tfx_ode_axpy(y.data_ptr(), fc.data_ptr(), fu.data_ptr(), 3., 0.25, out.data_ptr(), y.numel(), ctypes.c_void_p(stream())), 'cfg')
    assert torch.allclose(out, y + 0.25 * (fu + 3. * (fc - fu)), rtol=1e-6, atol=1e-6)
    capi.check(lib.tfx_ode_axpy(y.data_ptr(), fc.data_ptr(), None, 3., -0.5, out.data_ptr(), y.numel(), ctypes.c_void_p(stream())), 'plain')
    assert torch.allclose(out, y - 0.5 * fc, rtol=1e-6, atol=1e-6)


def test_allreduce_entry_points_world1():
    """K12: tfx_allreduce_unique_id / _init / _run / _destroy on a real RCCL communicator of ONE rank (the box has one GPU; the
    N > 1 exchange is covered on gloo, tests/test_dp_gloo.py, and by the driver's scaling run): the sum over one rank is the identity."""
    import ctypes
    lib = capi.lib()
    uid = (ctypes.c_char * 128)()
    capi.check(lib.tfx_allreduce_unique_id(ctypes.byref(uid)), 'unique_id')
    capi.check(lib.tfx_allreduce_init(0, 1, ctypes.byref(uid)), 'init')
    try:
        g = torch.randn(1 << 20, device=DEV)
        want = g.clone()
        capi.check(lib.tfx_allreduce_run(g.data_ptr(), g.numel(), ctypes.c_void_p(stream())), 'run')
        capi.check(lib.tfx_allreduce_run(g[1000:].data_ptr(), 4096, ctypes.c_void_p(stream())), 'run (range)')
        torch.cuda.synchronize()
        assert torch.equal(g, want)
    finally:
        capi.check(lib.tfx_allreduce_destroy(), 'destroy')
    assert lib.tfx_allreduce_run(g.data_ptr(), 16, ctypes.c_void_p(stream())) != 0        # no communicator: refused


@pytest.mark.parametrize('T,d,L', [(300, 512, 3), (130, 1024, 9)])
def test_decode_fused_launches_equal_the_separate_kernels(T, d, L):
    """decode plans fuse launches (a decode step is launch-bound): `tfx_adaln_post_pre_fwd` = adaln_post + adaln_pre, `tfx_layer_end_fwd` =
    adaln_post + AttentionResidual + adaln_pre, `tfx_qk_norm_rope_fwd` with `cache` = norm / rope + the two KV-cache scatters.  Every stored row is
    rounded to bf16 before it is used again, so the fused results must equal the separate kernels' BIT FOR BIT."""
    import ctypes
    torch.manual_seed(0)
    I = 7
    inst = torch.randint(-1, I, (T,), device=DEV, dtype=torch.int32)
    table = torch.randn(I, 6 * d, device=DEV) * 0.5
    ls, gt = torch.randn(d, device=DEV) * 0.1, torch.randn(d, device=DEV) * 0.1
    x, y = rnd(T, d), rnd(T, d)
    mk = lambda: (torch.zeros(T, d, device=DEV, dtype=BF), torch.zeros(T, d, device=DEV, dtype=BF), torch.zeros(T, device=DEV), torch.zeros(T, device=DEV))
    # ---- post + pre
    out_a, u_a, mean_a, rstd_a = mk(); out_b, u_b, mean_b, rstd_b = mk()
    post = lambda out: capi.make_args('tfx_adaln_post_args', T=T, d=d, x=x, y=y, out=out, tok_inst=inst, table=table, ld_table=6 * d, layerscale=ls)
    pre = lambda xin, u, mean, rstd: capi.make_args('tfx_adaln_pre_args', T=T, d=d, x=xin, u=u, tok_inst=inst, table=table.data_ptr() + 4 * 3 * d, ld_table=6 * d,
                                                    gamma_text=gt, mean=mean, rstd=rstd)
    capi.call('tfx_adaln_post_fwd', post(out_a), stream()); capi.call('tfx_adaln_pre_fwd', pre(out_a, u_a, mean_a, rstd_a), stream())
    pa, pb = post(out_b), pre(out_b, u_b, mean_b, rstd_b)
    assert capi.lib().tfx_adaln_post_pre_fwd(ctypes.byref(pa), ctypes.byref(pb), ctypes.c_void_p(stream())) == 0
    assert torch.equal(out_a, out_b) and torch.equal(u_a, u_b) and torch.equal(mean_a, mean_b) and torch.equal(rstd_a, rstd_b)
    # ---- layer end: post -> hidden L-1, AttentionResidual over L hiddens, pre of the next layer
    Ha, Hb = rnd(L, T, d, scale=2.0), None
    Hb = Ha.clone()
    gam, pq = torch.randn(d, device=DEV) * 0.3, torch.randn(d, device=DEV) * 0.5
    res_a, u_a, mean_a, rstd_a = mk(); res_b, u_b, mean_b, rstd_b = mk()
    post2 = lambda H: capi.make_args('tfx_adaln_post_args', T=T, d=d, x=x, y=y, out=H[L - 1], tok_inst=inst, table=table, ld_table=6 * d, layerscale=ls)
    ar = lambda H, out: capi.make_args('tfx_attnres_args', T=T, d=d, L=L, hiddens=H, stride_h=T * d, gamma=gam, pq=pq, out=out)
    capi.call('tfx_adaln_post_fwd', post2(Ha), stream()); capi.call('tfx_attnres_fwd', ar(Ha, res_a), stream())
    capi.call('tfx_adaln_pre_fwd', pre(res_a, u_a, mean_a, rstd_a), stream())
    a1, a2, a3 = post2(Hb), ar(Hb, res_b), pre(res_b, u_b, mean_b, rstd_b)
    assert capi.lib().tfx_layer_end_fwd(ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3), ctypes.c_void_p(stream())) == 0
    assert torch.equal(Ha, Hb) and torch.equal(res_a, res_b) and torch.equal(u_a, u_b) and torch.equal(mean_a, mean_b)
    res_c = torch.zeros_like(res_b); Hc = Ha.clone()
    a1, a2 = post2(Hc), ar(Hc, res_c)
    assert capi.lib().tfx_layer_end_fwd(ctypes.byref(a1), ctypes.byref(a2), None, ctypes.c_void_p(stream())) == 0 and torch.equal(res_c, res_a)
    bad = ar(Hc, res_c); bad.L = L - 1                         # post->out is not the last hidden of the mix: refused
    assert capi.lib().tfx_layer_end_fwd(ctypes.byref(a1), ctypes.byref(bad), None, ctypes.c_void_p(stream())) == -3
    # ---- qk norm + rope with the KV-cache append
    H = 4; HD = H * 64; ld = 3 * HD + 8
    qkv = rnd(T, ld, scale=1.5)
    gq, gk = torch.randn(64, device=DEV) * 0.3, torch.randn(64, device=DEV) * 0.3
    pos = torch.randint(0, 900, (T,), device=DEV, dtype=torch.int32)
    freqs = 1. / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(1024).float()[:, None] * freqs[None]
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    cpos = torch.randperm(2 * T, device=DEV)[:T].to(torch.int32); cpos[::11] = -1
    qk_a, qk_b = torch.zeros(T, 2 * HD, device=DEV, dtype=BF), torch.zeros(T, 2 * HD, device=DEV, dtype=BF)
    cache_a, cache_b = torch.zeros(2 * T, 2 * HD, device=DEV, dtype=BF), torch.zeros(2 * T, 2 * HD, device=DEV, dtype=BF)
    base = dict(T=T, H=H, qkv=qkv, ld_qkv=ld, ld_qk=2 * HD, gamma_q=gq, gamma_k=gk, rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t, q_scale=0.125)
    capi.call('tfx_qk_norm_rope_fwd', capi.make_args('tfx_qk_norm_rope_args', qk=qk_a, **base), stream())
    sp = ctypes.c_void_p(stream())
    capi.check(capi.lib().tfx_scatter_rows_bf16(qk_a.data_ptr() + 2 * HD, 2 * HD, HD, cache_a.data_ptr(), 2 * HD, cpos.data_ptr(), T, sp), 'scatter k')
    capi.check(capi.lib().tfx_scatter_rows_bf16(qkv.data_ptr() + 2 * 2 * HD, ld, HD, cache_a.data_ptr() + 2 * HD, 2 * HD, cpos.data_ptr(), T, sp), 'scatter v')
    capi.call('tfx_qk_norm_rope_fwd', capi.make_args('tfx_qk_norm_rope_args', qk=qk_b, cache=cache_b, ld_cache=2 * HD, cache_pos=cpos, **base), stream())
    assert torch.equal(qk_a, qk_b) and torch.equal(cache_a, cache_b)
