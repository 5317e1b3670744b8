"""Decode-step NT GEMMs with COLD weights (a ring of weight buffers larger than the 256 MB Infinity Cache, as in a real decode forward: 24 layers x 25 MB)
against the same launches with the weights touched just before (what a prefetch by spare blocks of the PREVIOUS launch would give).
    python tools/decode_cold_probe.py [M,M,...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
from bench_gemm import st, dev, BF


def loop_time(fns, n):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in ([int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else (64, 256)):
    for (N, K, epi) in [(1024, 1024, 'TFX_EPI_BF16'), (3088, 1024, 'TFX_EPI_BF16'), (5632, 1024, 'TFX_EPI_GEGLU'), (1024, 2816, 'TFX_EPI_BF16'), (1024, 2048, 'TFX_EPI_RESID')]:
        nb = max(8, int(520e6 / (N * K * 2)))
        Bs = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(nb)]
        A = torch.randn(M, K, device=dev).to(BF)
        C = torch.empty(M, N, device=dev, dtype=BF)
        kw = dict(A=A, lda=K, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS[epi], C=C, ldc=N)
        if epi.endswith('GEGLU'):
            kw.update(C2=torch.empty(M, N // 2, device=dev, dtype=BF), ldc2=N // 2, bias=torch.zeros(N, device=dev))
        if epi.endswith('RESID'):
            kw.update(R=torch.zeros(M, N, device=dev, dtype=BF), ldr=N)
        args = [capi.make_args('tfx_gemm_nt_args', B=b, **kw) for b in Bs]
        s = st()
        sink = torch.zeros(1, device=dev)
        warm = loop_time([lambda a=args[0]: capi.call('tfx_gemm_nt', a, s)], 200)
        cold = loop_time([lambda a=a: capi.call('tfx_gemm_nt', a, s) for a in args], 2 * nb)
        touch = loop_time([lambda b=b: b.view(torch.int32).bitwise_or_(0) for b in Bs], 2 * nb)          # reads (and rewrites) every byte: stands in for the prefetch
        both = loop_time([lambda a=a, b=b: (b.view(torch.int32).bitwise_or_(0), capi.call('tfx_gemm_nt', a, s)) for a, b in zip(args, Bs)], 2 * nb)
        print(f'M={M:4d} N={N:5d} K={K:5d} {epi[8:]:6s}: warm {warm:6.1f} us   cold {cold:6.1f} us   touch {touch:6.1f}   touch+gemm {both:6.1f}  -> gemm behind a touch {both - touch:6.1f} us', flush=True)
        del Bs, args
        torch.cuda.empty_cache()
