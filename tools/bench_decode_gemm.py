"""Decode-step NT GEMM shapes (dim 1024 / depth 24 model, 64 samples: M = 64 text step, 256 modality step) through the C ABI.
    TFX_NT_SPLITK=0 python tools/bench_decode_gemm.py   vs   python tools/bench_decode_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
from bench_gemm import timeit, st, dev, BF
for M in ([int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else (64, 128, 640)):
    for (N, K, epi) in [(1544, 1024, 'TFX_EPI_BF16'), (1024, 512, 'TFX_EPI_BF16'), (5504, 1024, 'TFX_EPI_GEGLU'), (1024, 2752, 'TFX_EPI_BF16'), (1024, 2048, 'TFX_EPI_RESID'), (448, 1024, 'TFX_EPI_F32')]:
        A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        C = torch.empty(M, N, device=dev, dtype=torch.float32 if epi.endswith('F32') else BF)
        kw = dict(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS[epi], C=C, ldc=N)
        if epi.endswith('GEGLU'):
            kw.update(C2=torch.empty(M, N // 2, device=dev, dtype=BF), ldc2=N // 2, bias=torch.zeros(N, device=dev))
        if epi.endswith('RESID'):
            kw.update(R=torch.zeros(M, N, device=dev, dtype=BF), ldr=N)
        a = capi.make_args('tfx_gemm_nt_args', **kw)
        t = timeit(lambda: capi.call('tfx_gemm_nt', a, st()), n=50)
        print(f'M={M:4d} N={N:5d} K={K:5d} {epi[8:]:6s}: {t * 1e6:7.1f} us   weights {N * K * 2 / t / 1e9:7.0f} GB/s')
