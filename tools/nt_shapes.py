"""time a few NT shapes through the C ABI (A/B of builds: TFX_LIB=<path> python tools/nt_shapes.py)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from transfusion_pytorch_amd import capi
from bench_gemm import timeit, st, dev, BF
for (M, N, K) in [(65536, 512, 2816), (8192, 4096, 4096), (65536, 512, 512), (65536, 2816, 512), (65536, 512, 1408)]:
    A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF); C = torch.empty(M, N, device=dev, dtype=BF)
    a = capi.make_args('tfx_gemm_nt_args', A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=N)
    t = timeit(lambda: capi.call('tfx_gemm_nt', a, st()))
    print(f'{M}x{N}x{K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF/s')
