"""Library yardstick for the NT GEMM shapes of the training step: torch.matmul (hipBLASLt / rocBLAS under PyTorch-ROCm) against tfx_gemm_nt on the
same random bf16 operands.  NOT part of the product (no library GEMM is linked into libtfx_hip.so): it tells how far the hand-written kernels are
from what AMD's tuned library reaches at these shapes.   python tools/bench_gemm_lib.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
from bench_gemm import timeit, st, dev, BF

T = 65536
# config 2 (dim 512): every in-step NT shape (forward projections and the dX products); config 3 (dim 1024): the same set; one square yardstick
NT_SHAPES = [(1544, 512), (512, 512), (2816, 512), (1408, 512), (512, 1408), (512, 2816), (512, 1024),
             (1544, 1024), (1024, 512), (5504, 1024), (1024, 2752), (2752, 1024), (1024, 5504), (1024, 2048), (1024, 1024), (4096, 4096)]
for (N, K) in NT_SHAPES:
    M = T if N * K < 4096 * 4096 else 8192
    A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    C = torch.empty(M, N, device=dev, dtype=BF)
    a = capi.make_args('tfx_gemm_nt_args', A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=N)
    t_own = timeit(lambda: capi.call('tfx_gemm_nt', a, st()))
    Bt = B.t()
    t_lib = timeit(lambda: torch.matmul(A, Bt, out=C))
    fl = 2 * M * N * K
    print(f'NT {M}x{N}x{K}: tfx {t_own * 1e6:8.1f} us {fl / t_own / 1e12:7.1f} TF/s | torch.matmul {t_lib * 1e6:8.1f} us {fl / t_lib / 1e12:7.1f} TF/s | tfx / lib = {t_lib / t_own:.2f}x')

# weight-gradient shapes: C[N, K] = A[M, N]^T B[M, K], contraction over the M = 65536 tokens (tfx_gemm_tn: split-M + fp32 atomics)
for (N, K) in [(512, 512), (2816, 512), (512, 1408), (1544, 512), (512, 1024), (1024, 1024), (5504, 1024), (1024, 2752), (1544, 1024), (1024, 4096)]:
    M = T
    A = torch.randn(M, N, device=dev).to(BF); B = torch.randn(M, K, device=dev).to(BF)
    C = torch.zeros(N, K, device=dev)
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    best = None
    for splits in ((1,) if tiles >= 512 else (0,)):                     # 0 = the library's own split rule (what the step runs)
        a = capi.make_args('tfx_gemm_tn_args', A=A, lda=N, a_cols=N, B=B, ldb=K, b_cols=K, M=M, N=N, K=K, C=C, ldc=K, k_valid=K, splits=splits, accumulate=1, alpha=1.0)
        t = timeit(lambda: capi.call('tfx_gemm_tn', a, st()))
        best = t if best is None else min(best, t)
    At = A.t(); Cb = torch.empty(N, K, device=dev, dtype=BF)
    t_lib = timeit(lambda: torch.matmul(At, B, out=Cb))
    fl = 2 * M * N * K
    print(f'TN {M}: {N}x{K}: tfx {best * 1e6:8.1f} us {fl / best / 1e12:7.1f} TF/s | torch.matmul {t_lib * 1e6:8.1f} us {fl / t_lib / 1e12:7.1f} TF/s | tfx / lib = {t_lib / best:.2f}x')
