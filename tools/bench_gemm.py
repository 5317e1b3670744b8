"""GEMM microbenchmark through the C ABI (random bf16 operands, HIP-event timing).
    python tools/bench_gemm.py            # the training-step shapes of the dim512/d8, batch 64 x 1024 workload"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi

dev = 'cuda'
BF = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def nt(M, N, K, epi='TFX_EPI_BF16', pad=0, padc=0):
    """pad / padc: extra elements in the row stride of A,B / of C (channel-camping experiment)"""
    A = torch.randn(M, K + pad, device=dev).to(BF); B = (torch.randn(N, K + pad, device=dev) * K ** -0.5).to(BF)
    C = torch.empty(M, N + padc, device=dev, dtype=BF)
    a = capi.make_args('tfx_gemm_nt_args', A=A, lda=K + pad, B=B, ldb=K + pad, M=M, N=N, K=K, epi=capi.ENUMS[epi], C=C, ldc=N + padc)
    t = timeit(lambda: capi.call('tfx_gemm_nt', a, st()))
    print(f'NT {M}x{N}x{K} pad={pad},{padc}: {t * 1e6:8.1f} us  {2 * M * N * K / t / 1e12:7.1f} TFLOP/s')


def tn(M, N, K, splits):
    A = torch.randn(M, N, device=dev).to(BF); B = torch.randn(M, K, device=dev).to(BF)
    C = torch.zeros(N, K, device=dev)
    a = capi.make_args('tfx_gemm_tn_args', A=A, lda=N, a_cols=N, B=B, ldb=K, b_cols=K, M=M, N=N, K=K, C=C, ldc=K, k_valid=K,
                       splits=splits, accumulate=1, alpha=1.0)
    t = timeit(lambda: capi.call('tfx_gemm_tn', a, st()))
    print(f'TN {M}: {N}x{K} splits={splits:3d}: {t * 1e6:8.1f} us  {2 * M * N * K / t / 1e12:7.1f} TFLOP/s')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'pad':
    T = 65536
    for (N, K) in [(512, 512), (1544, 512), (512, 1408), (2816, 512)]:
        for pad, padc in ((0, 0), (64, 0), (0, 64), (64, 64), (8, 8)):
            nt(T, N, K, pad=pad, padc=padc)
    for pad in (0, 64): nt(8192, 8192, 8192, pad=pad, padc=pad)
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'tn':      # weight-gradient shapes: library-chosen splits (0) and a sweep
    T = 65536
    for (N, K) in [(2816, 512), (1544, 512), (512, 1408), (512, 512), (1024, 1024), (4096, 1024)]:
        for s in (0, 8, 16, 32, 64):
            tn(T, N, K, s)
    sys.exit(0)

if __name__ == '__main__':
    T = 65536
    for (N, K) in [(1544, 512), (512, 512), (2816, 512), (512, 1408), (512, 1600), (512, 2816), (1408, 512)]:
        nt(T, N, K)
    for (N, K) in [(2816, 512), (1544, 512), (512, 1408), (512, 512)]:
        for s in (8, 16, 24, 32, 64):
            tn(T, N, K, s)
    nt(8192, 8192, 8192); nt(4096, 4096, 4096)
