"""Child process of test_kernels_gpu.py::test_gemm_nt_one_wave_kernels_bit_identical_to_ping_pong: runs tfx_gemm_nt on fixed seeded operands under the
TFX_NT_OW mode of its environment (the library reads the switch once per process) and prints one sha256 per case."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi  # noqa: E402

DEV, BF = 'cuda', torch.bfloat16


def h(t):
    return hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:24]


def main():
    st = torch.cuda.current_stream().cuda_stream
    E = capi.ENUMS
    # (M, N, K, epilogue, bias, row map, lda pad): one / two / three / four K-tiles (every entry path of the K loops), ragged M and N tiles, a padded lda,
    # the plain and the side-load form of the staged store, scattered output rows, N % 8 != 0 (the direct, unstaged store), fp32 / residual / GEGLU epilogues, a long K
    cases = [(2304, 512, 64, 'BF16', False, False, 0), (2304, 512, 128, 'BF16', False, False, 0), (2100, 520, 192, 'BF16', True, False, 0),
             (2304, 768, 256, 'BF16', False, False, 64), (33000, 1032, 320, 'BF16', False, True, 0), (33000, 1036, 320, 'BF16', True, False, 0), (65536, 512, 512, 'BF16', False, False, 0),
             (65536, 1544, 512, 'BF16', True, False, 0), (8192, 1024, 2752, 'BF16', False, False, 0), (4096, 512, 1408, 'F32', True, False, 0),
             (8192, 512, 512, 'RESID', True, False, 0), (8192, 2816, 512, 'GEGLU', True, False, 0), (8192, 1408, 512, 'GEGLU_BWD', False, False, 0)]
    for (M, N, K, epi, bias, rmap, pad) in cases:
        g = torch.Generator(device=DEV); g.manual_seed(M + 7 * N + 13 * K)
        lda = K + pad
        A = (torch.randn(M, lda, device=DEV, generator=g)).to(BF); B = (torch.randn(N, K, device=DEV, generator=g) * K ** -0.5).to(BF)
        kw = dict(A=A, lda=lda, B=B, ldb=K, M=M, N=N, K=K, epi=E['TFX_EPI_' + epi])
        ldc = 2 * N if epi == 'GEGLU_BWD' else N
        C = torch.full((M + 8, ldc), 3.0, device=DEV, dtype=torch.float32 if epi == 'F32' else BF)
        kw.update(C=C, ldc=ldc)
        keep = [A, B, C]
        if bias: b = torch.randn(N, device=DEV, generator=g); kw['bias'] = b; keep.append(b)
        if rmap:
            rm = torch.randperm(M + 8, device=DEV, generator=g)[:M].to(torch.int32); rm[5] = -1; kw['rowmap'] = rm; keep.append(rm)
        C2 = None
        if epi == 'RESID': R = torch.randn(M, N, device=DEV, generator=g).to(BF); kw.update(R=R, ldr=N); keep.append(R)
        if epi == 'GEGLU': C2 = torch.full((M, N // 2), 3.0, device=DEV, dtype=BF); kw.update(C2=C2, ldc2=N // 2)
        if epi == 'GEGLU_BWD': X = torch.randn(M, 2 * N, device=DEV, generator=g).to(BF); kw.update(aux=X, ldaux=2 * N); keep.append(X)
        a = capi.make_args('tfx_gemm_nt_args', **kw)
        rc = capi.call('tfx_gemm_nt', a, st)
        torch.cuda.synchronize()
        print(f'CASE {M}x{N}x{K} {epi} bias={int(bias)} rowmap={int(rmap)} {h(C)} {h(C2) if C2 is not None else "-"}', flush=True)


if __name__ == '__main__':
    main()
