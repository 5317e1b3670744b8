"""Per-block phase timing of the ping-pong NT kernel (needs a library built with -DTFX_PP_TIMING: tools/build_variant.sh timing WORK -DTFX_PP_TIMING,
   then TFX_LIB=.../libtfx_timing.so python tools/pp_timing.py).
   stamps: 0 block start, 1 first K-tile landed, 2 K loop done, 3 epilogue stores issued, 4 stores retired.
   Shapes: the plain qkvg projection, the GEGLU forward and the GEGLU backward of the training step (round 4: where do their epilogues' cycles go)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
dev = 'cuda'; BF = torch.bfloat16
E = capi.ENUMS
M = 65536
for name, N, K, epi in [('plain 1544x512', 1544, 512, 'TFX_EPI_BF16'), ('plain 512x512', 512, 512, 'TFX_EPI_BF16'), ('plain 512x2816', 512, 2816, 'TFX_EPI_BF16'),
                        ('GEGLU fwd 2816x512', 2816, 512, 'TFX_EPI_GEGLU'), ('GEGLU bwd 1408x512', 1408, 512, 'TFX_EPI_GEGLU_BWD')]:
    A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros(tiles, 8, device=dev, dtype=torch.int64)
    kw = dict(A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=E[epi])
    if epi == 'TFX_EPI_BF16':
        C = torch.empty(M, N, device=dev, dtype=BF); kw.update(C=C, ldc=N, aux=st.data_ptr())
    elif epi == 'TFX_EPI_GEGLU':
        C = torch.empty(M, N, device=dev, dtype=BF); C2 = torch.empty(M, N // 2, device=dev, dtype=BF)
        kw.update(C=C, ldc=N, C2=C2, ldc2=N // 2, bias=torch.randn(N, device=dev), aux=st.data_ptr())
    else:
        C = torch.empty(M, 2 * N, device=dev, dtype=BF); ag = torch.randn(M, 2 * N, device=dev).to(BF)
        kw.update(C=C, ldc=2 * N, aux=ag, ldaux=2 * N, R=st.data_ptr())
    a = capi.make_args('tfx_gemm_nt_args', **kw)
    for _ in range(3):
        capi.call('tfx_gemm_nt', a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); capi.call('tfx_gemm_nt', a, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    s = st.cpu().double()
    t0 = s[:, 0].min()
    span = float(s[:, 4].max() - t0)
    d = lambda i, j: (s[:, j] - s[:, i])
    print(f'{name}: {tiles} tiles, event time {us:.1f} us, kernel span {span:.0f} ticks = {span / us / 1e3:.2f} GHz, {2.0 * M * N * K / us / 1e6:.0f} TF/s')
    for nm, x in (('prologue 0->1', d(0, 1)), ('k-loop 1->2', d(1, 2)), ('epilogue issue 2->3', d(2, 3)), ('store drain 3->4', d(3, 4)), ('block total 0->4', d(0, 4))):
        print(f'   {nm:22s} median {float(x.median()):9.0f}  p10 {float(x.quantile(0.1)):9.0f}  p90 {float(x.quantile(0.9)):9.0f}  max {float(x.max()):9.0f}')
