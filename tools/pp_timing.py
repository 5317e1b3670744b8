"""Per-block phase timing of the ping-pong NT kernel (needs a library built with TFX_HIPCC_EXTRA=-DTFX_PP_TIMING).
   stamps: 0 block start, 1 first K-tile landed, 2 K loop done, 3 epilogue stores issued, 4 stores retired."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
dev = 'cuda'; BF = torch.bfloat16
for (M, N, K) in [(65536, 512, 512), (65536, 512, 2816)]:
    A = torch.randn(M, K, device=dev).to(BF); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    C = torch.empty(M, N, device=dev, dtype=BF)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    st = torch.zeros(tiles, 8, device=dev, dtype=torch.int64)
    a = capi.make_args('tfx_gemm_nt_args', A=A, lda=K, B=B, ldb=K, M=M, N=N, K=K, epi=capi.ENUMS['TFX_EPI_BF16'], C=C, ldc=N, aux=st.data_ptr())
    for _ in range(3):
        capi.call('tfx_gemm_nt', a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); capi.call('tfx_gemm_nt', a, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    print(f'   event time of this launch: {e0.elapsed_time(e1) * 1e3:.1f} us')
    s = st.cpu().double()
    t0 = s[:, 0].min()
    d = lambda i, j: (s[:, j] - s[:, i])
    print(f'{M}x{N}x{K}: tiles {tiles}  kernel span {float(s[:, 4].max() - t0):.0f} ticks')
    for name, x in (('prologue 0->1', d(0, 1)), ('k-loop 1->2', d(1, 2)), ('epilogue issue 2->3', d(2, 3)), ('store drain 3->4', d(3, 4)), ('block total 0->4', d(0, 4)), ('epi setup 2->5', d(2, 5)), ('epi block0 5->6', d(5, 6)), ('epi block1 6->7', d(6, 7)), ('block start offset', s[:, 0] - t0)):
        print(f'   {name:22s} median {float(x.median()):9.0f}  p10 {float(x.quantile(0.1)):9.0f}  p90 {float(x.quantile(0.9)):9.0f}  max {float(x.max()):9.0f}')
