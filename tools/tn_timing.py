"""Per-block phase timing of the wide TN kernels (needs a library built with -DTFX_TN_TIMING: tools/build_variant.sh tnt WORK -DTFX_TN_TIMING,
   run with TFX_LIB=.../libtfx_tnt.so).  stamps (wave 0 of every block): 0 block start, 1 first slab landed, 2 K loop done, 3 atomics issued,
   4 atomics retired, 5 / 6 / 7 = shader clocks summed over the loop in [s_waitcnt + barrier] / [LDS-DMA issue] / [fragment reads + MFMA issue]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
dev = 'cuda'; BF = torch.bfloat16
M = 65536
for (N, K, splits) in [(2816, 512, 0), (1544, 512, 0), (512, 512, 0), (4096, 1024, 0)]:
    A = torch.randn(M, N, device=dev).to(BF); B = torch.randn(M, K, device=dev).to(BF)
    C = torch.zeros(N, K, device=dev)
    st = torch.zeros(8192, 8, device=dev, dtype=torch.int64)
    a = capi.make_args('tfx_gemm_tn_args', A=A, lda=N, a_cols=N, B=B, ldb=K, b_cols=K, M=M, N=N, K=K, C=C, ldc=K, k_valid=K, splits=splits, accumulate=1, alpha=1.0,
                       a_rowmap=st.data_ptr())
    for _ in range(3):
        capi.call('tfx_gemm_tn', a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); st.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); capi.call('tfx_gemm_tn', a, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    s = st.cpu().double(); s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    d = lambda i, j: (s[:, j] - s[:, i])
    print(f'TN {M}: {N}x{K}  blocks {len(s)}  event {e0.elapsed_time(e1) * 1e3:.1f} us  kernel span {float(s[:, 4].max() - t0):.0f} ticks ({float(s[:, 4].max() - t0) / (e0.elapsed_time(e1) * 1e3):.0f} ticks/us)')
    for name, x in (('prologue 0->1', d(0, 1)), ('k loop 1->2', d(1, 2)), ('  in waitcnt+barrier', s[:, 5]), ('  in DMA issue', s[:, 6]), ('  in frag reads + MFMA', s[:, 7]),
                    ('atomics issue 2->3', d(2, 3)), ('atomics drain 3->4', d(3, 4)), ('block total 0->4', d(0, 4)), ('block start offset', s[:, 0] - t0), ('block end offset', s[:, 4] - t0)):
        print(f'   {name:24s} median {float(x.median()):9.0f}  p10 {float(x.quantile(0.1)):9.0f}  p90 {float(x.quantile(0.9)):9.0f}  max {float(x.max()):9.0f}')
