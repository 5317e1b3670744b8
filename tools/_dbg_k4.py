import sys, ctypes, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_kernels_gpu as tk
from transfusion_pytorch_amd import capi
DEV='cuda'; BF=torch.bfloat16
for T in (66000, 65536+256+208, 66000):
    H=8; torch.manual_seed(31)
    d, HD = 512, H*64; N = 3*HD+H; ldq=(N+63)//64*64
    u, W = tk.rnd(T, d), tk.rnd(N, d, scale=d**-0.5)
    gq = torch.randn(64, device=DEV)*0.2; gk = torch.randn(64, device=DEV)*0.2
    pos = torch.randint(0, 1000, (T,), device=DEV, dtype=torch.int32)
    freqs = 1./(10000**(torch.arange(0,64,2).float()/64)); ang = torch.arange(1024).float()[:,None]*freqs[None]
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    C0 = torch.zeros(T, ldq, device=DEV, dtype=BF); qk0 = torch.zeros(T, 2*HD, device=DEV, dtype=BF)
    tk.gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_BF16'], C=C0, ldc=ldq)
    a = capi.make_args('tfx_qk_norm_rope_args', T=T, H=H, qkv=C0, ld_qkv=ldq, qk=qk0, ld_qk=2*HD, gamma_q=gq, gamma_k=gk, rot_pos=pos, cos_tab=cos_t, sin_tab=sin_t, q_scale=0.125, norm_scale=8.0)
    capi.call('tfx_qk_norm_rope_fwd', a, tk.stream())
    for rep in range(3):
        C1 = torch.zeros(T, ldq, device=DEV, dtype=BF); qk1 = torch.zeros(T, 2*HD, device=DEV, dtype=BF)
        tk.gemm_nt(A=u, lda=d, B=W, ldb=d, M=T, N=N, K=d, epi=capi.ENUMS['TFX_EPI_QKV_NORM_ROPE'], C=C1, ldc=ldq, C2=qk1, ldc2=2*HD, qk_heads=H, qk_gamma_q=gq, qk_gamma_k=gk, qk_rot_pos=pos, qk_cos=cos_t, qk_sin=sin_t, qk_q_scale=0.125, qk_norm_scale=8.0)
        torch.cuda.synchronize()
        bad = (qk1 != qk0).nonzero()
        print(T, rep, 'mismatches', bad.shape[0], bad[:12].tolist(), 'raw equal', torch.equal(C1[:, :N], C0[:, :N]))
        if bad.shape[0]:
            r, c = bad[0].tolist(); print('   values', qk1[r, c].item(), qk0[r, c].item(), 'row%256', r % 256, 'col', c)
