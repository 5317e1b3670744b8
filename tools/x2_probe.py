"""EXPERIMENT (round 6): the K = 512 NT GEMMs of the config-2 step on the product kernels against the two-blocks-per-CU kernel (TFX_NT_X2, gemm_nt_x2_kernel),
with and without the first-round stagger of the second block on a CU.  The library reads its switches once per process: run once per setting.
    TFX_NT_X2=0 python tools/x2_probe.py ; TFX_NT_X2=2 TFX_X2_STAGGER=0 python tools/x2_probe.py ; TFX_NT_X2=2 TFX_X2_STAGGER=15000 python tools/x2_probe.py"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
from bench_gemm import st, dev, BF

E = capi.ENUMS
M, d, dip = 65536, 512, 1408
torch.manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(BF)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def h(*ts):
    torch.cuda.synchronize()
    m = hashlib.sha1()
    for t in ts:
        m.update(t.view(torch.int16).cpu().numpy().tobytes())
    return m.hexdigest()[:10]


A = rnd(M, d)
cases = []
W1, b1 = rnd(2 * dip, d, scale=d ** -0.5), torch.randn(2 * dip, device=dev)
ag, hm = torch.zeros(M, 2 * dip, device=dev, dtype=BF), torch.zeros(M, dip, device=dev, dtype=BF)
cases.append(('GEGLU fwd  2816 x 512', dict(A=A, lda=d, B=W1, ldb=d, M=M, N=2 * dip, K=d, epi=E['TFX_EPI_GEGLU'], C=ag, ldc=2 * dip, C2=hm, ldc2=dip, bias=b1), (ag, hm), 2 * M * 2 * dip * d))
W2t = rnd(dip, d, scale=d ** -0.5)
dag = torch.zeros(M, 2 * dip, device=dev, dtype=BF)
cases.append(('GEGLU bwd  1408 x 512', dict(A=A, lda=d, B=W2t, ldb=d, M=M, N=dip, K=d, epi=E['TFX_EPI_GEGLU_BWD'], C=dag, ldc=2 * dip, aux=ag, ldaux=2 * dip), (dag,), 2 * M * dip * d))
Wo, R, Cr = rnd(d, d, scale=d ** -0.5), rnd(M, d), torch.zeros(M, d, device=dev, dtype=BF)
cases.append(('resid       512 x 512', dict(A=A, lda=d, B=Wo, ldb=d, M=M, N=d, K=d, epi=E['TFX_EPI_RESID'], C=Cr, ldc=d, R=R, ldr=d), (Cr,), 2 * M * d * d))
Cp = torch.zeros(M, d, device=dev, dtype=BF)
cases.append(('plain       512 x 512', dict(A=A, lda=d, B=Wo, ldb=d, M=M, N=d, K=d, epi=E['TFX_EPI_BF16'], C=Cp, ldc=d), (Cp,), 2 * M * d * d))
Wq, Cq = rnd(1544, d, scale=d ** -0.5), torch.zeros(M, 1544, device=dev, dtype=BF)
cases.append(('plain      1544 x 512', dict(A=A, lda=d, B=Wq, ldb=d, M=M, N=1544, K=d, epi=E['TFX_EPI_BF16'], C=Cq, ldc=1544), (Cq,), 2 * M * 1544 * d))
A2, Wf, Cf = rnd(M, 2816), rnd(d, 2816, scale=2816 ** -0.5), torch.zeros(M, d, device=dev, dtype=BF)
cases.append(('plain       512 x 2816', dict(A=A2, lda=2816, B=Wf, ldb=2816, M=M, N=d, K=2816, epi=E['TFX_EPI_BF16'], C=Cf, ldc=d), (Cf,), 2 * M * d * 2816))
print(f"TFX_NT_X2={os.environ.get('TFX_NT_X2', '0')} TFX_X2_STAGGER={os.environ.get('TFX_X2_STAGGER', '-')} TFX_GELU_TABLE={os.environ.get('TFX_GELU_TABLE', '1')}")
for name, kw, outs, fl in cases:
    a = capi.make_args('tfx_gemm_nt_args', **kw)
    capi.call('tfx_gemm_nt', a, st())
    hs = h(*outs)
    t = timeit(lambda: capi.call('tfx_gemm_nt', a, st()))
    print(f'  {name}: {t:7.1f} us  {fl / t / 1e6:7.0f} TFLOP/s   out {hs}', flush=True)
