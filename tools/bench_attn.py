"""Attention kernels alone (through the C ABI), canonical geometry of the bench step: b = 64 samples x 8 heads x 1024 tokens,
32 x [24 text + 4 latent] per sample (prefix-extension mask), operands with the statistics QK-RMSNorm gives them (|q~| = 1, |k~| = 8:
scores ~ N(0, 1), the cubic branch of the soft-cap).  HIP events around `reps` back-to-back launches.

    python tools/bench_attn.py [reps]          TFX_LIB=<other build> selects the library (A/B inside one gpurun call)

Prints us per launch and the MFMA rate on the mask-aware algorithmic flops (pairs = n(n+1)/2 + sum L(L-1)/2; forward 4*64*pairs
flop per (sample, head), backward 2.5x that)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi  # noqa: E402

dev, BF = 'cuda', torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b, h, n = 64, 8, 1024
HD, T = h * 64, b * n
torch.manual_seed(0)
ldq = 3 * HD + 64
qk = torch.randn(T, 2 * HD, device=dev)
qk[:, :HD] *= 0.125
qk = qk.to(BF)
qkvg = torch.randn(T, ldq, device=dev).to(BF)            # (unused) | (unused) | v | gates
pos = torch.arange(n, device=dev)
kv_end, q_start = (pos + 1).clone(), pos.clone()
for s in range(32):
    a0 = 1 + s * 32 + 27                                  # first instance at offset 28 of the packed sample
    a0 = min(a0, n - 4)
    kv_end[a0:a0 + 4] = a0 + 4
    q_start[a0:a0 + 4] = a0
kv_end = kv_end.clamp(max=n).to(torch.int32).repeat(b).contiguous()
q_start = q_start.to(torch.int32).repeat(b).contiguous()
out = torch.empty(T, HD, device=dev, dtype=BF); lse = torch.empty(b, h, n, device=dev)
dout = torch.randn(T, HD, device=dev).to(BF); do_eff = torch.empty(T, HD, device=dev, dtype=BF)
delta = torch.empty(b, h, n, device=dev); dqk = torch.empty(T, 2 * HD, device=dev, dtype=BF); dqkvg = torch.zeros(T, ldq, device=dev, dtype=BF)
a = capi.make_args('tfx_attn_args', q=qk, k=qk[:, HD:], v=qkvg[:, 2 * HD:], ld_q=2 * HD, ld_k=2 * HD, ld_v=ldq, gate=qkvg[:, 3 * HD:], ld_gate=ldq,
                   kv_end=kv_end, q_start=q_start, out=out, ld_out=HD, lse=lse, b=b, h=h, n=n, softcap=50.0,
                   dout=dout, ld_dout=HD, do_eff=do_eff, ld_do=HD, delta=delta, dgate=dqkvg[:, 3 * HD:], ld_dgate=ldq,
                   dq=dqk, dk=dqk[:, HD:], dv=dqkvg[:, 2 * HD:], ld_dq=2 * HD, ld_dk=2 * HD, ld_dv=ldq)
st = torch.cuda.current_stream().cuda_stream
pairs = n * (n + 1) // 2 + 32 * 6
f_fwd = 4. * 64 * pairs * b * h


def timed(fn):
    for _ in range(3):
        capi.call(fn, a, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        capi.call(fn, a, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tf, tb = timed('tfx_attn_fwd'), timed('tfx_attn_bwd')
print(f'lib {os.path.basename(os.path.dirname(os.path.dirname(capi.LIB_PATH)))}/{os.path.basename(capi.LIB_PATH)}: '
      f'attn_fwd {tf:7.1f} us = {f_fwd / tf * 1e-6:6.1f} TFLOP/s ({f_fwd / tf * 1e-6 / 25:4.1f} % of 2.5 PF) | '
      f'attn_bwd (prep + dkv + dq) {tb:7.1f} us = {2.5 * f_fwd / tb * 1e-6:6.1f} TFLOP/s ({2.5 * f_fwd / tb * 1e-6 / 25:4.1f} %)')
print(f'  checksums: out {float(out.float().abs().mean()):.6f} dq {float(dqk[:, :HD].float().abs().mean()):.6f} dk {float(dqk[:, HD:].float().abs().mean()):.6f} '
      f'dv {float(dqkvg[:, 2 * HD:3 * HD].float().abs().mean()):.6f}')
