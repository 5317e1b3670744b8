"""Per-section cycle totals of the attention forward tile loop (library built with TFX_HIPCC_EXTRA=-DTFX_ATTN_TIMING).
   sections: 0 wait+barrier+DMA issue, 1 S = K.Q^T (8 MFMA + fragment reads), 2 soft-cap/exp (VALU), 3 P.V (8 MFMA + tr reads)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import capi
dev = 'cuda'; BF = torch.bfloat16
b, h, n = 64, 8, 1024
HD = h * 64; T = b * n
torch.manual_seed(0)
qk = torch.randn(T, 2 * HD, device=dev); qk[:, :HD] *= 0.125; qk = qk.to(BF)      # |q~| = 1, |k~| = 8: the statistics QK-RMSNorm gives
v = torch.randn(T, HD, device=dev).to(BF)
gate = torch.randn(T, h, device=dev).to(BF)
# canonical structure: 32 x [24 text + 4 latent]; latent tokens see their whole instance
pos = torch.arange(n, device=dev)
kv_end = (pos + 1).clone()
for s in range(32):
    a0 = s * 28 + 24
    kv_end[a0:a0 + 4] = a0 + 4
kv_end = kv_end.clamp(max=n).to(torch.int32).repeat(b).contiguous()
q_start = torch.zeros(T, device=dev, dtype=torch.int32)
out = torch.empty(T, HD, device=dev, dtype=BF); lse = torch.empty(b * h * n, device=dev)
nqb = (n + 127) // 128
stamps = torch.zeros(b * h * nqb * 4 * 5, device=dev, dtype=torch.int64)
a = capi.make_args('tfx_attn_args', q=qk, k=qk[:, HD:], v=v, ld_q=2 * HD, ld_k=2 * HD, ld_v=HD, gate=gate, ld_gate=h, kv_end=kv_end, q_start=q_start,
                   out=out, ld_out=HD, lse=lse, b=b, h=h, n=n, softcap=50.0, dq=stamps.data_ptr())
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    capi.call('tfx_attn_fwd', a, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); capi.call('tfx_attn_fwd', a, st); e1.record(); torch.cuda.synchronize()
print(f'attn_fwd b{b} h{h} n{n}: {e0.elapsed_time(e1) * 1e3:.1f} us')
s = stamps.view(-1, 5).cpu().double()
tiles = s[:, 4].sum()
names = ['wait+barrier+dma issue', 'S=K.Q^T (mfma+frag reads)', 'softcap+exp (valu)', 'P.V (mfma+tr reads)'] if os.environ.get('TFX_ATTN_PIPE') == '0' else ['vmcnt wait + barrier', 'DMA issue', 'stage A: soft-cap (2 units)', 'stage B: exp/pack + 16 MFMA (2 units)']
tot = s[:, :4].sum()
for i in range(4):
    print(f'  {names[i]:28s} {float(s[:, i].sum() / tiles):8.0f} ticks per wave-tile  ({100 * float(s[:, i].sum() / tot):4.1f} %)')
print(f'  total {float(tot / tiles):.0f} ticks per wave-tile; wave-tiles {int(tiles)}')
