"""GPU time of ONE decode forward (dim 1024 / depth 24, 64 samples) replayed as a hipGraph, without any host round trip: the floor of a text step
(64 rows) and of a joint modality evaluation (2 x 64 x 4 rows).   python tools/bench_decode_step.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import Transfusion
from transfusion_pytorch_amd.sampling import Sampler
dev = torch.device('cuda', 0)
m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=1024, depth=24)).to(dev).eval()
smp = Sampler(m)
m._decode_plans = {}
B, maxlen, filled = 64, 512, 200
stream = m._stream()
cases = ((B, 1, 'text step (64 rows)'), (2 * B, 4, 'joint modality evaluation (512 rows)'))
if len(sys.argv) > 1:                      # rows-per-step sweep: "samples:Lq,samples:Lq,..." (what compaction of a mixed step would buy)
    cases = tuple((int(a.split(':')[0]), int(a.split(':')[1]), f'{a.split(":")[0]} samples x {a.split(":")[1]} rows') for a in sys.argv[1].split(','))
for rows, Lq, name in cases:
    cache = smp._alloc_cache(rows, maxlen)
    p = smp._decode_plan((name, cache.data_ptr()), rows, Lq, cache, Lq > 1)
    T = rows * Lq
    pos = (np.arange(rows)[:, None] * maxlen + filled + np.arange(Lq)[None]).reshape(-1).astype(np.int32)
    smp._load(p, ids=np.zeros(T, np.int32), pos=pos, kve=np.full(T, filled + Lq, np.int32), rot=np.full(T, filled, np.int32),
              tok_inst=(np.repeat(np.arange(rows), Lq) if Lq > 1 else np.full(T, -1)).astype(np.int32))
    if Lq > 1:
        for t in p.row_tok:
            p.row_tok[t].copy_(torch.arange(T, dtype=torch.int32, device=dev)); p.row_src[t].copy_(p.row_tok[t]); p.row_inst[t].copy_(torch.arange(rows, dtype=torch.int32, device=dev).repeat_interleave(Lq)); p.set_noise(t, None)
    end = p.fwd_logits_end if Lq == 1 else p.fwd_embed_end
    for _ in range(3):
        smp._run(p, stream, 0, end)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        smp._run(p, stream, 0, end)
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 50:.3f} ms per forward ({len(p.fwd[:end])} launches)')
