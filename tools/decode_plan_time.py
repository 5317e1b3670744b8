import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from transfusion_pytorch_amd import Transfusion
from transfusion_pytorch_amd.sampling import Sampler
dev = torch.device('cuda', 0)
m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=1024, depth=24)).to(dev).eval()
smp = Sampler(m); m._decode_plans = {}
cache = smp._alloc_cache(128, 512)
stream = m._stream()
for name, Lq, mixed in (('mix', 5, True), ('txt', 1, False)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = smp._decode_plan((name, cache.data_ptr()), 128, Lq, cache, mixed, n_inst=31, tile_attn=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p.set_rope_tables(*m._rope_tables(512))
    end = p.fwd_pred_end if mixed else p.fwd_logits_end
    smp._run(p, stream, p.fwd_cond[1] if mixed else 0, end)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    smp._run(p, stream, p.fwd_cond[1] if mixed else 0, end)          # graph capture + first replay
    torch.cuda.synchronize(); t3 = time.perf_counter()
    smp._run(p, stream, p.fwd_cond[1] if mixed else 0, end)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'{name}: plan build {1e3*(t1-t0):.1f} ms, first (list) run {1e3*(t2-t1):.1f} ms, capture + replay {1e3*(t3-t2):.1f} ms, replay {1e3*(t4-t3):.2f} ms')
