"""Host profile of bench.py's isolated structure-miss step (after cached steps and a ragged phase on the same plan)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                # noqa: E402
from transfusion_pytorch_amd import Transfusion            # noqa: E402
from transfusion_pytorch_amd.optim import FusedAdam        # noqa: E402
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).to(dev).train()
opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
gen = torch.Generator(device=dev).manual_seed(1)


def step(batch):
    loss = model(batch); loss.backward(); opt.step(); opt.zero_grad()


main = [bench.canonical_batch(64, dev, gen) for _ in range(4)]
for b in main:
    step(b)
if '--ragged' in sys.argv:
    for k in range(6):
        step(bench.ragged_batch(64, dev, gen, seed=k))
miss = bench.canonical_batch(64, dev, gen, text_len=23, last_text_len=54)
if '--gc' in sys.argv:
    import gc; gc.collect()
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
step(miss)
pr.disable(); th = time.perf_counter() - t0
torch.cuda.synchronize(); tw = time.perf_counter() - t0
print(f'miss step: host {th * 1e3:.1f} ms, wall {tw * 1e3:.1f} ms, plans {len(model._plans)}')
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
