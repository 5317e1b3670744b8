"""host profile of the training step (bench workload, fresh batch per step): where the ~24 ms of host time per step go.   python tools/prof_train_host.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transfusion_pytorch_amd import Transfusion
from transfusion_pytorch_amd.optim import FusedAdam
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).to(dev).train()
opt = FusedAdam(m, lr=3e-4, max_grad_norm=0.5)
g = torch.Generator(device=dev).manual_seed(1234)
batches = [bench.canonical_batch(64, dev, g) for _ in range(12)]
def step(k):
    loss = m(batches[k % len(batches)]); loss.backward(); opt.step(); opt.zero_grad()
for k in range(4): step(k)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for k in range(10): step(k)
pr.disable(); host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f'host {host / 10 * 1e3:.2f} ms/step (under cProfile), wall {wall / 10 * 1e3:.2f} ms/step')
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
