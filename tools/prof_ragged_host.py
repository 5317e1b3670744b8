"""Host profile of the RAGGED steady state (every batch a new structure signature): where the ~12 ms per step go that the cached step does not pay.
    python tools/prof_ragged_host.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                # noqa: E402
from transfusion_pytorch_amd import Transfusion            # noqa: E402
from transfusion_pytorch_amd.optim import FusedAdam        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).to(dev).train()
opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
gen = torch.Generator(device=dev).manual_seed(1)
rb = [bench.ragged_batch(64, dev, gen, seed=k) for k in range(steps + 3)]


def step(batch):
    loss = model(batch); loss.backward(); opt.step(); opt.zero_grad()


for k in range(3):
    step(rb[k])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(steps):
    step(rb[3 + k])
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f'ragged: host {host / steps * 1e3:.1f} ms/step, wall {wall / steps * 1e3:.1f} ms/step (profiler on)')
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
