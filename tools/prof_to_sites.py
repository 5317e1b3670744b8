"""Which `.to(...)` call sites block the host in the ragged steady state (wall time per call site of torch.Tensor.to / copy_ / empty(pin_memory)).
    python tools/prof_to_sites.py [steps]"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                # noqa: E402
from transfusion_pytorch_amd import Transfusion            # noqa: E402
from transfusion_pytorch_amd.optim import FusedAdam        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).to(dev).train()
opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
gen = torch.Generator(device=dev).manual_seed(1)
rb = [bench.ragged_batch(64, dev, gen, seed=k) for k in range(steps + 3)]
acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(name, fn):
    def f(*a, **k):
        fr = sys._getframe(1)
        t0 = time.perf_counter()
        r = fn(*a, **k)
        e = acc[(name, os.path.basename(fr.f_code.co_filename), fr.f_lineno)]
        e[0] += 1; e[1] += time.perf_counter() - t0
        return r
    return f


def step(batch):
    loss = model(batch); loss.backward(); opt.step(); opt.zero_grad()


for k in range(3):
    step(rb[k])
torch.cuda.synchronize()
torch.Tensor.to = wrap('to', torch.Tensor.to)
torch.Tensor.copy_ = wrap('copy_', torch.Tensor.copy_)
torch.Tensor.long = wrap('long', torch.Tensor.long)
torch.Tensor.float = wrap('float', torch.Tensor.float)
torch.empty = wrap('empty', torch.empty)
torch.full = wrap('full', torch.full)
torch.cat = wrap('cat', torch.cat)
t0 = time.perf_counter()
marks = []
for k in range(steps):
    a = time.perf_counter(); loss = model(rb[3 + k]); b = time.perf_counter(); loss.backward(); c = time.perf_counter(); opt.step(); opt.zero_grad(); d = time.perf_counter()
    marks.append((b - a, c - b, d - c))
torch.cuda.synchronize()
print(f'wall {(time.perf_counter() - t0) / steps * 1e3:.1f} ms/step; host forward / backward / optimizer ms:', [tuple(round(x * 1e3, 1) for x in m) for m in marks])
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f'{t / steps * 1e3:8.2f} ms/step  {n / steps:5.1f} calls/step  {k}')
