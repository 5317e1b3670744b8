"""Does the chip run an MFMA-bound kernel of one stream beside an HBM-bound kernel of another?  Two independent training replicas at half the batch,
each issued from its own thread on its own HIP stream, against one replica at the full batch (weight gradients on the caller's stream in both arms:
TFX_SIDE_STREAM=0).  If two half-batch steps running concurrently finish clearly sooner than one full-batch step, splitting the step into two
micro-batches on two streams is worth building.
    TFX_SIDE_STREAM=0 python tools/bench_two_streams.py [steps]"""
import os
import sys
import threading
import time

import torch

os.environ.setdefault('TFX_SIDE_STREAM', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                # noqa: E402
from transfusion_pytorch_amd import Transfusion            # noqa: E402
from transfusion_pytorch_amd.optim import FusedAdam        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda', 0)


def make(b, seed):
    torch.manual_seed(seed)
    m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8)).to(dev).train()
    opt = FusedAdam(m, lr=3e-4, max_grad_norm=0.5)
    gen = torch.Generator(device=dev).manual_seed(seed)
    batch = bench.canonical_batch(b, dev, gen)
    return m, opt, batch


def run(m, opt, batch, n, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(n):
            loss = m(batch); loss.backward(); opt.step(); opt.zero_grad()


full = make(64, 0)
run(*full, 3); torch.cuda.synchronize()
t0 = time.perf_counter(); run(*full, steps); torch.cuda.synchronize(); t_full = (time.perf_counter() - t0) / steps
print(f'one replica, batch 64, one stream: {t_full * 1e3:.2f} ms/step = {64 / t_full:.0f} samples/s')
del full; torch.cuda.empty_cache()

halves = [make(32, 1), make(32, 2)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
for h, s in zip(halves, streams):
    run(*h, 3, s)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(*halves[0], steps, streams[0]); torch.cuda.synchronize(); t_half = (time.perf_counter() - t0) / steps
print(f'one replica, batch 32, alone: {t_half * 1e3:.2f} ms/step = {32 / t_half:.0f} samples/s')
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(*h, steps, s)) for h, s in zip(halves, streams)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize(); t_two = (time.perf_counter() - t0) / steps
print(f'two replicas, batch 32 each, two threads / two streams: {t_two * 1e3:.2f} ms per pair of steps = {64 / t_two:.0f} samples/s '
      f'({t_full / t_two:.3f}x the single-stream batch-64 step)')
