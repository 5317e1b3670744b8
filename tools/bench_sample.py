"""Throughput of the KV-cached decode twin (sample_many / generate_text_only) on one GPU: tokens per second, host-bound or not.
   python tools/bench_sample.py [--dim 512 --depth 8 --batch 64 --new 64]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import Transfusion

ap = argparse.ArgumentParser()
ap.add_argument('--dim', type=int, default=512); ap.add_argument('--depth', type=int, default=8)
ap.add_argument('--batch', type=int, default=64); ap.add_argument('--new', type=int, default=64); ap.add_argument('--prompt', type=int, default=128)
a = ap.parse_args()
torch.manual_seed(0)
m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=a.dim, depth=a.depth)).cuda().eval()
prompt = torch.randint(0, 256, (a.batch, a.prompt), device='cuda')
m.generate_text_only(prompt, a.prompt + 4, temperature=0.)          # warm-up (plans, shadows)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = m.generate_text_only(prompt, a.prompt + a.new, temperature=0.)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'generate_text_only dim{a.dim}/d{a.depth}: batch {a.batch}, prompt {a.prompt}, {a.new} new tokens: {dt * 1e3:.1f} ms '
      f'= {dt / a.new * 1e3:.2f} ms/step, {a.batch * a.new / dt:.0f} tokens/s')
prompts = [[torch.randint(0, 256, (16,), device='cuda'), (0, torch.randn(4, 384, device='cuda'))] for _ in range(a.batch)]
m.sample_many(prompts, max_length=8, modality_steps=4)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = m.sample_many(prompts, max_length=a.new, modality_steps=8)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ntok = sum(sum((len(p) if not isinstance(p, tuple) else p[1].shape[0]) for p in s) for s in res)
print(f'sample_many: {a.batch} prompts, max_length {a.new}: {dt * 1e3:.1f} ms, {ntok} tokens in the returned samples')
