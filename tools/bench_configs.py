"""The other SURVEY.md section 8(d) configurations on ONE MI355X (the 8-GPU / 4-GPU DP runs are the driver's; per-GPU work is the same):
   config 3: dim 1024 / depth 24, batch 64 x 1024 canonical samples          (train step)
   config 4: two modalities (384, 192), dim 768 / depth 16, batch 64 x 1024   (train step)
   config 5: sample_many, dim 1024 / depth 24, 64 prompts of the four README kinds, max_length 256, 16 ODE steps, cfg 3, greedy
   python tools/bench_configs.py [3 4 5]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from transfusion_pytorch_amd import Transfusion
from transfusion_pytorch_amd.optim import FusedAdam

dev = torch.device('cuda', 0)
which = [int(a) for a in sys.argv[1:]] or [3, 4, 5]


def train(model, batch, steps=5, warmup=2):
    opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
    def step():
        loss = model(batch); loss.backward(); opt.step(); opt.zero_grad(); return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, float(loss.detach())


if 3 in which:
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=1024, depth=24)).to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(1234)
    dt, loss = train(m, bench.canonical_batch(64, dev, gen))
    f = bench.f_core_per_sample(d=1024, D=24)
    print(f'config 3 (dim1024/d24, b=64): {dt * 1e3:.1f} ms/step = {64 / dt:.1f} samples/s, F_core {f / 1e9:.1f} GF/sample -> {64 / dt * f / 1e12:.0f} TFLOP/s '
          f'({64 / dt * f / 2.5e15 * 100:.1f} % of 2.5 PF), loss {loss:.4f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
    del m; torch.cuda.empty_cache()

if 4 in which:
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=256, dim_latent=(384, 192), modality_default_shape=((4,), (2,)), transformer=dict(dim=768, depth=16)).to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(1234)
    batch = []
    for _ in range(64):
        parts = []
        for i in range(32):
            parts.append(torch.randint(0, 256, (25 if i < 31 else 24,), device=dev, generator=gen))
            parts.append((0, torch.randn(4, 384, device=dev, generator=gen)) if i % 2 == 0 else (1, torch.randn(2, 192, device=dev, generator=gen)))
        batch.append(parts)
    dt, loss = train(m, batch)
    plan = m._live[0]
    # mask-aware F_core with 16 instances of length 4 and 16 of length 2
    d, D, h, dh, n = 768, 16, 8, 64, plan.n
    hd, di = h * dh, int(d * 8 / 3)
    pairs = n * (n + 1) / 2 + 16 * (4 * 3 / 2) + 16 * (2 * 1 / 2)
    f = 6 * n * D * (d * 2 * hd + d * hd + d * h + hd * d + d * 2 * di + di * d + 2 * dh * h * pairs / n)
    print(f'config 4 (two modalities, dim768/d16, b=64, n={n}): {dt * 1e3:.1f} ms/step = {64 / dt:.1f} samples/s, F_core {f / 1e9:.1f} GF/sample -> '
          f'{64 / dt * f / 1e12:.0f} TFLOP/s ({64 / dt * f / 2.5e15 * 100:.1f} % of 2.5 PF), loss {loss:.4f}', flush=True)
    del m; torch.cuda.empty_cache()

if 5 in which:
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=1024, depth=24)).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234)
    prompts = []
    for _ in range(16):
        prompts += [torch.randint(0, 256, (16,), device=dev, generator=g), (0, torch.randn(4, 384, device=dev, generator=g)), None,
                    [torch.randint(0, 256, (8,), device=dev, generator=g), (0, torch.randn(6, 384, device=dev, generator=g))]]
    noise = torch.randn(16, 384, device=dev, generator=g)
    for force in (None, 0):
        kw = dict(max_length=256, modality_steps=16, cfg_scale=3., text_temperature=0., init_modality_noise=noise, fixed_modality_shape=(4,))
        if force is not None:
            kw['force_modality_at_start'] = force
        m.sample_many(prompts, **{**kw, 'max_length': 24})                    # warm-up (plans, shadows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = m.sample_many(prompts, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ntok = sum(sum((p.numel() if not isinstance(p, tuple) else p[1].shape[0]) for p in s) for s in res)
        nmod = sum(sum(isinstance(p, tuple) for p in s) for s in res)
        print(f'config 5 (sample_many dim1024/d24, 64 prompts, max_length 256, 16 ODE steps, cfg 3, force_modality_at_start={force}): {dt * 1e3:.0f} ms, '
              f'{ntok} tokens in the returned samples, {nmod} modality instances', flush=True)
