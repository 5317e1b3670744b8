"""bench.py - training-step throughput of the native Transfusion hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...            # N > 1 without a launcher: re-executes itself under torch.distributed.run (one rank per GPU, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config 3 | --config 4   # BASELINE configs[2] (dim1024/d24, the 8-GPU model) / configs[3] (two modalities, dim768/d16, the 4-GPU model)

Workload (BASELINE.json configs[1]): Transfusion(num_text_tokens=256, dim_latent=384, dim=512, depth=8), per-GPU batch
64 x packed length 1024 of interleaved text + (4,384) latents (the canonical synthetic sample of SURVEY.md section 8(d)),
bf16 compute / fp32 master.  A step = model(batch) [packing + forward] + loss.backward() [native backward] + one RCCL
all-reduce of the flat gradient (N > 1) + fused clip(0.5)+Adam(3e-4) (train_toy.py:50-57).  Inputs are resident in HBM.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel family timed with HIP events on the launch stream inside the
timed region) and, at N == 1, `cpu_baseline` (the oracle restatement timed on the host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md


def measure_mfma_peak(dev):
    """`roofline.measured_peak`: what a register-resident MFMA loop (no memory traffic, random bf16 operands, two blocks per CU) sustains on THIS box, measured in the
    run behind the timed region (tfx_mfma_peak_probe; ~3 ms): the chip is power-limited - 1.66-1.86 PFLOP/s by box against the 2.5 PFLOP/s the peak is quoted at,
    and less again for kernels that keep the vector units busy next to the matrix pipe (profiles/r02_power_clock.txt, profiles/r06_attn_bwd_what_bounds_it.txt)."""
    from transfusion_pytorch_amd import capi
    ops = (torch.rand(128, 8, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.zeros(2, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks, iters = 2 * cus, 6000
    st = torch.cuda.current_stream(dev).cuda_stream
    capi.check(capi.lib().tfx_mfma_peak_probe(ops.data_ptr(), out.data_ptr(), iters // 10, blocks, st), 'tfx_mfma_peak_probe')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    capi.check(capi.lib().tfx_mfma_peak_probe(ops.data_ptr(), out.data_ptr(), iters, blocks, st), 'tfx_mfma_peak_probe')
    e1.record(); e1.synchronize()
    return 2.0 * 32 * 32 * 16 * 4 * iters * blocks * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def canonical_batch(b, device, gen, num_text_tokens=256, dim_latent=384, n_inst=32, latent_len=4, text_len=24, last_text_len=23):
    """per sample 64 parts alternating randint text(24) / randn latent (4,384); packs to 1025 tokens."""
    batch = []
    for _ in range(b):
        parts = []
        for i in range(n_inst):
            tl = text_len if i < n_inst - 1 else last_text_len
            parts.append(torch.randint(0, num_text_tokens, (tl,), device=device, generator=gen))
            parts.append(torch.randn(latent_len, dim_latent, device=device, generator=gen))
        batch.append(parts)
    return batch


def two_modality_batch(b, device, gen, num_text_tokens=256, dim_latents=(384, 192), lens=(4, 2), n_inst=32, text_len=25, last_text_len=24):
    """SURVEY.md section 8(d) config 4 (README.md:59-67 scaled): per sample 32 x [text, (type, latent)] with even instances type 0 = randn(4,384) and odd
    instances type 1 = randn(2,192); text fillers 25 (last 24) so that the sample packs to 1025 tokens (16 x 8 + 16 x 6 instance tokens + 799 text + sos/eos)."""
    batch = []
    for _ in range(b):
        parts = []
        for i in range(n_inst):
            parts.append(torch.randint(0, num_text_tokens, (text_len if i < n_inst - 1 else last_text_len,), device=device, generator=gen))
            ty = i % 2
            parts.append((ty, torch.randn(lens[ty], dim_latents[ty], device=device, generator=gen)))
        batch.append(parts)
    return batch


def ragged_batch(b, device, gen, num_text_tokens=256, dim_latent=384, seed=0):
    """a batch whose STRUCTURE is new: per sample a random number of [text, latent] pairs with random text lengths (latents of 2..6 rows), packed
    lengths between 982 and 1022 tokens incl. [sos] / [eos] (padded to the main workload's 1024 columns) - what a real corpus hands the packer
    every step (the reference re-packs every step, MP:850-936).  Until the end of round 3 the special tokens were under-counted by one per
    instance: the batches packed to ~1050 tokens and ran on 1088-column plans (6 % more tokens, 544 instead of 512 GEMM tiles per 512 columns)."""
    import random
    rng = random.Random(seed)
    batch = []
    for _ in range(b):
        parts, total, target = [], 0, rng.randint(980, 1020)
        while True:
            tl, ll = rng.randint(12, 36), rng.randint(2, 6)
            cost = tl + ll + 4                                     # + [meta], one shape character, [som] and [eom] of the instance (packing.py)
            if total + cost > target:
                break
            parts.append(torch.randint(0, num_text_tokens, (tl,), device=device, generator=gen))
            parts.append(torch.randn(ll, dim_latent, device=device, generator=gen))
            total += cost
        parts.append(torch.randint(0, num_text_tokens, (max(target - total, 1),), device=device, generator=gen))
        batch.append(parts)
    return batch


def score_pairs(n=1024, inst_lens=(4,) * 32):
    """mask-aware (query, key) pairs of one sample: the causal triangle + the in-instance pairs above the diagonal"""
    return n * (n + 1) / 2 + sum(L * (L - 1) / 2 for L in inst_lens)


def f_core_per_sample(d=512, D=8, h=8, dh=64, n=1024, inst_lens=(4,) * 32):
    """F_core = 6 n D (P_attn + P_ff + SDPA) flop / sample, mask-aware SDPA pairs."""
    hd, di = h * dh, int(d * 8 / 3)
    p_attn = d * 2 * hd + d * hd + d * h + hd * d
    p_ff = d * 2 * di + di * d
    sdpa = 2 * dh * h * (score_pairs(n, inst_lens) / n)
    return 6 * n * D * (p_attn + p_ff + sdpa)


def classify(launches, k, attn_flops):
    """(family, algorithmic work) of launch k of a plan's list, or None: the MFMA families by entry point (GEMM work = the unpadded 2 M N K the
    engine attached to the args struct; attention = mask-aware score pairs), the HBM-bound token-wise launches by the byte tags the engine
    attached (`LaunchList.meta`: passes of a [T, d] bf16 matrix)."""
    fn, a = launches[k]
    if fn in ('tfx_gemm_nt', 'tfx_gemm_tn'):
        return fn, getattr(a, '_algo_flops', 0.0)
    if fn == 'tfx_attn_fwd':
        return fn, attn_flops
    if fn == 'tfx_attn_bwd':
        return fn, 2.5 * attn_flops                       # S and dP recomputed once more than the minimum is NOT counted: 5 products of the forward's 2
    meta = getattr(launches, 'meta', None)
    if meta and k in meta:
        return meta[k]
    return None


def timed_run(orig_run, launches, stream, lo, hi, families, events, attn_flops=0.0):
    """Plan.run with HIP events (torch events on the launch stream) around every launch of `families`; the launches in between
    are replayed natively (tfx_run_list), exactly as the product path does."""
    n = len(launches)
    hi = n if hi is None else min(hi, n)
    seg = lo
    for k in range(lo, hi):
        c = classify(launches, k, attn_flops)
        if c is not None and c[0] in families:
            orig_run(launches, stream, seg, k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_run(launches, stream, k, k + 1)
            e1.record()
            fn_, a_ = launches[k]
            shape = (getattr(a_, 'M', 0), getattr(a_, 'N', 0), getattr(a_, 'K', 0), getattr(a_, 'epi', -1)) if c[0] in ('tfx_gemm_nt', 'tfx_gemm_tn') else None
            events.append((e0, e1, c[1], c[0], shape))
            seg = k + 1
    orig_run(launches, stream, seg, hi)


def cpu_baseline_reference(budget_s=30.0):
    """the UNMODIFIED reference (TFX_REFERENCE_ROOT, default /root/reference; through oracle/shims for its absent third-party packages) on the host
    cores of THIS box in THIS run: train_toy.py:50-57's step (fwd + bwd + clip 0.5 + Adam 3e-4) at dim512/depth8 on 4 canonical samples.  Returns
    None where the reference tree does not exist (the GPU box of this pool: the caller then times the oracle restatement, kind "port")."""
    from oracle import ref_runner
    if not ref_runner.reference_available():
        return None
    tp = ref_runner.import_reference()
    threads = min(32, os.cpu_count())
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = tp.Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=512, depth=8), modality_processing='flat')
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    bs = 4
    gen = torch.Generator().manual_seed(1234)
    batch = canonical_batch(bs, 'cpu', gen)
    def step():
        loss = model([list(s) for s in batch]); loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5); opt.step(); opt.zero_grad()
    t0 = time.time(); step(); t1 = time.time() - t0
    steps, tsum = 0, 0.0
    while tsum + t1 < budget_s and steps < 3:
        t0 = time.time(); step(); tsum += time.time() - t0; steps += 1
    per_step = (tsum / steps) if steps else t1
    return {'value': bs / per_step, 'unit': 'samples/s', 'cores': threads, 'host_cores': os.cpu_count(), 'kind': 'reference',
            'sample': f'the unmodified reference ({ref_runner.REF_ROOT}, torch fp32 CPU, modality_processing=flat), dim512/depth8, batch {bs} x 1024 canonical samples, '
                      f'{"1 warm-up + " + str(steps) + " timed" if steps else "1 timed (cold)"} step(s) of fwd+bwd+clip+Adam; {threads} torch threads of the box\'s {os.cpu_count()}'}


def cpu_baseline(budget_s=30.0):
    """the reference itself where its tree exists on this box (kind "reference"), else the oracle restatement ("port") on the host cores:
    1 train step (fwd+bwd+clip+Adam) on a bounded sample."""
    try:
        r = cpu_baseline_reference(budget_s)
        if r is not None:
            return r
    except Exception as e:
        print(f'[bench] reference CPU baseline unavailable ({e!r}); timing the oracle restatement', file=sys.stderr)
    from oracle import detdata as D
    from oracle.transfusion_oracle import OracleConfig, train_step
    # 256 torch threads on the GPU box's host oversubscribe badly on these small ops (measured: 469 s / step);
    # 32 threads is the sweet spot of the oracle's einsum-heavy graph - `cores` reports the threads actually used
    threads = min(32, os.cpu_count())
    torch.set_num_threads(threads)
    cfg = OracleConfig(num_text_tokens=256, dim=512, depth=8, dim_latents=(384,))
    sd = D.det_state_dict(cfg.state_dict_shapes(), tag='bench')
    sd = {k: (v.clone().requires_grad_(True) if k not in ('rotary_emb.freqs', 'transformer.to_time_cond.0.weights') else v) for k, v in sd.items()}
    bs = 4                                                    # micro-batch 4 x seq 1024 on the CPU
    batch = D.canonical_batch(bs, key='bench')
    times = D.det_times('bench/t', batch)
    noise = D.det_noise('bench/n', batch, 1)
    state = {}
    t0 = time.time(); train_step(sd, cfg, batch, times, noise, state); t1 = time.time() - t0
    steps, tsum = 0, 0.0
    while tsum + t1 < budget_s and steps < 3:
        t0 = time.time(); train_step(sd, cfg, batch, times, noise, state); dt = time.time() - t0
        tsum += dt; steps += 1
    per_step = (tsum / steps) if steps else t1
    return {'value': bs / per_step, 'unit': 'samples/s', 'cores': threads, 'host_cores': os.cpu_count(), 'kind': 'port',
            'sample': f'oracle restatement (torch fp32 CPU), dim512/depth8, batch {bs} x 1024 canonical samples, '
                      f'{"1 warm-up + " + str(steps) + " timed" if steps else "1 timed (cold)"} step(s) of fwd+bwd+clip+Adam; {threads} torch threads of the box\'s {os.cpu_count()} '
                      '(more threads oversubscribe this graph: 256 threads measured 100x slower)'}


def sample_prompts(n_each, dev, gen, dim_latent=384):
    """n_each x the four README prompt kinds (README.md:162-167)"""
    prompts = []
    for _ in range(n_each):
        prompts += [torch.randint(0, 256, (16,), device=dev, generator=gen), (0, torch.randn(4, dim_latent, device=dev, generator=gen)), None,
                    [torch.randint(0, 256, (8,), device=dev, generator=gen), (0, torch.randn(6, dim_latent, device=dev, generator=gen))]]
    return prompts


def time_sample_many(dev):
    """wall time of `sample_many` (dim1024/depth24, 64 mixed prompts, max_length 256, 16 ODE grid points, cfg 3,
    greedy text, fixed initial noise; free-running and with a forced modality at the start), plus the reduced configuration (dim512/depth8,
    8 prompts, max_length 32) the CPU reference was timed on.  Returns the `runs` dict."""
    from transfusion_pytorch_amd import Transfusion
    out = {'runs': {}}
    for name, dim, depth, n_each, max_len in (('config5', 1024, 24, 16, 256), ('reduced', 512, 8, 2, 32)):
        torch.manual_seed(0)
        m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=dim, depth=depth)).to(dev).eval()
        g = torch.Generator(device=dev).manual_seed(1234)
        prompts = sample_prompts(n_each, dev, g)
        noise = torch.randn(16, 384, device=dev, generator=g)
        for force in (None, 0):
            kw = dict(max_length=max_len, modality_steps=16, cfg_scale=3., text_temperature=0., init_modality_noise=noise, fixed_modality_shape=(4,))
            if force is not None:
                kw['force_modality_at_start'] = force
            m.sample_many(prompts, **{**kw, 'max_length': min(24, max_len)})                    # warm-up (plans, shadows)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = m.sample_many(prompts, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            # the same call again: a serving process repeats a geometry, and the KV-cache buffer + decode plans of a call are kept on the model
            # (sampling.py, round 5) - `seconds` is the FIRST full-length call (plans built and captured inside it, as in rounds 1-4)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.sample_many(prompts, **kw)
            torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
            ntok = sum(sum((p.numel() if not isinstance(p, tuple) else p[1].shape[0]) for p in s) for s in res)
            nmod = sum(sum(isinstance(p, tuple) for p in s) for s in res)
            out['runs'][f'{name}{"_forced" if force is not None else ""}'] = {
                'seconds': dt, 'seconds_repeat_call': dt2, 'prompts': len(prompts), 'tokens_returned': ntok, 'modality_instances': nmod,
                'config': f'dim={dim} depth={depth} max_length={max_len} modality_steps=16 cfg_scale=3 greedy force_modality_at_start={force}'}
        del m
        torch.cuda.empty_cache()
    return out['runs']


def bench_sample(args):
    """`--sample`: `time_sample_many` as ONE JSON line of its own"""
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    out = {'metric': 'sample_many wall time, dim1024 d24, 64 mixed prompts, max_length 256, 16 ODE steps, cfg 3', 'unit': 's', 'n_gpus': 1,
           'higher_is_better': False, 'dtype': 'bf16', 'data': 'synthetic', 'runs': time_sample_many(dev)}
    out['value'] = out['runs']['config5_forced']['seconds']
    # the CPU side of SURVEY 8(d): the UNMODIFIED reference on the reduced configuration, timed in the build container (the reference tree
    # is not on the GPU box) by oracle/time_reference_sampling.py and committed as a fixture - NOT a same-box measurement
    fx = os.path.join(ROOT, 'tests', 'golden', 'reference_sampling_time.json')
    if os.path.exists(fx):
        out['cpu_reference'] = json.load(open(fx))
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + '\n').encode())


# BASELINE.json configs[1] (the metric's), configs[2] (the 8-GPU model) and configs[3] (two modality types, the 4-GPU model); SURVEY.md section 8(d)
CONFIGS = {2: dict(dim=512, depth=8, two=False), 3: dict(dim=1024, depth=24, two=False), 4: dict(dim=768, depth=16, two=True)}


def build_model(dim, depth, two, dev=None):
    """the model of a BASELINE config (README.md:27-35 / :59-67): random init under the caller's seed"""
    from transfusion_pytorch_amd import Transfusion
    if two:
        m = Transfusion(num_text_tokens=256, dim_latent=(384, 192), modality_default_shape=((4,), (2,)), transformer=dict(dim=dim, depth=depth))
    else:
        m = Transfusion(num_text_tokens=256, dim_latent=384, modality_default_shape=(4,), transformer=dict(dim=dim, depth=depth))
    return m if dev is None else m.to(dev)


def make_batch(two, b, dev, gen):
    return two_modality_batch(b, dev, gen) if two else canonical_batch(b, dev, gen)


def inst_lens_of(two):
    return (4, 2) * 16 if two else (4,) * 32


def workload_label(dim, depth, batch, world, use_pg, overlap, two=False):
    cfgname = {(512, 8, False): 'BASELINE config 2', (1024, 24, False): 'BASELINE config 3 (per-GPU share of global batch 512 at 8 GPUs)',
               (768, 16, True): 'BASELINE config 4 (two modality types; per-GPU share of the 4-GPU DP run)'}.get((dim, depth, two), 'non-BASELINE dims')
    sample = ('16 x [25 text + (0, (4,384) latent)] interleaved with 16 x [25 text + (1, (2,192) latent)] per sample' if two
              else '32 x [24 text tokens + (4,384) latent] per sample')
    return (f'{cfgname}: Transfusion dim={dim} depth={depth} heads=8 dim_head=64 num_text_tokens=256 dim_latent={"(384,192)" if two else "384"}; '
            f'per-GPU batch {batch} x seq 1024 ({sample}); '
            'step = pack + fwd + bwd + ' + (('grad all-reduce (4 layer groups + tail, collective launches overlapped with the backward) + '
                                            if overlap else 'ONE grad all-reduce after the backward + ') if use_pg else '')
            + 'clip(0.5) + Adam(3e-4)' + ('' if use_pg else ' (one GPU: no gradient exchange)'))


def dp_setup(world, rank, dev, backend='nccl'):
    """process group + per-rank seeds of the data-parallel run (one process per GPU; torch.distributed `nccl` IS RCCL on ROCm).  Returns use_pg."""
    import torch.distributed as dist
    use_pg = world > 1 or bool(os.environ.get('TFX_BENCH_FORCE_PG'))   # the env flag exercises the RCCL path on a 1-GPU box (world 1)
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        kw = dict(device_id=dev) if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return use_pg


def make_optimizer(model, use_pg):
    from transfusion_pytorch_amd.optim import FusedAdam
    opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
    opt.always_sync = use_pg
    opt.time_exchange = use_pg
    overlap = use_pg and os.environ.get('TFX_DP_OVERLAP', '1') != '0'
    if overlap:
        # the gradient all-reduce goes out in 4 layer groups DURING the backward; fp32 on the links like the reference's DDP (TFX_DP_BF16=1: bf16)
        opt.overlap_grad_sync(groups=4, exchange_dtype=torch.bfloat16 if os.environ.get('TFX_DP_BF16') == '1' else None)
    return opt, overlap


def exchange_prediction(numel, world):
    """what the per-step gradient exchange moves and how long it should take, so that the first real N > 1 run can be read against a number.
    Model: all-reduce of S bytes on the node's full xGMI mesh = reduce-scatter + all-gather, each sending S / N to every peer over its own link at once:
    t = 2 S / (N x link) ; link = 76.5 GB/s (one direction of a 153 GB/s link, the cautious figure) ... 153 GB/s.  With TFX_DP_OVERLAP (default) four fifths of it
    run under the backward; TFX_DP_BF16=1 halves S."""
    if world <= 1:
        return None
    bf16 = os.environ.get('TFX_DP_BF16') == '1'
    S = numel * (2 if bf16 else 4)
    t = lambda link: 2.0 * S / (world * link * 1e9) * 1e3
    return {'bytes_per_step': S, 'wire_dtype': 'bf16' if bf16 else 'fp32', 'predicted_ms_full_mesh': [round(t(153.0), 3), round(t(76.5), 3)],
            'model': 't = 2 S / (N x link), link = 153 ... 76.5 GB/s per direction; reduce-scatter + all-gather, every peer over its own xGMI link'}


def gather_ranks(elapsed, my_elapsed, exchange_ms, steps, world, use_pg, dev):
    """max-over-ranks wall time (the contract's `value` clock) + every rank's own clock and exposed exchange time"""
    import torch.distributed as dist
    per_rank = [my_elapsed / steps * 1e3]
    if use_pg:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = torch.tensor([my_elapsed / steps * 1e3, exchange_ms or 0.0], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x[0]) for x in allr]
        exchange_ms = max(float(x[1]) for x in allr)
        elapsed = float(t.item())
    return elapsed, per_rank, exchange_ms


def dry_run(args, world, rank, json_fd):
    """`--dry-run` (tests/test_dp_gloo.py, no GPU): everything of the N > 1 code path that is not a kernel launch - process group (gloo), identical
    init on every rank, per-rank seeds, `overlap_grad_sync`, the exchange driven by the REAL training plan's cut list on a rank-dependent fake
    gradient, the exchange section of `step()`, the closing barrier, max-over-ranks timing, `per_rank_ms_per_step` and the JSON assembly.
    `value` is null: nothing was computed."""
    import collections
    import torch.distributed as dist
    from transfusion_pytorch_amd.engine import Plan
    from transfusion_pytorch_amd.params import geglu_phys_to_ref_rows
    dev = torch.device('cpu')
    use_pg = dp_setup(world, rank, dev, backend='gloo')
    torch.manual_seed(0)
    model = build_model(args.dim, args.depth, args.two)
    opt, overlap = make_optimizer(model, use_pg)
    opt.time_exchange = False                                   # (HIP events)
    ps = model.store
    ps.grad = torch.zeros(ps.numel)
    ps.shadows = collections.defaultdict(lambda: torch.zeros(8, 8, dtype=torch.bfloat16))
    ps._map('geglu', geglu_phys_to_ref_rows(model.md.di, model.md.dip))
    plan = Plan(ps, b=2, n=64, I=4, R=({0: 8, 1: 4} if args.two else {0: 8}), training=True, dp_groups=getattr(model, '_dp_groups', 0))
    torch.manual_seed(7 + rank)
    same_init = float(ps.flat.double().sum())
    t0 = time.perf_counter()
    for k in range(args.steps):
        ps.grad.copy_(torch.full((ps.numel,), float(rank + 1)))
        if overlap:
            opt.reducer.check_fresh(); opt.reducer.begin()
            for _, first, last in plan.bwd_cuts:
                opt.reducer.group_ready(first, last)
        opt.sync_grads()
    my_elapsed = time.perf_counter() - t0
    if use_pg:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, per_rank, _ = gather_ranks(elapsed, my_elapsed, None, args.steps, world, use_pg, dev)
    want = float(sum(range(1, world + 1)))
    ok = bool((ps.grad == want).all())
    sums = torch.tensor([same_init], dtype=torch.float64)
    if use_pg:
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok &= bool(lo == hi)                                    # replicas start from identical weights
    if rank == 0:
        out = {'metric': f'train samples/sec, dim{args.dim} d{args.depth} seq1024 text+latent', 'value': None, 'unit': 'samples/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
               'dry_run': True, 'exchange_ok': ok, 'collective_launches_per_step': opt.reducer.last_launches if overlap else 1,
               'config': {'workload': workload_label(args.dim, args.depth, args.batch, world, use_pg, overlap, args.two), 'global_batch': world * args.batch,
                          'seq_len': 1024, 'parallelism': f'dp{world}'}, 'per_rank_ms_per_step': per_rank,
               'grad_exchange': exchange_prediction(ps.numel, world), 'bwd_cut_groups': len(plan.bwd_cuts)}
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_pg:
        dist.destroy_process_group()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start the N ranks ourselves, the way the reference's examples are started by
    `accelerate launch` (train_mnist.py:114-126) - one process per GPU under `torch.distributed.run` on 127.0.0.1 with a free port.  The ranks inherit
    this process's fd 1; rank 0 alone writes the JSON line there (every rank points its own fd 1 at stderr first), so the line arrives on the
    caller's stdout unchanged.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this pool's hosts (RCCL across processes needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] self-launch:', ' '.join(cmd), file=sys.stderr, flush=True)
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def time_other_config(c, dev, batch=64, steps=3, warmup=2):
    """one BASELINE config behind the timed region of the N = 1 line (`other_configs`): the 8-GPU model (config 3) and the two-modality 4-GPU model
    (config 4) at their per-GPU batch, full step (pack + fwd + bwd + clip + Adam), `steps` timed steps after `warmup`."""
    from transfusion_pytorch_amd.optim import FusedAdam
    cfg = CONFIGS[c]
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    m = build_model(cfg['dim'], cfg['depth'], cfg['two'], dev).train()
    opt = FusedAdam(m, lr=3e-4, max_grad_norm=0.5)
    gen = torch.Generator(device=dev).manual_seed(1234)
    batches = [make_batch(cfg['two'], batch, dev, gen) for _ in range(2)]
    def step(k):
        loss = m(batches[k % 2]); loss.backward(); opt.step(); opt.zero_grad(); return loss
    for k in range(warmup):
        step(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        loss = step(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    f = f_core_per_sample(d=cfg['dim'], D=cfg['depth'], inst_lens=inst_lens_of(cfg['two']))
    out = {'ms_per_step': dt * 1e3, 'samples_per_s': batch / dt, 'batch': batch, 'steps': steps, 'loss': float(loss.detach()),
           'model_flops_utilization': batch / dt * f / (PEAK_BF16_TFLOPS * 1e12), 'f_core_gflop_per_sample': f / 1e9,
           'peak_mem_gib': torch.cuda.max_memory_allocated() / 2**30}
    del m, opt, batches
    torch.cuda.empty_cache()
    return out


def parity_in_run(dev):
    """parity measured by THIS process: the `canon512` golden - outputs of the UNMODIFIED reference (fp32 CPU) on two canonical
    1024-token samples at the metric's model, tests/golden/canon512.pt - against the native step at the BENCH geometry (the two samples 32 times
    over = b 64 x 1024: the GEMM tilings, split counts and segment grids the timed step ran on).  The oracle package only supplies the case's
    deterministic inputs here (checker role, after the timed region)."""
    from oracle.cases import build_case
    from transfusion_pytorch_amd import Transfusion
    g = torch.load(os.path.join(ROOT, 'tests', 'golden', 'canon512.pt'), weights_only=False)
    cfg, sd, batch, times, noise = build_case('canon512')
    rep = 32
    m = Transfusion(num_text_tokens=cfg.num_text_tokens, dim_latent=cfg.dim_latents[0],
                    transformer=dict(dim=cfg.dim, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads), prob_uncond=0.)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    m._noise_override = {t: v.repeat(rep, 1).to(dev) for t, v in noise.items()}
    loss, bd = m(batch * rep, times=times.repeat(rep, 1), return_breakdown=True)
    loss.backward()
    torch.cuda.synchronize()
    plan = m._live[0]
    nt, rs = m._live_n_true, g['row_step']
    lg = plan.logits.view(plan.b, plan.n, -1)[:, :nt:rs, :cfg.vocab].float().cpu().double()
    lr = g['logits'].double().repeat(rep, 1, 1)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    top2 = lr.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 0.05
    same = lg.argmax(-1) == lr.argmax(-1)
    num = den = 0.
    for k, gn in g['grad_norms'].items():
        go = dict(m.named_parameters())[k].grad.detach().float().cpu()
        e = rel(go.double(), g['grads'][k].double()) if 'grads' in g else abs(float(go.double().norm()) - gn) / (gn + 1e-30)
        num += e * gn; den += gn
    ref = float(g['loss'])
    out = {'case': 'canon512 golden (unmodified reference, fp32 CPU) x32 = b 64 x 1024, dim512/depth8, measured in this run',
           'loss_native': float(loss.detach()), 'loss_reference': ref, 'loss_rel': abs(float(loss.detach()) - ref) / max(1., abs(ref)),
           'text_loss_delta': abs(float(bd.text) - float(g['text_loss'])), 'flow_loss_delta': abs(float(bd.flow[0]) - float(g['flow_losses'][0])),
           'logits_rel': rel(lg, lr), 'logits_rel_worst_sample': max(rel(lg[i], lr[i]) for i in range(lg.shape[0])),
           'argmax_unfiltered': float(same.float().mean()), 'argmax_margin_gt_0p05': float(same[safe].float().mean()), 'margin_gt_0p05_share': float(safe.float().mean()),
           'grad_mean_rel': num / den, 'gates': {'loss_rel': 1e-3, 'logits_rel': 1e-2, 'argmax_margin_gt_0p05': 1.0}}
    out['met'] = bool(out['loss_rel'] <= 1e-3 and out['logits_rel'] <= 1e-2 and out['argmax_margin_gt_0p05'] == 1.0)
    del m
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dim', type=int, default=None, help='default: the config\'s (512)')
    ap.add_argument('--depth', type=int, default=None, help='default: the config\'s (8)')
    ap.add_argument('--config', type=int, default=0, help='2 = dim512/depth8 (the metric, default), 3 = dim1024/depth24 (BASELINE config 3, the 8-GPU model), '
                    '4 = two modality types (384,192), dim768/depth16 (BASELINE config 4, the 4-GPU model)')
    ap.add_argument('--two-modality', dest='two', action='store_true', help='the two-modality sample layout of config 4 at --dim / --depth')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the config 3 / config 4 steps that ride in the N = 1 config-2 line (other_configs)')
    ap.add_argument('--no-parity', action='store_true', help='skip the in-run parity measurement (parity_met.bench_shape)')
    ap.add_argument('--roofline-kernel', default='tfx_gemm_nt')
    ap.add_argument('--roofline-every', type=int, default=5, help='bracket the roofline kernels with HIP events on every E-th timed step '
                    '(each bracketed launch costs two ~5 us event bubbles: 84 launches = ~0.9 ms on a step)')
    ap.add_argument('--family-steps', type=int, default=3, help='steps AFTER the timed region with every kernel family bracketed (roofline_by_family); 0 = skip')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-profile', action='store_true')
    ap.add_argument('--sample', action='store_true', help='time sample_many (SURVEY 8(d) config 5) instead of the training step')
    ap.add_argument('--no-sample', action='store_true', help='skip the sample_many timing (SURVEY 8(d) config 5) that rides in the N = 1 line')
    ap.add_argument('--ragged-steps', type=int, default=10, help='timed steps of the ragged steady state (every batch a new structure signature); 0 = skip')
    ap.add_argument('--dry-run', action='store_true', help='no GPU: drive the N > 1 host path (gloo) without launching kernels (tests/test_dp_gloo.py)')
    args = ap.parse_args()
    c = CONFIGS[args.config or 2]
    args.dim, args.depth, args.two = args.dim or c['dim'], args.depth or c['depth'], args.two or c['two']
    if args.sample:
        return bench_sample(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks'
    # keep stdout for the ONE JSON line: libraries print banners there (RCCL's version block at communicator init), so everything
    # else this process writes to fd 1 goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, world, rank, json_fd)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_pg = dp_setup(world, rank, dev)

    from transfusion_pytorch_amd import capi
    from transfusion_pytorch_amd.engine import Plan

    torch.manual_seed(0)                                      # identical init on every rank (replicas)
    model = build_model(args.dim, args.depth, args.two, dev).train()
    opt, overlap = make_optimizer(model, use_pg)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    torch.manual_seed(7 + rank)                               # per-rank noise / times / CFG streams
    # a FRESH batch for every step (train_toy.py:50-52 draws new data each iteration): generated before the timed region (the metric excludes
    # data generation), so no step ever sees token / latent values it has seen before.  The batches share one structure
    # signature (the canonical sample), i.e. the steady state the metric is quoted on; `structure_miss_ms` below prices a new signature.
    batches = [make_batch(args.two, args.batch, dev, gen) for _ in range(min(args.steps + args.warmup, 24))]
    state = {'k': 0}

    def step(batch=None):
        if batch is None:
            batch = batches[state['k'] % len(batches)]
            state['k'] += 1
        loss = model(batch)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()

    # instrument the dominant kernel family with HIP events inside the timed region
    events = []
    stream = torch.cuda.current_stream(dev).cuda_stream
    orig_run = Plan.run
    sampled = [False]
    families = [{args.roofline_kernel}]
    inst_lens = inst_lens_of(args.two)
    attn_flops = 4.0 * 64 * score_pairs(1024, inst_lens) * args.batch * 8      # per launch: 4 dh pairs per (sample, head), mask-aware
    def run(launches, stream_, lo=0, hi=None, graph=False):
        if sampled[0]:
            timed_run(orig_run, launches, stream_, lo, hi, families[0], events, attn_flops)
        else:
            orig_run(launches, stream_, lo, hi, graph=graph)
    Plan.run = staticmethod(run)
    n_sampled = 0
    host_t = 0.0
    prof = None
    if args.host_profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for k in range(args.steps):
        h0 = time.perf_counter()
        sampled[0] = k % max(args.roofline_every, 1) == 0
        n_sampled += int(sampled[0])
        # a bracketed step runs the weight-gradient GEMMs on the caller's stream too: the bracketed durations are then the kernels' own,
        # not theirs plus a side-stream neighbour's (the other steps overlap the two streams; both kinds count in `value`)
        capi.lib().tfx_set_single_stream(1 if sampled[0] else 0)
        loss = step()
        host_t += time.perf_counter() - h0
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof, stream=sys.stderr).sort_stats('tottime').print_stats(18)
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0                      # this rank's own clock, before it waits for the others
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sampled[0] = False
    capi.lib().tfx_set_single_stream(0)
    exchange_ms = opt.exchange_ms() if use_pg else None
    nt_events, events = events, []

    # ---- every kernel family, bracketed on `--family-steps` extra steps behind the timed region (same batches, one stream; NOT part of `value`: ~450
    # brackets cost ~2.5 ms of event bubbles per step): SURVEY 8(d) asks for the MFMA fraction of the attention + MLP kernels and the HBM rate
    # of the token-wise ones.  Also the host's issue time on an IDLE queue: `host_ms_per_step` above includes the time the issuing thread is
    # blocked behind a full queue (the step is GPU-bound), which says nothing about the host path itself.
    by_family, host_idle_ms, step_gpu_ms = None, None, None
    if args.family_steps > 0:                                  # (every rank: a step holds collectives)
        fams = {'tfx_gemm_nt', 'tfx_gemm_tn', 'tfx_attn_fwd', 'tfx_attn_bwd', 'hbm'}
        families[0] = fams
        capi.lib().tfx_set_single_stream(1)
        sampled[0] = True
        se = []
        for _ in range(args.family_steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); step(); e1.record(); se.append((e0, e1))
        torch.cuda.synchronize()
        sampled[0] = False
        capi.lib().tfx_set_single_stream(0)
        step_ms_bracketed = sum(a.elapsed_time(b) for a, b in se) / len(se)
        agg = {}
        for e0, e1, work, fam, _ in events:
            d = agg.setdefault(fam, [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3; d[1] += work; d[2] += 1
        by_family = []
        for fam in ('tfx_gemm_nt', 'tfx_gemm_tn', 'tfx_attn_fwd', 'tfx_attn_bwd', 'hbm'):
            if fam not in agg:
                continue
            tsec, work, cnt = agg[fam]
            hbm = fam == 'hbm'
            ach = work / tsec / 1e12
            by_family.append({'kernel': 'token-wise (AdaLN, QK-norm/RoPE, AttentionResidual, norms, CE)' if hbm else fam, 'bound': 'hbm' if hbm else 'mfma',
                              'achieved': ach, 'peak': 8.0 if hbm else PEAK_BF16_TFLOPS, 'unit': 'TB/s' if hbm else 'TFLOP/s',
                              'frac': ach / (8.0 if hbm else PEAK_BF16_TFLOPS), 'ms_per_step': tsec / args.family_steps * 1e3,
                              'launches_per_step': cnt / args.family_steps,
                              ('algorithmic_gbyte_per_step' if hbm else 'algorithmic_gflop_per_step'): work / args.family_steps / 1e9})
            if fam == 'tfx_attn_bwd' and os.environ.get('TFX_ATTN_QKNR', '1') != '0':
                by_family[-1]['note'] = ('since round 5 these launches also run the backward of QK-RMSNorm + RoPE in their epilogues (token-wise work, ~0.6 ms per step as its '
                                         'own launches through round 4: +0.4 ms here, -0.65 ms in the token-wise family); the algorithmic flops are the attention\'s alone')
        if os.environ.get('TFX_BENCH_SHAPES'):                   # per GEMM shape (M, N, K, epilogue): launches / step, us / launch, TFLOP/s -> stderr
            bys = {}
            for e0, e1, work, fam, shape in events:
                if shape is not None:
                    d = bys.setdefault((fam,) + shape, [0.0, 0.0, 0])
                    d[0] += e0.elapsed_time(e1) * 1e-3; d[1] += work; d[2] += 1
            for key, (tsec, work, cnt) in sorted(bys.items(), key=lambda kv: -kv[1][0]):
                print(f'[shape] {key[0][4:]:8s} M={key[1]:6d} N={key[2]:5d} K={key[3]:5d} epi={key[4]:2d}  x{cnt / args.family_steps:5.1f}/step  {tsec / cnt * 1e6:7.1f} us  '
                      f'{work / tsec / 1e12:7.1f} TF/s  {tsec / args.family_steps * 1e3:6.3f} ms/step', file=sys.stderr)
        events = []
        mf = [f for f in by_family if f['bound'] == 'mfma']
        agg_flops = sum(f['algorithmic_gflop_per_step'] for f in mf) * 1e9
        agg_t = sum(f['ms_per_step'] for f in mf) * 1e-3
        step_gpu_ms = {'bracketed_step_ms': step_ms_bracketed, 'families_ms': sum(f['ms_per_step'] for f in by_family),
                       'note': 'one stream, every family bracketed (event bubbles included in bracketed_step_ms); the rest = optimizer, casts, losses, small launches'}
        aggregate = {'attn_mlp_tflops': agg_flops / agg_t / 1e12, 'frac': agg_flops / agg_t / 1e12 / PEAK_BF16_TFLOPS, 'ms_per_step': agg_t * 1e3,
                     'what': 'all GEMMs (NT + weight-gradient TN) + attention forward + backward: algorithmic flops / their summed kernel time'}
        # host issue time with nothing queued: sync, then time until step() RETURNS (launches cannot block on a full queue)
        hs = []
        for _ in range(3):
            torch.cuda.synchronize(); h0 = time.perf_counter(); step(); hs.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        host_idle_ms = sorted(hs)[1] * 1e3
    Plan.run = orig_run
    # ragged steady state: EVERY batch has a structure the model has never seen (new signature, its own packed lengths) - the host scan, token maps,
    # segments and index uploads are paid on every step, as on a real corpus; steps are issued back to back (no sync in between), so whatever of
    # that host work hides behind the previous step's GPU work is hidden here too.  Two warm-up steps create the plan of the padded length.
    ragged_ms = None
    if args.ragged_steps > 0 and not args.two:
        rb = [ragged_batch(args.batch, dev, gen, seed=1000 * rank + k) for k in range(args.ragged_steps + 2)]
        step(rb[0]); step(rb[1])
        torch.cuda.synchronize(); tr0 = time.perf_counter()
        for batch in rb[2:]:
            step(batch)
        torch.cuda.synchronize(); ragged_ms = (time.perf_counter() - tr0) / args.ragged_steps * 1e3
        del rb
    # one step on a structure the model has never seen (same packed length, different text / latent placement): host structure scan, index
    # uploads and - at a new padded length - a new plan are paid here and nowhere in `value`
    miss = two_modality_batch(args.batch, dev, gen, text_len=24, last_text_len=55) if args.two else canonical_batch(args.batch, dev, gen, text_len=23, last_text_len=54)
    # the ragged phase above leaves ~50 k dead tensor objects behind; without this collection CPython's generation-2 pass (46 ms on this heap,
    # tools/prof_miss.py) lands inside the one step timed here and is reported as structure work (107 vs 40 ms)
    import gc
    gc.collect()
    torch.cuda.synchronize(); tm0 = time.perf_counter()
    step(miss)
    torch.cuda.synchronize(); structure_miss_ms = (time.perf_counter() - tm0) * 1e3

    elapsed, per_rank, exchange_ms = gather_ranks(elapsed, my_elapsed, exchange_ms, args.steps, world, use_pg, dev)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed

    if rank == 0:
        kt = sum(e0.elapsed_time(e1) for e0, e1, _, _, _ in nt_events) * 1e-3
        kf = sum(f for _, _, f, _, _ in nt_events)
        achieved = kf / kt / 1e12 if kt > 0 else 0.0
        n_launch = len(nt_events)
        fcore = f_core_per_sample(d=args.dim, D=args.depth, inst_lens=inst_lens)
        # HBM bytes per launch of the roofline kernel family: rocprofv3 PMC passes of this same command, committed under
        # profiles/ (tools/pmc_traffic.sh; FETCH_SIZE x2 gfx950 correction applied there) - counters cannot be read in-process
        traffic, traffic_src = None, None
        pdir = os.path.join(ROOT, 'profiles')
        for tname in ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json'):
            tj = os.path.join(pdir, tname)
            if os.path.exists(tj) and (args.batch, args.dim, args.depth, args.two) == (64, 512, 8, False):
                tr = json.load(open(tj))
                if tr.get('kernel_family') == args.roofline_kernel:
                    traffic, traffic_src = tr['bytes_per_launch'], tr['source']
                    break
        measured_peak = measure_mfma_peak(dev)
        out = {
            'metric': f'train samples/sec, dim{args.dim} d{args.depth} seq1024 text+latent', 'value': value, 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': workload_label(args.dim, args.depth, args.batch, world, use_pg, overlap, args.two),
                       'global_batch': world * args.batch, 'seq_len': 1024, 'parallelism': f'dp{world}'},
            'loss': float(loss.detach()),
            'model_flops_utilization': value / world * fcore / (PEAK_BF16_TFLOPS * 1e12),
            'f_core_gflop_per_sample': fcore / 1e9,
            'host_ms_per_step': host_t / args.steps * 1e3,     # wall time of the issuing thread per step INCLUDING its waits behind the full queue (GPU-bound step)
            'host_issue_ms_idle_queue': host_idle_ms,          # the same path with nothing queued (sync before the step): scan + pointer gather + ~440 launches
            'structure_miss_ms': structure_miss_ms,
            'ragged_ms_per_step': ragged_ms,                    # every batch a never-seen structure, steps back to back (rank 0)
            'per_rank_ms_per_step': per_rank,                   # each rank's own clock over the timed steps (before the closing barrier)
            'grad_exchange_exposed_ms': exchange_ms,            # max over ranks: compute-stream time per step inside the exchange section (tail collective + waits)
            'grad_exchange': exchange_prediction(model.store.numel, world),       # N > 1: bytes on the wire per step and the predicted xGMI time (null at N = 1)
            'fresh_batch_every_step': True,
            'roofline': {'bound': 'mfma', 'kernel': args.roofline_kernel, 'achieved': achieved, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_BF16_TFLOPS, 'traffic': traffic, 'traffic_unit': 'bytes/launch', 'traffic_source': traffic_src, 'launches_per_step': n_launch / max(n_sampled, 1), 'sampled_steps': n_sampled,
                         'note': 'bracketed on every --roofline-every-th timed step; those steps replay on one stream (no side-stream overlap) so the durations are the kernels own',
                         'avg_launch_us': kt / max(n_launch, 1) * 1e6, 'algorithmic_gflop_per_step': kf / max(n_sampled, 1) / 1e9,
                         'measured_peak': measured_peak, 'measured_peak_frac': achieved / measured_peak,
                         'measured_peak_source': 'this run, behind the timed region: tfx_mfma_peak_probe - register-resident v_mfma_f32_32x32x16_bf16 loop, random operands, two blocks per CU (the power-limited clock of this box)'},
        }
        if by_family is not None:
            out['roofline_by_family'] = by_family
            out['aggregate_attn_mlp'] = aggregate
            out['step_kernel_time'] = step_gpu_ms
        # parity: the suite-wide figures (every golden case, measured by tests/test_model_gpu.py on MI355X and committed under profiles/) are QUOTED;
        # `bench_shape` is MEASURED by this process on the bench geometry against the reference's golden (parity_in_run)
        for pname in ('r06_parity.json', 'r05_parity.json', 'r04_parity.json'):
            pj = os.path.join(pdir, pname)
            if os.path.exists(pj):
                out['parity_met'] = json.load(open(pj))
                out['parity_met']['suite_figures_source'] = f'profiles/{pname} (quoted; measured by the GPU test suite)'
                break
        is_metric_cfg = (args.batch, args.dim, args.depth, args.two) == (64, 512, 8, False)
        model._plans, model._struct_cache = {}, {}              # the riders below build their own models: drop the training plans first
        torch.cuda.empty_cache()
        if world == 1 and not args.no_parity:
            try:
                out.setdefault('parity_met', {})['bench_shape'] = parity_in_run(dev)
                out['parity_met']['bench_shape_source'] = 'measured in this bench.py run (parity_in_run), after the timed region'
            except Exception as e:                               # a rider never costs the line
                out.setdefault('parity_met', {})['bench_shape_error'] = repr(e)
        if world == 1 and not args.no_other_configs and is_metric_cfg:
            # the 8-GPU model (config 3) and the two-modality 4-GPU model (config 4) at their per-GPU batch, 3 timed steps each, one GPU: the
            # per-GPU work of those DP runs is exactly this (weak scaling, one all-reduce per step on top)
            oc = {}
            for c in (3, 4):
                try:
                    r = time_other_config(c, dev)
                    oc[f'config{c}_b64_ms_per_step'] = r['ms_per_step']
                    oc[f'config{c}'] = r
                except Exception as e:
                    oc[f'config{c}_error'] = repr(e)
            out['other_configs'] = oc
        if world == 1 and not args.no_sample and is_metric_cfg:
            # SURVEY 8(d) config 5 rides in the same line (outside the timed region)
            runs = time_sample_many(dev)
            fx = os.path.join(ROOT, 'tests', 'golden', 'reference_sampling_time.json')
            out['sample_many'] = {'config5_forced_s': runs['config5_forced']['seconds'], 'config5_free_s': runs['config5']['seconds'],
                                  'config5_forced_repeat_call_s': runs['config5_forced']['seconds_repeat_call'], 'config5_free_repeat_call_s': runs['config5']['seconds_repeat_call'],
                                  'config5_forced_tokens': runs['config5_forced']['tokens_returned'], 'config5_free_tokens': runs['config5']['tokens_returned'],
                                  'config5_forced_modalities': runs['config5_forced']['modality_instances'], 'config5_free_modalities': runs['config5']['modality_instances'],
                                  'reduced_forced_s': runs['reduced_forced']['seconds'], 'reduced_free_s': runs['reduced']['seconds'],
                                  'workload': 'dim1024 depth24, 64 mixed prompts, max_length 256, 16 ODE grid points, cfg 3, greedy; reduced = dim512 depth8, 8 prompts, max_length 32',
                                  'cpu_reference_reduced': json.load(open(fx)) if os.path.exists(fx) else None,
                                  'cpu_reference_note': 'the UNMODIFIED reference on the reduced configuration, timed in the build container (the reference tree is not on the GPU box; the oracle restatement has no sampler) - NOT a same-box measurement'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_pg:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
