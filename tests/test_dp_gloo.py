"""N > 1 path on CPU: world_size-2 `gloo` run of the data-parallel gradient exchange (the ONE collective of a step:
all-reduce of the flat gradient buffer), launched exactly like bench.py is (env:// rendezvous on 127.0.0.1)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.optim import FusedAdam
    torch.manual_seed(0)                                        # identical replicas
    m = Transfusion(num_text_tokens=16, dim_latent=32, transformer=dict(dim=64, depth=2, heads=1))
    ps = m.store
    ps.grad = torch.zeros(ps.numel)
    for i, (name, p) in enumerate(ps.params.items()):
        ps.grad_view(name).fill_(float(rank + 1) * (i + 1))     # rank-dependent "gradients"
    opt = FusedAdam(m, lr=1e-3)
    w = opt.sync_grads()
    ok = w == world
    for i, name in enumerate(ps.params):
        ok &= bool((ps.grad_view(name) == float(sum(r + 1 for r in range(world))) * (i + 1)).all())
    flat0 = ps.flat.clone()
    dist.all_reduce(flat0, op=dist.ReduceOp.MAX)
    ok &= bool(torch.equal(flat0, ps.flat))                     # replicas hold identical parameters
    q.put((rank, ok))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_overlap(rank, world, port, q):
    """the overlapped exchange (optim.GradReducer): layer groups all-reduced in backward order, tail at the end - every element exactly once,
    the result equal to ONE all-reduce of the whole buffer, and a (torch-emulated) clip + Adam step on the mean gradient leaves identical
    parameters on every rank, equal to the single-process step on the mean gradient."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.optim import GradReducer
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=16, dim_latent=(32, 8), transformer=dict(dim=64, depth=5, heads=1))      # odd depth: uneven groups
    ps = m.store
    gen = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(ps.numel, generator=gen)
    ps.grad = local.clone()
    red = GradReducer(m, None, groups=3)
    red.begin()
    per = red.per
    ok = True
    for first in range(((5 - 1) // per) * per, -1, -per):                     # the cut order of engine.Plan: last group first
        red.group_ready(first, min(first + per, 5) - 1)
    red.finish()
    expect = local.clone()
    dist.all_reduce(expect, op=dist.ReduceOp.SUM)
    ok &= bool(torch.equal(ps.grad, expect))
    # step on the mean gradient (train_toy.py:55-57 semantics, emulated with torch on CPU): identical on every rank
    g = ps.grad / world
    g = g * min(1., 0.5 / (float(g.norm()) + 1e-6))
    new = ps.flat.detach() - 3e-4 * g / (g.abs() + 1e-8)                      # first Adam step: m / sqrt(v) = sign(g)
    ref = new.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.MAX)
    ok &= bool(torch.equal(ref, new))
    all_local = [torch.randn(ps.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    gm = sum(all_local) / world
    gm = gm * min(1., 0.5 / (float(gm.norm()) + 1e-6))
    ok &= bool(torch.allclose(new, ps.flat.detach() - 3e-4 * gm / (gm.abs() + 1e-8), rtol=0, atol=1e-6))
    # the same exchange in bf16 (half the bytes on the links): every element still reduced exactly once, the sum within bf16 rounding of the fp32 one
    ps.grad = local.clone()
    red16 = GradReducer(m, None, groups=3, exchange_dtype=torch.bfloat16)
    red16.begin()
    for first in range(((5 - 1) // per) * per, -1, -per):
        red16.group_ready(first, min(first + per, 5) - 1)
    red16.finish()
    ok &= bool(((ps.grad - expect).abs() <= 2 ** -7 * (sum(a.abs() for a in all_local)) + 1e-6).all())
    ok &= not bool(torch.equal(ps.grad, expect))                              # (it really travelled in bf16)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_gradient_exchange_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _cpu_plan(depth=5, groups=3):
    """a REAL training Plan (engine.Plan: the launch lists the GPU replays) built on the CPU: nothing is launched, but the lists, their order and
    the cut points `bwd_cuts` are the product's own"""
    import collections
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.engine import Plan
    from transfusion_pytorch_amd.params import geglu_phys_to_ref_rows
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=16, dim_latent=(32, 8), transformer=dict(dim=64, depth=depth, heads=1))
    ps = m.store
    ps.grad = torch.zeros(ps.numel)
    ps.shadows = collections.defaultdict(lambda: torch.zeros(8, 8, dtype=torch.bfloat16))      # bf16 kernel-layout copies: only their addresses enter the lists
    ps._map('geglu', geglu_phys_to_ref_rows(m.md.di, m.md.dip))
    plan = Plan(ps, b=2, n=64, I=4, R={0: 8, 1: 4}, training=True, dp_groups=groups)
    return m, ps, plan


def _grad_pointers(item, lo, hi):
    """every address inside the flat gradient buffer that a launch item carries (struct fields or positional arguments)"""
    import ctypes
    fn, a = item
    vals = []
    if isinstance(a, (tuple, list)):
        vals = [v for v in a if isinstance(v, int)]
    elif isinstance(a, ctypes.Structure):
        vals = [getattr(a, f) for f, t in a._fields_ if t is ctypes.c_void_p and getattr(a, f)]
        # argument structs passed by address (fused launches) ride in the positional form above; nested structs are kept alive in plan._keep
    return [v for v in vals if lo <= v < hi]


def test_real_plan_cuts_leave_exchanged_ranges_untouched():
    """with the overlapped exchange a layer group's gradient ranges go out at `Plan.bwd_cuts`; nothing the backward list launches
    AFTER a cut may write into a range that left at or before it.  Checked on the product's own launch list (built on the CPU), including the
    structs the fused launches reference by address and the AttentionResidual source table (its gradients land in the tail by construction)."""
    import ctypes
    from transfusion_pytorch_amd import capi
    from transfusion_pytorch_amd.optim import GradReducer
    m, ps, plan = _cpu_plan()
    red = GradReducer(m, None, groups=3)
    base = ps.grad.data_ptr(); lo, hi = base, base + 4 * ps.numel
    assert [c[1:] for c in plan.bwd_cuts] == [(4, 4), (2, 3), (0, 1)]              # depth 5 in 3 groups of <= 2 layers, last group first
    # structs referenced by address from positional launches (tfx_attnres_pull_bwd, tfx_adaln_pre_post_bwd, ...)
    by_addr = {ctypes.addressof(k): k for k in getattr(plan, '_keep', []) if isinstance(k, ctypes.Structure)}
    def pointers(idx):
        item = plan.bwd[idx]
        out = _grad_pointers(item, lo, hi)
        fn, a = item
        if isinstance(a, (tuple, list)):
            for v in a:
                if isinstance(v, int) and v in by_addr:
                    out += _grad_pointers(('', by_addr[v]), lo, hi)
        return out
    sent = []
    cuts = {idx: (first, last) for idx, first, last in plan.bwd_cuts}
    n_written = 0
    for idx in range(len(plan.bwd)):
        if idx in cuts:
            sent += red.ranges(*cuts[idx])
        for ptr in pointers(idx):
            off = (ptr - base) // 4
            n_written += 1
            assert not any(a <= off < b for a, b in sent), (idx, plan.bwd[idx][0], off, sent)
    assert n_written >= 12 * 5                                                      # (the scan really saw the weight / bias / gain gradients: >= 12 per layer)
    # the AttentionResidual gradients are written when the backward reaches hidden 0 (tfx_attnres_finish): they must sit in the tail
    tab = bytes(plan._src_tab.numpy().tobytes())
    SRC = capi.STRUCTS['tfx_attnres_src']
    recs = (SRC * 5).from_buffer_copy(tab)
    covered = [r for first in (4, 2, 0) for r in red.ranges(first, min(first + 1, 4))]
    for j in range(5):
        for ptr in (recs[j].dgamma, recs[j].dpq):
            off = (ptr - base) // 4
            assert lo <= ptr < hi and not any(a <= off < b for a, b in covered), j


def _worker_plan_overlap(rank, world, port, q):
    """the exchange driven by a REAL plan's cut list (not a hand-derived order): groups + 1 collective launches, every element exactly once, the
    result equal to one all-reduce; then the step on the mean gradient is identical on every rank"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from transfusion_pytorch_amd.optim import GradReducer
    m, ps, plan = _cpu_plan()
    local = torch.randn(ps.numel, generator=torch.Generator().manual_seed(100 + rank))
    ok = True
    os.environ['TFX_DP_COALESCE'] = '1'                         # this test counts the COALESCED exchange's launches (groups + 1); the public-API default (one
                                                                # collective per range, round 5) is counted in the next test and in the bench dry run
    for dt in (None, torch.bfloat16):
        ps.grad.copy_(local)
        red = GradReducer(m, None, groups=3, exchange_dtype=dt)
        red.begin()
        for _, first, last in plan.bwd_cuts:                    # the order the backward replay hands the groups over (transfusion._native_backward)
            red.group_ready(first, last)
        try:
            red.check_fresh(); ok = False                       # a second backward before the step must be refused
        except RuntimeError:
            pass
        red.finish()
        ok &= red.launches == 3 + 1
        expect = local.clone()
        dist.all_reduce(expect, op=dist.ReduceOp.SUM)
        if dt is None:
            ok &= bool(torch.equal(ps.grad, expect))
        else:
            both = [torch.randn(ps.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
            ok &= bool(((ps.grad - expect).abs() <= 2 ** -7 * sum(a.abs() for a in both) + 1e-6).all())
    g = expect / world
    g = g * min(1., 0.5 / (float(g.norm()) + 1e-6))
    new = ps.flat.detach() - 3e-4 * g / (g.abs() + 1e-8)
    ref = new.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.MAX)
    ok &= bool(torch.equal(ref, new))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_exchange_follows_real_plan_cuts_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plan_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_ext(rank, world, port, q):
    """external parameters (positional-embedding MLPs): a rank whose batch produced NO gradient for them still takes part in the same collective
    (zeros), and ends up with the summed gradient - replicas cannot diverge on ragged multi-modal data"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from transfusion_pytorch_amd import Transfusion
    from transfusion_pytorch_amd.optim import FusedAdam
    torch.manual_seed(0)
    m = Transfusion(num_text_tokens=16, dim_latent=32, modality_default_shape=(2, 2), add_pos_emb=True, modality_num_dim=2, transformer=dict(dim=64, depth=2, heads=1))
    ps = m.store
    ps.grad = torch.zeros(ps.numel)
    opt = FusedAdam(m, lr=1e-3)
    ok = len(opt.ext_params) > 0
    if rank == 0:                                               # only rank 0 saw the modality
        for i, p in enumerate(opt.ext_params):
            p.grad = torch.full_like(p, float(i + 1))
    opt.sync_grads()
    for i, p in enumerate(opt.ext_params):
        ok &= p.grad is not None and bool((p.grad == float(i + 1)).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_external_parameter_gradients_with_missing_modalities_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ext, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_world1_gates(rank, world, port, q):
    """ world size 1 with torch.distributed initialised (`torchrun --nproc-per-node 1`) and the DEFAULT `always_sync`: the backward takes
    the overlapped path (it only asks whether a process group exists), so the optimizer's exchange section must consume what it sent - wait for
    the handles, reduce the tail, re-arm - or the next step's backward is refused.  Two steps, driven by a real plan's cut list; then the
    accumulation contract of `no_sync()` and the public-API (non-coalesced) exchange."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from transfusion_pytorch_amd.optim import FusedAdam
    m, ps, plan = _cpu_plan()
    opt = FusedAdam(m, lr=1e-3)
    opt.overlap_grad_sync(groups=3)
    red = opt.reducer
    ok = opt.always_sync is False
    local = torch.randn(ps.numel, generator=torch.Generator().manual_seed(7))
    for step in range(2):
        ps.grad.copy_(local)
        red.check_fresh()                                       # what transfusion._native_backward does before it replays the list
        red.begin()
        for _, first, last in plan.bwd_cuts:
            red.group_ready(first, last)
        ok &= red.exchanged
        opt.sync_grads()                                        # the exchange section of step()
        ok &= (not red.exchanged) and not red.handles and not red.done
        ok &= bool(torch.equal(ps.grad, local))                 # world 1: the sum is the rank's own gradient
    # no_sync: the flag the backward reads; nesting restores it
    with opt.no_sync():
        ok &= red.defer
        with opt.no_sync():
            ok &= red.defer
        ok &= red.defer
    ok &= not red.defer
    # public API only (the default; TFX_DP_COALESCE unset or 0): one collective per range, every element still exactly once
    os.environ.pop('TFX_DP_COALESCE', None)
    ps.grad.copy_(local)
    red.begin()
    for _, first, last in plan.bwd_cuts:
        red.group_ready(first, last)
    red.finish()
    ok &= red.launches == 2 * 3 + 2 and bool(torch.equal(ps.grad, local))      # 3 groups x 2 ranges + the tail's 2 ranges (coalesced: 3 + 1)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_world1_default_gates_no_sync_and_public_api_exchange():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    p = ctx.Process(target=_worker_world1_gates, args=(0, 1, port, q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=60)
    assert res == (0, True)


def test_bench_two_rank_host_path_dry_run():
    """the driver's own N = 2 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...`) with `--dry-run`: gloo instead of RCCL and no kernel launches, everything else of bench.py's
    N > 1 path - RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, identical initial weights on both ranks, `overlap_grad_sync`, the exchange
    driven by the real plan's cut list (5 collective launches per step, every element summed once), the closing barrier, max-over-ranks
    timing, one JSON line from rank 0 with `per_rank_ms_per_step` of both ranks and a self-describing workload label."""
    import json
    import subprocess
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port']
    # the first case is the PLAIN command (`python bench.py --gpus 2 ...`, no launcher): bench.py starts its two ranks itself
    for extra_env, want_launches, plain in ((dict(), 10, True), (dict(TFX_DP_COALESCE='1'), 5, False), (dict(TFX_DP_OVERLAP='0'), 1, False)):
        cmd = ([sys.executable] if plain else launcher + [str(port)]) + [
               os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run', '--dim', '128', '--depth', '8']
        r = subprocess.run(cmd, env=dict(env, **extra_env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1, r.stdout                          # ONE JSON line, from rank 0
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['dry_run'] and d['value'] is None and d['exchange_ok']
        assert d['collective_launches_per_step'] == want_launches, d
        assert len(d['per_rank_ms_per_step']) == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 128
        port = _free_port()
    # a driver line at the 8-GPU model's dimensions names BASELINE config 3
    sys.path.insert(0, ROOT)
    import bench
    c3 = bench.CONFIGS[3]
    lab = bench.workload_label(c3['dim'], c3['depth'], 64, 8, True, True, c3['two'])
    assert 'BASELINE config 3' in lab and 'dim=1024 depth=24' in lab and 'collective launches' in lab
    assert 'BASELINE config 2' in bench.workload_label(512, 8, 64, 1, False, False)
    # BASELINE config 4 (two modality types, dim768 / depth16) is a named bench configuration: plain command, self-launched, dry run
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', '4', '--steps', '2', '--warmup', '1', '--dry-run', '--depth', '2'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['n_gpus'] == 2 and d['exchange_ok'] and 'dim_latent=(384,192)' in d['config']['workload'] and 'dim=768' in d['config']['workload']


def test_bench_eight_rank_self_launch_dry_run_at_the_depth_of_config_3():
    """VERDICT r5 item 9 (multi-GPU pre-flight, no hardware): the PLAIN `python bench.py --gpus 8 --config 3 --dry-run` - bench.py starts its eight ranks itself -
    with the depth-24 training plan's cut list driving the overlapped exchange on every rank (the model dimension is reduced to keep eight replicas inside this
    container's memory: the plan's structure - 24 layers, 4 layer groups + tail - is config 3's), and `--gpus 4 --config 4` (two modality types).  One JSON line
    from rank 0 with all per-rank clocks, the exchange sum verified on every element, and the bytes / predicted xGMI time of the per-step exchange."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    for gpus, cfg, dim, frag in ((8, 3, 128, 'depth=24'), (4, 4, 192, 'dim_latent=(384,192)')):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--config', str(cfg), '--dim', str(dim), '--steps', '2', '--warmup', '1',
                            '--dry-run'], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d['n_gpus'] == gpus and d['dry_run'] and d['exchange_ok'] and d['value'] is None
        assert len(d['per_rank_ms_per_step']) == gpus and d['config']['parallelism'] == f'dp{gpus}' and d['config']['global_batch'] == 64 * gpus
        assert frag in d['config']['workload'], d['config']['workload']
        assert d['collective_launches_per_step'] == 10 and d['bwd_cut_groups'] == 4           # 4 layer groups + tail, two contiguous ranges each
        ge = d['grad_exchange']
        assert ge['wire_dtype'] == 'fp32' and ge['bytes_per_step'] % 4 == 0 and 0 < ge['predicted_ms_full_mesh'][0] < ge['predicted_ms_full_mesh'][1]
    # at the real dimensions: config 3 exchanges ~3.5 GB of fp32 per step, config 2 ~0.32 GB (pure host arithmetic)
    sys.path.insert(0, ROOT)
    import bench
    p3 = bench.exchange_prediction(874_000_000, 8)
    assert 3.4e9 < p3['bytes_per_step'] < 3.6e9 and 5.0 < p3['predicted_ms_full_mesh'][0] < p3['predicted_ms_full_mesh'][1] < 13.0
    assert bench.exchange_prediction(79_545_968, 1) is None
