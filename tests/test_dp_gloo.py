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
