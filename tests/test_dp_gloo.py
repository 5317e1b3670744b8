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
