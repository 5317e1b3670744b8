"""can two RCCL ranks share ONE device on this box?  (the GPU box has a single MI355X: if yes, the N = 2 path of bench.py can run there with real RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/rccl_one_device_probe.py"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
try:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    x = torch.full((1 << 20,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print(f'rank {rank}: all_reduce over {world} ranks on one device -> {float(x[0])} (expected {world * (world + 1) / 2})', flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f'rank {rank}: FAILED {type(e).__name__}: {str(e)[:300]}', flush=True)
    sys.exit(3)
