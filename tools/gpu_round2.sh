#!/bin/bash
# GPU tests, bench, the RCCL path at world size 1 under torch.distributed.run, decode bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','host_ms_per_step')}, d['roofline']['achieved'])"
TFX_BENCH_FORCE_PG=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/pg.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced-PG', {k:d[k] for k in ('value','ms_per_step','host_ms_per_step','n_gpus')})" || tail -20 gpurun_out/pg.err
python tools/bench_sample.py
