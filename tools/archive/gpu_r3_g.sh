#!/bin/bash
# round-3 call G: all GPU tests (bucketed training plans, cache_kv=False paths, pos + clean golden), bench with the ragged steady state
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ragged-steps 0 --no-sample > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_line.json')); print({k: d[k] for k in ('value','ms_per_step','host_ms_per_step','structure_miss_ms','ragged_ms_per_step','per_rank_ms_per_step','grad_exchange_exposed_ms')}); print(d['config']['workload'][-120:])"
TFX_PLAN_BUCKETS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --ragged-steps 0 --no-sample --ragged-steps 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TFX_PLAN_BUCKETS=0', {k: d[k] for k in ('ms_per_step','ragged_ms_per_step')})"
TFX_BENCH_FORCE_PG=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --ragged-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('world-1 RCCL', {k: d[k] for k in ('ms_per_step','per_rank_ms_per_step','grad_exchange_exposed_ms')}); print(d['config']['workload'][-150:])"
tail -3 gpurun_out/${TAG}_bench.err
