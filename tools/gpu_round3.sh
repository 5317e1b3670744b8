#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/bench_sample.py 2>&1 | grep -v amdgpu
python tools/bench_configs.py 5 2>&1 | grep -v amdgpu
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','host_ms_per_step')}, d['roofline']['achieved'])"
