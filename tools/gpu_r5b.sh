#!/bin/bash
# round 5, call B: the two fixed tests + sampling equivalence print, same-box A/B of the round-4 library vs the new one (step time, families),
# per-shape GEMM table, RCCL world-1 bench on the (default) public-API exchange
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -s -k "table_range or overlapped_exchange_world1 or sample_one_equals or geglu" > gpurun_out/r05b_pytest.log 2>&1
grep -n "passed\|failed\|sample_one vs\|^E  " gpurun_out/r05b_pytest.log | head -20
AB_FAMILY_STEPS=3 TFX_AB="TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_r04.so;TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_hip.so" bash tools/gpu_run.sh r05b ab 2>&1 | tee gpurun_out/r05b_ab.txt
TFX_BENCH_SHAPES=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > gpurun_out/r05b_shapes.json 2> gpurun_out/r05b_shapes.txt
grep "\[shape\]" gpurun_out/r05b_shapes.txt | head -40
TFX_BENCH_FORCE_PG=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity --family-steps 0 > gpurun_out/r05b_pg.json 2> gpurun_out/r05b_pg.err
python -c "
import json; d=json.loads(open('gpurun_out/r05b_pg.json').read().strip().splitlines()[-1]); print('forced-PG world 1:', d['ms_per_step'], d.get('grad_exchange_exposed_ms'), d['config']['workload'][-160:])" || tail -5 gpurun_out/r05b_pg.err
