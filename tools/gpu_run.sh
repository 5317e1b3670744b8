#!/bin/bash
# ONE parameterised GPU round trip (replaces the gpu_r3_*.sh family).  usage (on the GPU box, via gpurun):
#   bash tools/gpu_run.sh <tag> <mode>[,<mode>...] [pytest -k expression]
# modes: tests   all GPU tests (or -k expr)           -> gpurun_out/<tag>_pytest.log
#        bench   bench.py, 20 steps                    -> gpurun_out/<tag>_bench.json
#        quick   bench.py 10 steps without the riders  -> gpurun_out/<tag>_quick.json
#        prof    rocprofv3 kernel summary of 4 steps on one stream -> gpurun_out/<tag>_kernel_summary.txt
#        stats   rocprofv3 --kernel-trace --stats of the bench command (two streams) -> gpurun_out/<tag>_kernel_stats.csv
#        ab      A/B of env settings given in $TFX_AB (";"-separated "K=V K=V" groups), quick bench each, 2 rounds interleaved
TAG=${1:-run}; MODES=${2:-tests,quick}; KEXPR=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
QUICK="--steps 10 --warmup 3 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ('value','ms_per_step','host_ms_per_step','host_issue_ms_idle_queue','loss') if k in d}, 'NT', round(d['roofline']['achieved'],1), [(f['kernel'][:12], round(f['achieved'],2), round(f['ms_per_step'],2)) for f in d.get('roofline_by_family',[])], d.get('aggregate_attn_mlp',{}).get('frac'))"; }
for M in ${MODES//,/ }; do
  case $M in
    tests) if [ -n "$KEXPR" ]; then python -m pytest tests -m gpu -q -x -k "$KEXPR" > gpurun_out/${TAG}_pytest.log 2>&1; else python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; fi
           grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -20 | cut -c1-300;;
    bench) python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; line < gpurun_out/${TAG}_bench.json;;
    quick) python bench.py $QUICK > gpurun_out/${TAG}_quick.json 2> gpurun_out/${TAG}_quick.err; line < gpurun_out/${TAG}_quick.json || tail -5 gpurun_out/${TAG}_quick.err;;
    prof)  (cd /tmp && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -o p -- python $R/bench.py --steps 3 --warmup 1 --family-steps 0 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/prof_$TAG.log 2>&1)
           python tools/prof_summary.py /tmp/prof_$TAG/p_kernel_trace.csv --steps 5 > gpurun_out/${TAG}_kernel_summary.txt; head -${PROF_LINES:-32} gpurun_out/${TAG}_kernel_summary.txt | cut -c1-150;;
    stats) (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/stats_$TAG.log 2>&1)
           cp /tmp/stats_$TAG/p_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null; head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160;;
    ab)    IFS=';' read -ra GROUPS_ <<< "$TFX_AB"
           for i in 1 2; do for G in "${GROUPS_[@]}"; do
             echo "== [$G] round $i: $(env $G python bench.py $QUICK --family-steps ${AB_FAMILY_STEPS:-0} 2>/dev/null | line)"; done; done;;
  esac
done
