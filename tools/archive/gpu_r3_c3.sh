#!/bin/bash
# round-3: config 3 launches by grid for the slow outliers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/p3 && TFX_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p3 -o p -- python $R/tools/bench_configs.py 3 > /tmp/p3.log 2>&1)
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/p3/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'gemm_nt_glds' in n or 'gemm_nt_pp_kernelILi1' in n or 'gemm_nt_pp_kernel<1>' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        prev = rows[i - 1]['Kernel_Name'][:60]; nxt = rows[i + 1]['Kernel_Name'][:60] if i + 1 < len(rows) else ''
        agg[(n[:50], r['Grid_Size_X'], r['Workgroup_Size_X'])].append((d, prev, nxt))
for k, v in agg.items():
    print(k, len(v), 'avg us', round(sum(x[0] for x in v) / len(v), 1), '| after:', v[-1][1], '| before:', v[-1][2])
PY
