#!/bin/bash
# per-kernel times of the attention kernels in the stand-alone probe (rocprofv3 kernel trace): tools/gpu_attn6.sh "<ENV=..> ..." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp ATTNP_REPS=20 ATTNP_BWD=1 TFX_LIB=${TFX_LIB:-$R/transfusion_pytorch_amd/lib/libtfx_hip.so}
cd /tmp
for spec in "$@"; do
  rm -rf /tmp/kt
  env $spec timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- $R/tools/attn_probe run kt ${CASE:-bench} > /tmp/kt.log 2>&1
  echo "== $spec"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'attn' in r['Name']: print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
