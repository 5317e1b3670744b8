#!/bin/bash
# per-kernel times of the attention backward with its tile loops cut to 0 / 2 tiles (timing builds): the fixed cost of a block
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp ATTNP_REPS=20 ATTNP_BWD=1
cd /tmp
for lib in hip HOTLOAD; do
  rm -rf /tmp/kt
  TFX_LIB=$R/transfusion_pytorch_amd/lib/libtfx_$lib.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o p -- $R/tools/attn_probe run kt bench > /tmp/kt.log 2>&1
  echo "== lib $lib"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
