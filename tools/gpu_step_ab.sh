#!/bin/bash
# same-box A/B of the training step over environment settings: tools/gpu_step_ab.sh "<tag>:<ENV=..> <ENV=..>" ...   (two rounds; family table of each)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
O=$R/gpurun_out/step_ab.txt
: > $O
for round in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --steps ${STEPS:-12} --warmup 4 --no-cpu-baseline --no-sample --no-other-configs --no-parity --ragged-steps 0 > /tmp/ab_$tag.json 2>/tmp/ab_$tag.err || tail -3 /tmp/ab_$tag.err >> $O
    python - "$tag" /tmp/ab_$tag.json >> $O <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    fam = d.get('roofline_by_family', [])
    parts = ' '.join(f"{(v['kernel'][:5] if v['bound'] == 'hbm' else v['kernel'][4:])}={v['ms_per_step']:.3f}" for v in fam)
    print(f"{tag:14s} ms/step {d['ms_per_step']:.3f}  {parts}")
except Exception as e:
    print(tag, 'FAILED', e)
PY
  done
done
cat $O
