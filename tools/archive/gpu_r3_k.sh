#!/bin/bash
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -3; grep -n "^FAILED\|^E  " gpurun_out/${TAG}_pytest.log | head -20
bash tools/gpu_evidence_r3.sh $TAG 2>&1 | tail -12 | cut -c1-400
