#!/bin/bash
# sample socket power / shader clock with rocm-smi while the training-step bench runs: is the step power-limited?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rocm-smi --showmaxpower --showpowerprofile 2>/dev/null | grep -iv "^$\|====" | head -12
(python bench.py --steps 500 --warmup 5 --no-cpu-baseline --ragged-steps 0 --no-sample --no-other-configs --no-parity > /tmp/pw.json 2>/dev/null) &
BP=$!
sleep 9
for i in $(seq 16); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "Package Power\|sclk\|mclk\|junction\|fclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.4
done
wait $BP
python -c "import json; d=json.load(open('/tmp/pw.json')); print(d['ms_per_step'])"
