#!/bin/bash
# usage: gpu_clocks.sh <tag>  -> sample rocm-smi clocks while the C2 bench runs, several times
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^$\|====" | head -20
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 1500 --warmup 2 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err &
  pid=$!
  sleep 3.5
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power" | tr '\n' ';' | cut -c1-600
  echo
  wait $pid
  python - <<PY
import json
d=json.load(open("$O/b$i.json"))
print("kern %.3f ms/step %.3f meta %.3f detect %.3f" % (d["roofline"]["avg_kernel_ms"], d["ms_per_step"], d["pipeline"]["ms_meta"], d["pipeline"]["ms_detect_and_emit"]))
PY
done
