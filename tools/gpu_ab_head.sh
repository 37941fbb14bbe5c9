#!/bin/bash
# usage: gpu_ab_head.sh <tag> [reps]  -> all GPU tests, then C2 bench alternating gpurun_in/libtad_head.so (previous build) and the tree's library
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|rc=" $O/pytest_gpu.log | head -8; grep -B5 -A30 "Error\|FAILED" $O/pytest_gpu.log | head -80
run() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/$label.json 2> $O/$label.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$label.json"))
    print("$label", "ms/step %.3f" % d["ms_per_step"], "kern %.3f" % d["roofline"]["avg_kernel_ms"], "stage0 %.3f" % d["pipeline"]["ms_stage0_clear_plus_scatter"], "detect+emit %.3f" % d["pipeline"]["ms_detect_and_emit"], d["result"]["anomalies"])
except Exception as e:
    print("$label FAILED", e); print(open("$O/$label.err").read()[-800:])
PY
}
for rep in $(seq 1 ${2:-3}); do
run head_$rep TAD_LIBRARY_PATH=gpurun_in/libtad_head.so
run new_$rep A=1
done
