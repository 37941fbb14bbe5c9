#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|rc=" $O/pytest_gpu.log | head -8; grep -B5 -A30 "Error\|FAILED" $O/pytest_gpu.log | head -80
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/$label.json 2> $O/$label.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$label.json"))
    print("$label", "ms/step %.3f" % d["ms_per_step"], "kern %.3f" % d["roofline"]["avg_kernel_ms"], "stage0 %.3f" % d["pipeline"]["ms_stage0_clear_plus_scatter"], "detect+emit %.3f" % d["pipeline"]["ms_detect_and_emit"], d["roofline"]["kernel"][:16], d["result"]["anomalies"])
except Exception as e:
    print("$label FAILED", e); print(open("$O/$label.err").read()[-800:])
PY
}
for rep in 1 2 3; do
run head_$rep TAD_LIBRARY_PATH=gpurun_in/libtad_head.so --
run new_$rep A=1 --
run new_kp64_$rep TAD_WIDE_KP=0 --
done
