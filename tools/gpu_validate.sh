#!/bin/bash
# usage: gpu_validate.sh <tag>  -> GPU parity tests, 2-rank gloo bench check, default bench.py (with cpu_baseline)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|rc=" $O/pytest_gpu.log | head -8; grep -B5 -A25 "Error\|FAILED" $O/pytest_gpu.log | head -60
bash $R/tools/gpu_multirank_check.sh $1/mr
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<PY
import json
for line in open("$O/bench_default.json"):
    if line.startswith("{"):
        d = json.loads(line)
        print("value %.3e ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
        print(json.dumps(d["cpu_baseline"]))
PY
