#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|GPU vs oracle|5-digit|job: within|rc=" $O/pytest_gpu.log | head -12
bash $R/tools/gpu_quick.sh $1/q "A=1" "B=1"
timeout 400 python bench.py --algo ARIMA --rows 2000000 --keys 2000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_arima_small.json 2> $O/bench_arima_small.err
python - <<PY
import json
d=json.load(open("$O/bench_arima_small.json")); print("arima small ms/step", d["ms_per_step"], d["arima"])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o ewma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$O/prof_kt/ewma_kernel_stats.csv")):
    if float(r["AverageNs"]) > 20000: print(r["Name"][:70].ljust(70), r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
