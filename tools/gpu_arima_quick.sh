#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_arima.py tests/test_gpu_job.py -m gpu -x -q -s ) > $O/pytest_arima.log 2>&1; echo "pytest rc=$?" >> $O/pytest_arima.log
grep -E "passed|failed|within|flips|matches|rc=" $O/pytest_arima.log | head -20
timeout 400 python bench.py --algo ARIMA --rows 2000000 --keys 2000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_arima_small.json 2> $O/bench_arima_small.err
python - <<PY
import json
d=json.load(open("$O/bench_arima_small.json")); print("arima small ms/step", d["ms_per_step"], d["arima"])
PY
