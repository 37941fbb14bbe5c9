#!/bin/bash
# Round-6 fourth call: ARIMA fit with mid-fit suspension: ARIMA / concurrency parity tests, the default bench line (C3 timing, short job beside long job).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests/test_gpu_arima.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "arima or yields or concurrent or job_contexts or c3" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
cat $O/pytest.log; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default_line.json").read().strip().splitlines()[-1])
print("C2", d["ms_per_step"], d["roofline"]["frac"], "cold", d.get("cold", {}).get("ms_first_step"), "same", d.get("same_columns", {}).get("ms_per_step"))
c = d.get("concurrency", {})
print({k: (v["value"], v["vs_one_in_flight"]) for k, v in c.get("levels", {}).items()})
print(json.dumps(c.get("short_job_beside_long_job"), indent=1))
for k, v in d.get("other_configs", {}).items():
    print(k, v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("error"))
PY
