#!/bin/bash
# Round-6 second call: the library after the removals (one-synchronisation form, placement search) with job contexts (ABI 12):
# the full -m gpu suite, the default bench line (fresh-columns loop, same_columns, cold, concurrency) and the cold probe.   usage: tools/gpu_r6_second.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 300 python tools/cold_probe.py --config c2 > $O/cold_c2.log 2>&1
timeout 300 python tools/cold_probe.py --config c4 > $O/cold_c4.log 2>&1
cat $O/pytest.log $O/cold_c2.log $O/cold_c4.log; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default_line.json").read().strip().splitlines()[-1])
print("C2", d["ms_per_step"], d["roofline"]["frac"], "cold", d.get("cold", {}).get("ms_first_step"), "same", d.get("same_columns"))
print(json.dumps(d.get("concurrency"), indent=1))
for k, v in d.get("other_configs", {}).items():
    print(k, v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("error"))
PY
