#!/bin/bash
# what the driver runs at round end: the -m gpu suite, smoke(), the default bench line (wall time recorded)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > $O/pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
S=$(date +%s.%N); timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; E=$(date +%s.%N)
echo "bench.py wall seconds: $(python -c "print($E - $S)")" > $O/bench_wall.log
cat $O/pytest.log $O/smoke.log $O/bench_wall.log
python - <<PY
import json
d = json.loads(open("$O/bench_default_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data")})
print(d["roofline"]["frac"], d["roofline"]["traffic"]["source"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
PY
