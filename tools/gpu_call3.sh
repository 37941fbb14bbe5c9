#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c5
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
for rpt in 8 4; do
TAD_RPT=$rpt timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_ewma_rpt$rpt.json 2> $O/bench_ewma_rpt$rpt.err
python - <<PY
import json
d=json.load(open("$O/bench_ewma_rpt$rpt.json"))
print("rpt$rpt", d["ms_per_step"], d["pipeline"], d["roofline"]["avg_kernel_ms"])
PY
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o ewma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
cut -c1-120 $O/prof_kt/ewma_kernel_stats.csv | head -14
