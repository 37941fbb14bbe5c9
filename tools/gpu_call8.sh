#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log | grep -E "passed|failed|Error|assert" | head -8
bash $R/tools/gpu_quick.sh $1/q "TAD_RPT=12" "TAD_RPT=10" "TAD_RPT=8"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o ewma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$O/prof_kt/ewma_kernel_stats.csv")):
    if float(r["AverageNs"]) > 3000: print(r["Name"][:70].ljust(70), r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
