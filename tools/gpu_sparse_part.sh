#!/bin/bash
# Sparse Stage 0, partition + LDS sort against the LSD sort: parity tests of both, 1e8-row timing of both, kernel stats of the partition form.
# usage: tools/gpu_sparse_part.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
if [ "$2" != notests ]; then ( timeout 900 python -m pytest tests/test_gpu_sparse_partition.py tests/test_gpu_sparse.py -m gpu -q -x -s 2>&1 | tail -15 ) > $O/pytest_sparse.log 2>&1; fi
timeout 600 python tools/sparse_bench.py --steps 5 --rows 100000000 --only-scale --sorts lsd,partition > $O/sparse_scale_ab.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o sp -- python $R/tools/sparse_bench.py --steps 3 --rows 100000000 --only-scale --algos EWMA --sorts partition > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/sparse_part_kernel_stats.csv
rm -rf $O/kt
cd $R
cat $O/pytest_sparse.log 2>/dev/null | tail -8; cat $O/sparse_scale_ab.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/sparse_part_kernel_stats.csv")))
for r in rows[:22]:
    print("  %-64s calls %4s avg %9.1f us  %5s%%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
