#!/bin/bash
# usage: gpu_ab.sh <tag> "<ENV..>" ...  -> Stage-0 parity tests, then bench.py (C2, 10 steps) once per env set
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_job.py tests/test_gpu_stream.py -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|rc=" $O/pytest_gpu.log | head -8; grep -B5 -A30 "Error\|FAILED" $O/pytest_gpu.log | head -80
tag=$1; shift
bash $R/tools/gpu_quick.sh $tag/q "$@"
