#!/bin/bash
# usage: gpu_tests_and_quick.sh <tag> "<ENV..>" ...   -> GPU parity tests, then bench.py (C2) once per env set
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|Error|rc=" $O/pytest_gpu.log | head -8; grep -B5 -A25 "Error\|FAILED" $O/pytest_gpu.log | head -60
tag=$1; shift
bash $R/tools/gpu_quick.sh $tag/q "$@"
