#!/bin/bash
# Round-6 ingest call 2: per-connection keys (mode None, 5e7 rows, ~5e7 keys) after the full-size-table fix; factorize tests.   usage: tools/gpu_r6_ingest2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_factorize.py -m gpu -q 2>&1 | tail -5 ) > $O/pytest_factorize.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 50000000 --mode default --connections 8 > $O/ingest_e2e_default_c8.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 4 > $O/ingest_e2e_pod_c4.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 1 --runs 2 > $O/ingest_e2e_pod_c1.log 2>&1
cat $O/pytest_factorize.log $O/ingest_e2e_*.log
