#!/bin/bash
# Round-6 ingest call: the device ingest end to end at 1e8 rows (pod mode), 8 and 16 connections, page-locked and pageable receive buffers;
# the new ingest tests on the GPU.    usage: tools/gpu_r6_ingest.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
nproc > $O/host.txt; free -g >> $O/host.txt
( timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py -m gpu -q -k "ingest or yields or concurrent or job_contexts" 2>&1 | tail -5 ) > $O/pytest_ingest.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 8 --compare-host 2000000 > $O/ingest_e2e_pod_c8.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 16 > $O/ingest_e2e_pod_c16.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 8 --no-pinned > $O/ingest_e2e_pod_c8_pageable.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 100000000 --mode svc --connections 8 > $O/ingest_e2e_svc_c8.log 2>&1
timeout 900 python tools/ingest_e2e.py --rows 50000000 --mode default --connections 8 > $O/ingest_e2e_default_c8.log 2>&1
cat $O/host.txt $O/pytest_ingest.log $O/ingest_e2e_*.log
