#!/bin/bash
# ingest kernels on the GPU: parity tests of tad_factorize / tad_encode_strings (+ the host paths that use them), the 1e8-row timings, kernel stats.
# usage: tools/gpu_ingest.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
if [ "$2" != notests ]; then ( timeout 900 python -m pytest tests/test_gpu_factorize.py tests/test_rest.py tests/test_clickhouse_http.py tests/test_controller.py tests/test_gpu_job.py -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_ingest.log 2>&1; fi
timeout 600 python tools/factorize_bench.py > $O/factorize_bench.log 2>&1
timeout 600 python tools/strings_bench.py > $O/strings_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o sb -- python $R/tools/strings_bench.py --steps 3 --host-rows 1000000 --arrow-rows 1000000 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/strings_kernel_stats.csv
rm -rf $O/kt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o fb -- python $R/tools/factorize_bench.py --steps 3 --pandas-rows 1000000 > $O/kt2.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/factorize_kernel_stats.csv
rm -rf $O/kt
cd $R
cat $O/pytest_ingest.log 2>/dev/null; cat $O/factorize_bench.log $O/strings_bench.log; head -8 $O/strings_kernel_stats.csv; head -8 $O/factorize_kernel_stats.csv
