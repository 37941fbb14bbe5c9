#!/bin/bash
# tad_encode_strings + the read-back path on the GPU: their parity tests, then the 1e8-row timing and kernel stats.  usage: tools/gpu_strings.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_factorize.py tests/test_rest.py tests/test_clickhouse_http.py tests/test_controller.py tests/test_capi_abi.py -m gpu -q -x 2>&1 | tail -12 ) > $O/pytest_strings.log 2>&1
timeout 600 python tools/strings_bench.py > $O/strings_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o sb -- python $R/tools/strings_bench.py --rows 50000000 --steps 3 --host-rows 1000000 --arrow-rows 1000000 > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/strings_kernel_stats.csv
rm -rf $O/kt
cd $R
cat $O/pytest_strings.log $O/strings_bench.log; head -12 $O/strings_kernel_stats.csv
