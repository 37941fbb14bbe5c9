#!/bin/bash
# One gpurun call: parity tests, bench lines, rocprofv3 kernel trace + PMC passes. Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench_ewma.json 2> $O/bench_ewma.err
timeout 300 python bench.py --algo DBSCAN --keys 1000000 --buckets 100 --agg "" --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_dbscan.json 2> $O/bench_dbscan.err
timeout 400 python bench.py --algo ARIMA --rows 2000000 --keys 2000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_arima_small.json 2> $O/bench_arima_small.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o ewma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o ewma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o ewma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1
cd $R
ls -R $O | head -50
tail -3 $O/pytest_gpu.log
cat $O/bench_ewma.json | head -c 1500
