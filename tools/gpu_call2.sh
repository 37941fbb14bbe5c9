#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c2
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --algo ARIMA --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_arima_c3.json 2> $O/bench_arima_c3.err
cat $O/bench_arima_c3.json | head -c 2000
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_arima -o arima -- python $R/bench.py --algo ARIMA --rows 10000000 --keys 10000 --steps 1 --warmup 0 --no-cpu-baseline > $O/prof_arima.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dbscan -o dbscan -- python $R/bench.py --algo DBSCAN --keys 1000000 --buckets 100 --agg "" --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_dbscan.log 2>&1
cd $R
head -8 $O/prof_arima/arima_kernel_stats.csv | cut -c1-150
head -8 $O/prof_dbscan/dbscan_kernel_stats.csv | cut -c1-150
