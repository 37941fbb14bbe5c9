#!/bin/bash
# k_arima_start (block order, prefetching window, occupancy): ARIMA parity tests on the shipped build, same-process A/B of the C3 job over
# the variant builds given, kernel-trace stats of C3 for each.   usage: tools/gpu_r5_arima_start.sh <tag> [variant ...]  (names under theia_amd/lib/variants/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_gpu_arima.py tests/test_gpu_fullsize.py -m gpu -q -k "arima or ARIMA or c2" 2>&1 | tail -5 ) > $O/pytest_arima.log 2>&1
V="shipped="
for v in "$@"; do V="$V;$v=lib:$R/theia_amd/lib/variants/libtad_$v.so"; done
timeout 300 python tools/ab_plans.py --config c3 --variants "$V" --rounds 3 --steps 1 > $O/ab_c3.log 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs"
kt() {  # name, command...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n
}
kt arima_c3_shipped $B --config c3 --steps 1 --warmup 0
for v in "$@"; do TAD_LIBRARY_PATH=$R/theia_amd/lib/variants/libtad_$v.so kt arima_c3_$v $B --config c3 --steps 1 --warmup 0; done
grep -H "k_arima_start\|k_arima_prep\|k_arima_fit" $O/*_kernel_stats.csv | cut -c1-260; tail -8 $O/ab_c3.log; cat $O/pytest_arima.log
