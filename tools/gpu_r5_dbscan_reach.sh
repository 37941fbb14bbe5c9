#!/bin/bash
# k_dbscan_list_wave walking the smaller of (core, non-core) points for `reach`: DBSCAN parity tests on the shipped build, same-process A/B of
# the C4 job against the build before the change, kernel-trace stats of C4.   usage: tools/gpu_r5_dbscan_reach.sh <tag> [variant ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -q -k "dbscan or DBSCAN or c4" 2>&1 | tail -5 ) > $O/pytest_dbscan.log 2>&1
V="shipped="
for v in "$@"; do V="$V;$v=lib:$R/theia_amd/lib/variants/libtad_$v.so"; done
timeout 300 python tools/ab_plans.py --config c4 --variants "$V" --rounds 6 --steps 20 > $O/ab_c4.log 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs"
kt() {  # name, command...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n
}
kt dbscan_c4_shipped $B --config c4 --steps 5 --warmup 1
for v in "$@"; do TAD_LIBRARY_PATH=$R/theia_amd/lib/variants/libtad_$v.so kt dbscan_c4_$v $B --config c4 --steps 5 --warmup 1; done
grep -H "k_dbscan" $O/*_kernel_stats.csv | cut -c1-220; tail -8 $O/ab_c4.log; cat $O/pytest_dbscan.log
