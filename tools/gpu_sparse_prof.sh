#!/bin/bash
# kernel stats (rocprofv3 --kernel-trace --stats) and HBM counters (separate --pmc passes) of the sparse Stage 0 at scale.
# usage: tools/gpu_sparse_prof.sh <tag> [rows]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
ROWS=${2:-100000000}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
kt() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n; }
pmc() { n=$1; c=$2; shift; shift
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o x -- "$@" > $O/pmc_$n.log 2>&1
  cp $(find $O/pmc_$n -name '*counter_collection.csv' | head -1) $O/pmc_$n.csv 2>/dev/null
  rm -rf $O/pmc_$n; }
kt sparse_scale python $R/tools/sparse_bench.py --steps 3 --rows $ROWS --only-scale --algos EWMA
kt sparse_small python $R/tools/sparse_bench.py --steps 3
pmc sparse_scale_fetch FETCH_SIZE python $R/tools/sparse_bench.py --steps 1 --rows $ROWS --only-scale --algos EWMA
pmc sparse_scale_write WRITE_SIZE python $R/tools/sparse_bench.py --steps 1 --rows $ROWS --only-scale --algos EWMA
cd $R
python tools/pmc_to_json.py $O/pmc_sparse_scale_fetch.csv $O/pmc_sparse_scale_write.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two passes, tools/sparse_bench.py --rows $ROWS --only-scale (gpurun $1)" > $O/pmc_sparse_scale.json
rm -f $O/pmc_sparse_scale_fetch.csv $O/pmc_sparse_scale_write.csv $O/kt_*.log $O/pmc_*.log
head -25 $O/sparse_scale_kernel_stats.csv | cut -c1-150
