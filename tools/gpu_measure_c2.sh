#!/bin/bash
# C2 refresh: rocprofv3 kernel-trace stats, the two HBM PMC passes (separate runs, no tracing flags) and the bench line.
# usage: tools/gpu_measure_c2.sh <tag>   -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs --config c2"
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o c2 -- $B --steps 5 --warmup 1 > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $O/kt/c2_results.db > $O/ewma_c2_kernel_stats.csv 2>> $O/kt.log
rm -rf $O/kt
head -9 $O/ewma_c2_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o x -- $B --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_$c -name '*counter_collection.csv' | head -1) $O/pmc_$c.csv
  rm -rf $O/pmc_$c
done
cd $R
python tools/pmc_to_json.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --config c2 --steps 2 --warmup 1 (gpurun $1)" > $O/pmc_c2.json
rm -f $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv
timeout 60 $B --steps 20 --warmup 3 > $O/bench_c2_line.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_c2_line.json')); print(d['ms_per_step'], d['roofline']['frac'], d['pipeline'])"
