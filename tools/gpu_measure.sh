#!/bin/bash
# The round's measurement call: GPU parity tests, bench lines (C2 EWMA with cpu_baseline, C4 DBSCAN, C3 ARIMA),
# rocprofv3 kernel-trace stats and the two HBM PMC passes of the C2 command.  usage: gpu_measure.sh <tag> [full]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log | head -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
timeout 300 python bench.py > $O/bench_ewma_c2.json 2> $O/bench_ewma_c2.err; head -c 600 $O/bench_ewma_c2.json; echo
timeout 300 python bench.py --algo DBSCAN --keys 1000000 --buckets 100 --agg "" --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_dbscan_c4.json 2> $O/bench_dbscan_c4.err
if [ "$2" = "full" ]; then
timeout 600 python bench.py --algo ARIMA --steps 1 --warmup 1 --cpu-rows 40000 > $O/bench_arima_c3.json 2> $O/bench_arima_c3.err
fi
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o ewma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o ewma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o ewma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1
cd $R
python tools/pmc_to_json.py $O/prof_fetch/ewma_counter_collection.csv $O/prof_write/ewma_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps 2 --warmup 1 (gpurun $1)" > $O/pmc.json
python - <<PY
import csv, json
for r in csv.DictReader(open("$O/prof_kt/ewma_kernel_stats.csv")):
    if float(r["AverageNs"]) > 3000: print(r["Name"][:70].ljust(70), r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
print(json.dumps(json.load(open("$O/pmc.json"))["kernels"]))
for f in ("bench_dbscan_c4", "bench_arima_c3"):
    try:
        d = json.load(open("$O/%s.json" % f)); print(f, d["value"], d["ms_per_step"], d["pipeline"], d.get("arima"))
    except Exception as e: print(f, "n/a", e)
PY
