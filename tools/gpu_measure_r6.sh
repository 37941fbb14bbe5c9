#!/bin/bash
# Round-6 measurement call (the committed library): the full -m gpu suite, the default bench line (C2 headline over 4 tables + same_columns + cold +
# concurrency + C4 + C3 + C5 at N = 1 with the CPU baselines), the host-input line, the cold probe, rocprofv3 kernel-trace stats of C2 / C4 / C3 / C5 at
# N = 1 and of the sparse path, HBM PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no tracing flags) for C2, C4, C5 and the sparse scale run,
# tad_factorize / tad_encode_strings at 1e8 rows, the row orders / key lifetimes / time windows of tools/order_bench.py, the device ingest end to end at 1e8 rows.   usage: tools/gpu_measure_r6.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
if [ "$2" != notests ]; then ( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; fi
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 120 python bench.py --host-input --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_host_input.json 2>/dev/null
timeout 300 python tools/cold_probe.py --config c2 > $O/cold_c2.log 2>&1
timeout 300 python tools/cold_probe.py --config c4 > $O/cold_c4.log 2>&1
timeout 400 python tools/sparse_bench.py --steps 5 --rows 100000000 --sorts lsd,auto > $O/sparse_bench.log 2>&1
timeout 600 python tools/factorize_bench.py > $O/factorize_bench.log 2>&1
( timeout 300 python tools/order_bench.py --config c2 2>&1 | grep -v amdgpu.ids ) > $O/order_c2.log
( timeout 300 python tools/order_bench.py --config c4 2>&1 | grep -v amdgpu.ids ) > $O/order_c4.log
( timeout 300 python tools/hist_byproduct_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/hist_byproduct_c4.log
( timeout 300 python tools/skew_check.py 100000000 0.0 0.1 0.5 2>&1 | grep -v amdgpu.ids ) > $O/hot_key.log
timeout 600 python tools/strings_bench.py > $O/strings_bench.log 2>&1
timeout 600 python tools/ingest_e2e.py --rows 100000000 --mode pod --connections 8 --compare-host 2000000 > $O/ingest_e2e_pod.log 2>&1
timeout 600 python tools/ingest_e2e.py --rows 50000000 --mode default --connections 8 > $O/ingest_e2e_default.log 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs --tables 1"
kt() {  # name, command...
  n=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n
}
kt ewma_c2 $B --config c2 --steps 5 --warmup 1
kt dbscan_c4 $B --config c4 --steps 5 --warmup 1
kt arima_c3 $B --config c3 --steps 1 --warmup 0
kt ewma_c5_n1 $B --config c5 --algo EWMA --steps 3 --warmup 1
kt sparse python $R/tools/sparse_bench.py --steps 3
kt sparse_scale python $R/tools/sparse_bench.py --steps 3 --rows 100000000 --only-scale --algos EWMA
pmc() {  # name, counters, command...
  n=$1; c=$2; shift; shift
  timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o x -- "$@" > $O/pmc_$n.log 2>&1
  cp $(find $O/pmc_$n -name '*counter_collection.csv' | head -1) $O/pmc_$n.csv 2>/dev/null
  rm -rf $O/pmc_$n
}
pmc c2_fetch FETCH_SIZE $B --config c2 --steps 2 --warmup 1
pmc c2_write WRITE_SIZE $B --config c2 --steps 2 --warmup 1
pmc c4_fetch FETCH_SIZE $B --config c4 --steps 2 --warmup 1
pmc c4_write WRITE_SIZE $B --config c4 --steps 2 --warmup 1
pmc c5_fetch FETCH_SIZE $B --config c5 --algo EWMA --steps 1 --warmup 1
pmc c5_write WRITE_SIZE $B --config c5 --algo EWMA --steps 1 --warmup 1
pmc sparse_fetch FETCH_SIZE python $R/tools/sparse_bench.py --steps 1 --rows 100000000 --only-scale --algos EWMA
pmc sparse_write WRITE_SIZE python $R/tools/sparse_bench.py --steps 1 --rows 100000000 --only-scale --algos EWMA
cd $R
for n in c2 c4 c5 sparse; do
  python tools/pmc_to_json.py $O/pmc_${n}_fetch.csv $O/pmc_${n}_write.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two passes, $n (gpurun $1)" > $O/pmc_$n.json
done
python - <<PY
import json
for n in ("c2", "c4", "c5", "sparse"):
    k = json.load(open("$O/pmc_%s.json" % n))["kernels"]
    job = {a: b for a, b in k.items() if a not in ("k_synth",)}     # (the table generator is not the job)
    print(n, "bytes fetched %.2f GB written %.2f GB per job (one launch of every kernel of the job)" % (sum(v["fetch_bytes"] for v in job.values()) / 1e9, sum(v["write_bytes"] for v in job.values()) / 1e9), {a: (round(b["fetch_bytes"] / 1e6), round(b["write_bytes"] / 1e6)) for a, b in k.items() if b["fetch_bytes"] + b["write_bytes"] > 2e7})
PY
rm -f $O/pmc_c*_fetch.csv $O/pmc_c*_write.csv $O/pmc_sparse_*.csv $O/kt_*.log $O/pmc_*.log
cat $O/pytest.log $O/cold_c2.log $O/cold_c4.log $O/order_c2.log $O/order_c4.log $O/hist_byproduct_c4.log $O/hot_key.log $O/sparse_bench.log $O/factorize_bench.log $O/strings_bench.log $O/ingest_e2e_pod.log $O/ingest_e2e_default.log 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['roofline']['frac'], d['pipeline']['hbm_frac_whole_job']); [print(k, v.get('ms_per_step'), v.get('roofline',{}).get('frac')) for k,v in d.get('other_configs',{}).items()]; print(json.loads(open('$O/bench_host_input.json').read().strip().splitlines()[-1])['ms_per_step'])"
