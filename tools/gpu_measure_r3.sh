#!/bin/bash
# Round-3 measurement call (the committed library): the full -m gpu suite, the default bench line (C2 headline + C4 + C3 with the
# CPU baselines), rocprofv3 kernel-trace stats of C2 / C4 / C3 and of the sparse path, HBM PMC passes (FETCH_SIZE, WRITE_SIZE:
# separate runs, no tracing flags) for C2, C4 and the sparse path.   usage: tools/gpu_measure_r3.sh <tag> [notests]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
if [ "$2" != notests ]; then ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1; fi
timeout 600 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 120 python bench.py --host-input --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_host_input.json 2>/dev/null
timeout 120 python tools/sparse_bench.py 5 > $O/sparse_bench.log 2>&1
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs"
kt() {  # name, command...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o $n -- "$@" > $O/kt_$n.log 2>&1
  f=$(find $O/kt_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv
  rm -rf $O/kt_$n
}
kt ewma_c2 $B --config c2 --steps 5 --warmup 1
kt dbscan_c4 $B --config c4 --steps 5 --warmup 1
kt arima_c3 $B --config c3 --steps 1 --warmup 0
kt sparse python $R/tools/sparse_bench.py 3
pmc() {  # name, counters, command...
  n=$1; c=$2; shift; shift
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o x -- "$@" > $O/pmc_$n.log 2>&1
  cp $(find $O/pmc_$n -name '*counter_collection.csv' | head -1) $O/pmc_$n.csv 2>/dev/null
  rm -rf $O/pmc_$n
}
pmc c2_fetch FETCH_SIZE $B --config c2 --steps 2 --warmup 1
pmc c2_write WRITE_SIZE $B --config c2 --steps 2 --warmup 1
pmc c4_fetch FETCH_SIZE $B --config c4 --steps 2 --warmup 1
pmc c4_write WRITE_SIZE $B --config c4 --steps 2 --warmup 1
pmc sparse_fetch FETCH_SIZE python $R/tools/sparse_bench.py 2
pmc sparse_write WRITE_SIZE python $R/tools/sparse_bench.py 2
pmc c3_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES" $B --config c3 --rows 10000000 --keys 10000 --steps 1 --warmup 0
pmc c3_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" $B --config c3 --rows 10000000 --keys 10000 --steps 1 --warmup 0
cd $R
for n in c2 c4 sparse; do
  python tools/pmc_to_json.py $O/pmc_${n}_fetch.csv $O/pmc_${n}_write.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two passes, $n (gpurun $1)" > $O/pmc_$n.json
done
python - <<PY
import csv, collections, json, glob
out = {}
for f in sorted(glob.glob("$O/pmc_c3_*.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_arima_fit" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({c: sum(v) / len(v) for c, v in d.items()})
json.dump({"kernel": "k_arima_fit", "workload": "bench.py --config c3 --rows 10000000 --keys 10000 --steps 1 (1/10 of C3: 2.42e6 fits)", "counters": out}, open("$O/pmc_c3_arima_fit.json", "w"), indent=1)
for n in ("c2", "c4", "sparse"):
    k = json.load(open("$O/pmc_%s.json" % n))["kernels"]
    print(n, "bytes fetched %.2f GB written %.2f GB per step-set" % (sum(v["fetch_bytes"] for v in k.values()) / 1e9, sum(v["write_bytes"] for v in k.values()) / 1e9), {a: (round(b["fetch_bytes"] / 1e6), round(b["write_bytes"] / 1e6)) for a, b in k.items() if b["fetch_bytes"] + b["write_bytes"] > 2e7})
PY
rm -f $O/pmc_c*_fetch.csv $O/pmc_c*_write.csv $O/pmc_sparse_*.csv $O/pmc_c3_sq*.csv $O/kt_*.log $O/pmc_*.log
cat $O/pytest.log $O/sparse_bench.log 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['roofline']['frac'], d['pipeline']['hbm_frac_whole_job']); [print(k, v['ms_per_step'], v['roofline']['frac']) for k,v in d.get('other_configs',{}).items()]; print(json.loads(open('$O/bench_host_input.json').read().strip().splitlines()[-1])['ms_per_step'])"
