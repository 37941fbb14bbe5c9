#!/bin/bash
# Round-2 measurement call: rocprofv3 kernel-trace stats of the three BASELINE configs the bench line carries (C2 EWMA, C4
# DBSCAN, C3 ARIMA), the HBM PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, no tracing flags) for C2 and C4, and the
# SQ counter passes for the ARIMA fit kernel.  usage: tools/gpu_measure_r2.sh <tag>   -> gpurun_out/<tag>/*.csv|json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-other-configs"
kt() {  # name, bench args
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$n -o $n -- $B "$@" > $O/kt_$n.log 2>&1
  python $R/tools/rocpd_summary.py $O/kt_$n/${n}_results.db > $O/${n}_kernel_stats.csv 2>> $O/kt_$n.log
  rm -rf $O/kt_$n
  head -9 $O/${n}_kernel_stats.csv
}
kt ewma_c2 --config c2 --steps 5 --warmup 1
kt dbscan_c4 --config c4 --steps 5 --warmup 1
kt arima_c3 --config c3 --steps 1 --warmup 0
pmc() {  # name, counters, bench args...
  n=$1; c=$2; shift; shift
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$n -o x -- $B "$@" > $O/pmc_$n.log 2>&1
  cp $O/pmc_$n/x_counter_collection.csv $O/pmc_$n.csv 2>/dev/null || cp $(find $O/pmc_$n -name '*counter_collection.csv' | head -1) $O/pmc_$n.csv
  rm -rf $O/pmc_$n
}
pmc c2_fetch FETCH_SIZE --config c2 --steps 2 --warmup 1
pmc c2_write WRITE_SIZE --config c2 --steps 2 --warmup 1
pmc c4_fetch FETCH_SIZE --config c4 --steps 2 --warmup 1
pmc c4_write WRITE_SIZE --config c4 --steps 2 --warmup 1
pmc c3_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES" --config c3 --rows 10000000 --keys 10000 --steps 1 --warmup 0
pmc c3_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" --config c3 --rows 10000000 --keys 10000 --steps 1 --warmup 0
pmc c3_fetch FETCH_SIZE --config c3 --rows 10000000 --keys 10000 --steps 1 --warmup 0
cd $R
python tools/pmc_to_json.py $O/pmc_c2_fetch.csv $O/pmc_c2_write.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --config c2 --steps 2 --warmup 1 (gpurun $1)" > $O/pmc_c2.json
python tools/pmc_to_json.py $O/pmc_c4_fetch.csv $O/pmc_c4_write.csv "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --config c4 --steps 2 --warmup 1 (gpurun $1)" > $O/pmc_c4.json
python - <<PY
import csv, collections, json, glob
out = {}
for f in sorted(glob.glob("$O/pmc_c3_*.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_arima_fit" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({c: sum(v) / len(v) for c, v in d.items()})
json.dump({"kernel": "k_arima_fit_w3", "workload": "bench.py --config c3 --rows 10000000 --keys 10000 --steps 1 (1/10 of C3: 2.42e6 fits)", "counters": out}, open("$O/pmc_c3_arima_fit.json", "w"), indent=1)
print(json.dumps(out))
for n in ("c2", "c4"):
    k = json.load(open("$O/pmc_%s.json" % n))["kernels"]
    print(n, "job bytes fetched %.2f GB written %.2f GB" % (sum(v["fetch_bytes"] for v in k.values()) / 1e9, sum(v["write_bytes"] for v in k.values()) / 1e9), {a: (round(b["fetch_bytes"] / 1e6), round(b["write_bytes"] / 1e6)) for a, b in k.items() if b["fetch_bytes"] + b["write_bytes"] > 2e7})
PY
rm -f $O/pmc_c*_fetch.csv $O/pmc_c*_write.csv $O/pmc_c3_sq*.csv $O/*.log
