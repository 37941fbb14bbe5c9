#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/a$i -o arima -- python $R/bench.py --algo ARIMA --rows 2000000 --keys 2000 --steps 1 --warmup 0 --no-cpu-baseline > $O/a$i.log 2>&1
done
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/a*/arima_counter_collection.csv")):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in d:
        if "arima" in k:
            print(k, {c: "%.3g" % (sum(v)/len(v)) for c, v in d[k].items()})
PY
