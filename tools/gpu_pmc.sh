#!/bin/bash
# SQ counter passes for the EWMA C2 bench (one --pmc set per run; no tracing flags).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o ewma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
done
cd $R
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$O/p*/ewma_counter_collection.csv")):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in d:
        if "k_partition" in k or "k_tile" in k or "k_meta_hist" in k or "k_emit" in k:
            print(k, {c: "%.3g" % (sum(v)/len(v)) for c, v in d[k].items()})
PY
