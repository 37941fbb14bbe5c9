#!/bin/bash
# FETCH_SIZE calibration by load width (tools/probes/fetch_calibrate.hip): plain run for the rates, then one --pmc pass per counter.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$R/tools/probes/fetch_calibrate > $O/fetch_calibrate.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o x -- $R/tools/probes/fetch_calibrate > /dev/null 2>&1
  f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" $c >> $O/fetch_calibrate.log <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_" in k:
        print("%s %-40s launches %d, last launch %.1f KiB = %.4f of 1 GiB (x2: %.4f)" % (sys.argv[2], k[-40:], len(v), v[-1], v[-1] * 1024 / 2**30, 2 * v[-1] * 1024 / 2**30))
PY
  rm -rf $O/pmc_$c
done
cat $O/fetch_calibrate.log
