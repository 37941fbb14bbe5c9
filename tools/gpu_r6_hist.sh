#!/bin/bash
# Round-6: the factorisation's key-bin histogram in front of the C4 job (item 7): parity tests, same-process A/B, kernel stats and FETCH / WRITE
# of the job with the histogram; pass B with the key-range check against the library before it (C2 / C4 A/B of two builds).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_gpu_factorize.py tests/test_gpu_parity.py -m gpu -q -k "histogram or factorisation or key_out_of_range or sparse_table" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
timeout 600 python tools/hist_byproduct_bench.py > $O/hist_byproduct_c4.log 2>&1
timeout 600 python tools/hist_byproduct_bench.py --algo EWMA --agg svc --keys 100000 --buckets 250 > $O/hist_byproduct_c2.log 2>&1
cd /tmp; export TMPDIR=/tmp
for w in with without; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o x -- python $R/tools/hist_byproduct_bench.py --only $w --jobs 6 > /dev/null 2>&1
  f=$(find $O/kt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c4_ingest_${w}_kernel_stats.csv; rm -rf $O/kt_$w
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -o x -- python $R/tools/hist_byproduct_bench.py --only $w --jobs 2 > /dev/null 2>&1
    cp $(find $O/pmc_${w}_$c -name '*counter_collection.csv' | head -1) $O/pmc_${w}_$c.csv 2>/dev/null; rm -rf $O/pmc_${w}_$c
  done
  cd $R; python tools/pmc_to_json.py $O/pmc_${w}_FETCH_SIZE.csv $O/pmc_${w}_WRITE_SIZE.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes, tools/hist_byproduct_bench.py --only $w (gpurun $1)" > $O/pmc_c4_ingest_$w.json; cd /tmp
  rm -f $O/pmc_${w}_*.csv
done
cd $R
V=theia_amd/lib/variants
[ -f $V/libtad_prev.so ] && timeout 600 python tools/ab_plans.py --config c2 --variants "now=;prev=lib:$V/libtad_prev.so" > $O/ab_c2_keyrange_check.log 2>&1
[ -f $V/libtad_prev.so ] && timeout 600 python tools/ab_plans.py --config c4 --variants "now=;prev=lib:$V/libtad_prev.so" > $O/ab_c4_keyrange_check.log 2>&1
cat $O/pytest.log $O/hist_byproduct_c4.log $O/hist_byproduct_c2.log $O/ab_c*_keyrange_check.log 2>/dev/null
python - <<PY
import json
for w in ("with", "without"):
    k = json.load(open("$O/pmc_c4_ingest_%s.json" % w))["kernels"]
    print(w, {a: (round(b["fetch_bytes"] / 1e6), round(b["write_bytes"] / 1e6)) for a, b in k.items() if a.startswith("k_meta") or a.startswith("k_partition") or a.startswith("k_tile")})
PY
