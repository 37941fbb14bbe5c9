#!/bin/bash
# C3 (ARIMA) timing of the shipped library + the time split of k_arima_fit from a profiling build (-DTAD_ARIMA_PROF, tools/build_variants.py)
cd /root/repo
O=gpurun_out/${1:-arima}; mkdir -p $O
V=$PWD/theia_amd/lib/variants
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('$1', round(d['ms_per_step'],3), 'ms; detect', round(p['ms_detect_and_emit'],3), 'anomalies', d['result']['anomalies'], 'nan fits', d['arima']['nan_fits'], 'frac', round(d['arima']['frac'],4), 'frac60', round(d['arima']['frac_60flop_equivalent'],4))"; }
{
( timeout 300 python -m pytest tests/test_gpu_arima.py -m gpu -x -q 2>&1 | tail -2 )
for r in 1 2; do timeout 120 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "C3"; done
for v in $V/libtad_prof*.so; do echo $v; TAD_LIBRARY_PATH=$v timeout 120 python bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "arima prof"; done
for v in $(ls $V/libtad_*.so | grep -v prof); do for r in 1 2; do TAD_LIBRARY_PATH=$v timeout 120 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | line "C3 $(basename $v)"; done; done
} > $O/c3.log 2>&1
cat $O/c3.log
