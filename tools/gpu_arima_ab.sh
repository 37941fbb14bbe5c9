#!/bin/bash
# ARIMA (C3): parity tests, then the shipped library against every build under theia_amd/lib/variants, same box alternating
cd /root/repo
O=gpurun_out/${1:-arab}; mkdir -p $O
V=$PWD/theia_amd/lib/variants
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],1), 'ms; kernel', round(r.get('avg_kernel_ms',0),1), 'frac', round(r['frac'],3), 'nan_fits', d.get('arima',{}).get('nan_fits'), 'anomalies', d['result']['anomalies'])"; }
{
( timeout 900 python -m pytest tests/test_gpu_arima.py tests/test_gpu_job.py tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | cut -c1-300 | tail -8 )
for r in 1 2; do
  timeout 300 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C3 shipped"
  for v in $(ls $V | grep -v prof | sed 's/libtad_//; s/.so//'); do
    TAD_LIBRARY_PATH=$V/libtad_$v.so timeout 300 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C3 $v"
  done
done
} > $O/ab.log 2>&1
cat $O/ab.log
