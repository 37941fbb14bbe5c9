#!/bin/bash
cd /root/repo
O=gpurun_out/r3c4; mkdir -p $O
V=$PWD/theia_amd/lib/variants
bash tools/gpu_arima_prof.sh r3c4 > /dev/null 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('$1', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'partB', round(d['roofline'].get('avg_kernel_ms',0),3), 'detect', round(p['ms_detect_and_emit'],3), 'anomalies', d['result']['anomalies'], 'path', d['roofline']['kernel'][:16])"; }
{
for r in 1 2; do
  timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 default"
  for v in kp10 kp11; do
    TAD_LIBRARY_PATH=$V/libtad_$v.so timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 $v"
    TAD_LIBRARY_PATH=$V/libtad_$v.so timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline --plan partition_pass=wc 2>/dev/null | line "C4 $v wc-forced"
  done
done
} > $O/c4.log 2>&1
cat $O/c3.log $O/c4.log
