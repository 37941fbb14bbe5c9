#!/bin/bash
# same-box A/B of the shipped library against every build under theia_amd/lib/variants (tools/build_variants.py) on C2 and C4
cd /root/repo
O=gpurun_out/${1:-abv}; mkdir -p $O
V=$PWD/theia_amd/lib/variants
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('$1', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'partB', round(d['roofline'].get('avg_kernel_ms',0),3), 'detect', round(p['ms_detect_and_emit'],3), 'anomalies', d['result']['anomalies'])"; }
{
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 )
for r in 1 2; do
  timeout 120 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C2 shipped"
  for v in $(ls $V | grep -v prof | sed 's/libtad_//; s/.so//'); do
    TAD_LIBRARY_PATH=$V/libtad_$v.so timeout 120 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | line "C2 $v"
  done
  timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 shipped"
  for v in $(ls $V | grep -v prof | sed 's/libtad_//; s/.so//'); do
    TAD_LIBRARY_PATH=$V/libtad_$v.so timeout 120 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | line "C4 $v"
  done
done
} > $O/ab.log 2>&1
cat $O/ab.log
