#!/bin/bash
# DBSCAN parity tests, then the C4 bench line per env set.  usage: tools/gpu_c4_emit.sh "ENV=.." ...
cd /root/repo
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_gpu_job.py tests/test_gpu_random.py -m gpu -x -q -k "DBSCAN or dbscan or concurrent or aggregate or empty" 2>&1 | tail -3
run() { env $1 timeout 60 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C4 [$1]', round(d['ms_per_step'],3), 'ms; stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'detect', round(p['ms_detect_and_emit'],3), d['result']['anomalies'])"; }
for e in "$@"; do run "$e"; done
