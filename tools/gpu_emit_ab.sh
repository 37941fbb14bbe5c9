#!/bin/bash
# staged-emit check on the GPU box: the emit parity tests, then the C2 bench line with k_emit_staged on / off, alternating.
cd /root/repo
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "emit_staged or synthetic_tables or irregular or empty" 2>&1 | tail -3
run() { env $1 timeout 60 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C2 [$1]', round(d['ms_per_step'],3), 'ms; detect', round(p['ms_detect_and_emit'],3), 'anoms', d['result']['anomalies'])"; }
for e in "$@"; do run "$e"; done
