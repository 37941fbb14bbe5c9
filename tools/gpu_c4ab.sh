#!/bin/bash
# C4 A/B on one box: bench line per env set.  usage: tools/gpu_c4ab.sh "ENV=.." ...
cd /root/repo
run() { env $1 python bench.py --config c4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C4 [$1]', round(d['ms_per_step'],3), 'ms; stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'partition', round(d['roofline']['avg_kernel_ms'],3), 'detect', round(p['ms_detect_and_emit'],3))"; }
for e in "$@"; do run "$e"; done
