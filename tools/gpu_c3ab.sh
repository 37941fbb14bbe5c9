#!/bin/bash
# C3 A/B on one box: ARIMA parity tests under an env set, then the C3 bench line per env set.  usage: tools/gpu_c3ab.sh "ENV=.. ENV=.." ...
cd /root/repo
run() { env $1 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 [$1]', round(d['ms_per_step'],1), 'ms  frac', round(d['arima']['frac'],4))"; }
for e in "$@"; do
  env $e python -m pytest tests/test_gpu_arima.py -m gpu -x -q 2>&1 | tail -1
  run "$e"
done
