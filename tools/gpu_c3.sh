#!/bin/bash
# C3 iteration loop on the GPU box: ARIMA parity tests, the C3 bench line per env set, a kernel trace.  usage: tools/gpu_c3.sh <tag> ["ENV=.. ENV=.."]...
tag=$1; shift
O=/root/repo/gpurun_out/$tag; mkdir -p $O
cd /root/repo
python -m pytest tests/test_gpu_arima.py -m gpu -x -q 2>&1 | tail -2
run() { env $1 python bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 [$1]', round(d['ms_per_step'],1), 'ms  frac', round(d['arima']['frac'],4))"; }
run "X=1"
for e in "$@"; do run "$e"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o c3 -- python /root/repo/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py $O/c3_results.db | head -8
rm -f $O/c3_results.db
