#!/bin/bash
# C2 iteration loop on the GPU box: Stage-0 parity tests, the C2 bench line per env set, a kernel trace.  usage: tools/gpu_c2.sh <tag> ["ENV=.."]...
tag=$1; shift
O=/root/repo/gpurun_out/$tag; mkdir -p $O
cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_job.py tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -4
run() { env $1 python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']; print('C2 [$1]', round(d['ms_per_step'],3), 'ms; meta', round(p['ms_meta'],3), 'stage0', round(p['ms_stage0_clear_plus_scatter'],3), 'partition', round(d['roofline']['avg_kernel_ms'],3), 'detect', round(p['ms_detect_and_emit'],3), 'anoms', d['result']['anomalies'])"; }
run "X=1"
for e in "$@"; do run "$e"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o c2 -- python /root/repo/bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py $O/c2_results.db | head -12
rm -f $O/c2_results.db
