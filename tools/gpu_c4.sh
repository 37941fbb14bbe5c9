#!/bin/bash
# C4 iteration loop on the GPU box: DBSCAN parity tests, the C4 bench line, a kernel trace.  usage: tools/gpu_c4.sh <tag> [ENV=...]
tag=$1; shift
O=/root/repo/gpurun_out/$tag; mkdir -p $O
cd /root/repo
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py -k "dbscan or DBSCAN or job or random or parity" 2>&1 | tail -3
env "$@" python bench.py --config c4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > $O/c4.json
python -c "import json; d=json.loads(open('$O/c4.json').read().strip().splitlines()[-1]); print('C4 ms/step', d['ms_per_step'], d['pipeline'])"
cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d $O -o c4 -- python /root/repo/bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python /root/repo/tools/rocpd_summary.py $O/c4_results.db | head -12
rm -f $O/c4_results.db
