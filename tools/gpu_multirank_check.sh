#!/bin/bash
# Exercise bench.py's N>1 path on a 1-GPU box: 2 ranks share cuda:0, collectives over gloo on host tensors.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
TAD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --rows 20000000 --keys 20000 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "rc=$?"; tail -c 1500 $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
timeout 300 python bench.py --steps 3 --warmup 1 --rows 20000000 --keys 20000 --no-cpu-baseline > $O/bench_1rank.json 2>/dev/null
python - <<PY
import json
a=json.load(open("$O/bench_2rank_gloo.json")); b=json.load(open("$O/bench_1rank.json"))
print("2-rank result", a["result"], "n_gpus", a["n_gpus"], a["value"])
print("1-rank result", b["result"], b["value"])
PY
