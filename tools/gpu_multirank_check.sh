#!/bin/bash
# Exercise bench.py's N>1 path on a 1-GPU box: 2 ranks share cuda:0, collectives over gloo on host tensors.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
TAD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --rows 20000000 --keys 20000 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "rc=$?"; tail -c 1500 $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
timeout 300 python bench.py --steps 3 --warmup 1 --rows 20000000 --keys 20000 --no-cpu-baseline > $O/bench_1rank.json 2>/dev/null
TAD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --ingest rows --steps 3 --warmup 1 --rows 20000000 --keys 20000 > $O/bench_2rank_rows.json 2> $O/bench_2rank_rows.err; echo "rows-mode rc=$?"; tail -3 $O/bench_2rank_rows.err
timeout 300 python bench.py --ingest rows --steps 3 --warmup 1 --rows 20000000 --keys 20000 --no-cpu-baseline > $O/bench_1rank_rows.json 2> $O/bench_1rank_rows.err; echo "rows-mode 1 rank rc=$?"; tail -3 $O/bench_1rank_rows.err
python - <<PY
import json
def last_json(p):
    for line in open(p):
        if line.startswith("{"):
            d = json.loads(line)
    return d
for f in ("bench_2rank_gloo", "bench_1rank", "bench_2rank_rows", "bench_1rank_rows"):
    try:
        d = last_json("$O/%s.json" % f); print(f, "n_gpus", d["n_gpus"], "value %.3e" % d["value"], "ms/step %.3f" % d["ms_per_step"], d["result"])
    except Exception as e: print(f, "FAILED", e)
PY
