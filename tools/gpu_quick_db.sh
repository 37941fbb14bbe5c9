#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout 300 python bench.py --algo DBSCAN --keys 1000000 --buckets 100 --agg "" --steps 5 --warmup 1 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b$i.json"))
    print("$envs", "ms/step %.3f" % d["ms_per_step"], {k: round(v,3) for k,v in d["pipeline"].items()}, d["result"]["anomalies"])
except Exception as e:
    print("$envs FAILED", e); print(open("$O/b$i.err").read()[-1500:])
PY
done
