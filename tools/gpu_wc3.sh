#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
run() {  # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/$label.json 2> $O/$label.err
  python - <<PY
import json
try:
    d=json.load(open("$O/$label.json"))
    print("$label", "ms/step %.3f" % d["ms_per_step"], "kern %.3f" % d["roofline"]["avg_kernel_ms"], "stage0 %.3f" % d["pipeline"]["ms_stage0_clear_plus_scatter"], d["roofline"]["kernel"][:16], d["result"]["anomalies"])
except Exception as e:
    print("$label FAILED", e); print(open("$O/$label.err").read()[-800:])
PY
}
for rep in 1 2; do
run c2_wc8_$rep A=1 --
run c2_kp128_wc16_$rep TAD_KP_SHIFT_MIN=7 --
run c2_kp128_wc8_$rep TAD_KP_SHIFT_MIN=7 TAD_WC_SEC=8 --
run c2_kp128_sort_$rep TAD_KP_SHIFT_MIN=7 TAD_PARTB=sort --
run k4e4_wc16_$rep A=1 -- --keys 40000
run k4e4_wc8_$rep TAD_WC_SEC=8 TAD_PARTB=wc -- --keys 40000
run k4e4_wc16r4_$rep TAD_WC_RPT=4 -- --keys 40000
done
