#!/usr/bin/env python3
"""Placement-neutral A/B of tad_plan overrides: ONE engine, ONE table, ONE set of workspace buffers — the plan is swapped with
tad_engine_set_plan between blocks of jobs, so both variants write the same record buffer from the same columns (what
tools/ab_plans.py cannot give: its variants are separate engines whose buffers may land in different placement classes).

usage: python tools/ab_one_engine.py --config c2|c4 --variants "auto=;exact=histogram=exact" [--base partition_pass=wc] [--alternations 8 --steps 20]
Prints per variant the median / min / max ms per job over the alternations and the relative difference of the medians."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from theia_amd import TadEngine  # noqa: E402

CONFIGS = {"c2": dict(algo="EWMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c4": dict(algo="DBSCAN", rows=100_000_000, keys=1_000_000, buckets=100, agg="")}


def parse_plan(s):
    plan = {}
    for kv in filter(None, s.split(",")):
        k, _, v = kv.partition("=")
        plan[k] = int(v) if v.lstrip("-").isdigit() else v
    return plan


ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
ap.add_argument("--variants", default="auto=;exact=histogram=exact")
ap.add_argument("--base", default="", help="plan fields every variant carries")
ap.add_argument("--alternations", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
cfg = CONFIGS[args.config]
base = parse_plan(args.base)
variants = []
for item in args.variants.split(";"):
    name, _, plan_s = item.partition("=")
    p = dict(base)
    p.update(parse_plan(plan_s))
    variants.append((name, p))

eng = TadEngine(device=0, plan=base)
n, K, T = cfg["rows"], cfg["keys"], cfg["buckets"]
cols = eng.synth(0, n, K, T)
job = eng.prepare(cfg["algo"], cols[0], cols[1], cols[2], K, agg_flow=cfg["agg"], out="device")
for _ in range(3):
    job.run().close()
times = {name: [] for name, _ in variants}
last = {}
for r in range(args.alternations):
    for name, plan in (variants if r % 2 == 0 else variants[::-1]):
        eng.set_plan(**plan)
        job.run().close()          # the first job after a swap is not timed
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = job.run()
            st = res.stats
            res.close()
        times[name].append((time.perf_counter() - t0) / args.steps * 1e3)
        last[name] = st
print("%s: %d rows / %d keys / %d buckets, ONE engine, %d alternations x %d jobs, base plan {%s}" % (args.config, n, K, T, args.alternations, args.steps, args.base))
med = {}
for name, _ in variants:
    st = last[name]
    med[name] = statistics.median(times[name])
    print("  %-10s median %.4f  min %.4f  max %.4f ms/job | pass B %.3f, device %.3f, host syncs %d, rows %d"
          % (name, med[name], min(times[name]), max(times[name]), st["ms_scatter"], st["ms_total"], st["host_syncs"], st["n_anomalies"]))
names = [v[0] for v in variants]
for other in names[1:]:
    print("  %s vs %s: %+.2f %% (negative = %s faster)" % (names[0], other, (med[names[0]] / med[other] - 1) * 100, names[0]))
