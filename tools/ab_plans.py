#!/usr/bin/env python3
"""Same-box, same-process A/B of tad_plan overrides (or of two library builds): one engine per variant, ONE table in HBM, the
variants alternate round by round so that box state (clocks, HBM refresh, neighbours) hits them alike.

usage: python tools/ab_plans.py --config c2|c3|c4 [--rows N --keys K --buckets T] --variants "auto=;exact=histogram=exact" [--rounds 6 --steps 20]
       a variant is name=plan (comma-separated tad_plan fields, empty = the engine decides); `lib:<path>` as a plan field loads another
       library build for that variant (TAD_LIBRARY_PATH semantics of tools/build_variants.py, in a subprocess-free way: separate ctypes handle).
Prints per variant: median / min ms per job over the rounds, and the device-side split (meta / pass B / stage0 / detect) of the last round."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

CONFIGS = {"c2": dict(algo="EWMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c3": dict(algo="ARIMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c4": dict(algo="DBSCAN", rows=100_000_000, keys=1_000_000, buckets=100, agg="")}

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
ap.add_argument("--rows", type=int)
ap.add_argument("--keys", type=int)
ap.add_argument("--buckets", type=int)
ap.add_argument("--variants", default="auto=;exact=histogram=exact")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
cfg = dict(CONFIGS[args.config])
for f in ("rows", "keys", "buckets"):
    if getattr(args, f):
        cfg[f] = getattr(args, f)

variants = []
for item in args.variants.split(";"):
    name, _, plan_s = item.partition("=")
    plan, lib = {}, None
    for kv in filter(None, plan_s.split(",")):
        if kv.startswith("lib:"):
            lib = kv[4:]
            continue
        k, _, v = kv.partition("=")
        plan[k] = int(v) if v.lstrip("-").isdigit() else v
    variants.append((name, plan, lib))

eng0 = TadEngine(device=0)
n, K, T = cfg["rows"], cfg["keys"], cfg["buckets"]
cols = eng0.synth(0, n, K, T)
engines = [(name, TadEngine(device=0, plan=plan, library_path=lib)) for name, plan, lib in variants]
jobs = [(name, e.prepare(cfg["algo"], cols[0], cols[1], cols[2], K, agg_flow=cfg["agg"], out="device")) for name, e in engines]
times = {name: [] for name, _ in jobs}
last = {}
for name, j in jobs:          # warm-up: buffers
    for _ in range(3):
        j.run().close()
for r in range(args.rounds):
    for name, j in (jobs if r % 2 == 0 else jobs[::-1]):
        acc = None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = j.run()
            st = res.stats
            res.close()
            acc = {k: acc[k] + st[k] for k in acc} if acc else {k: st[k] for k in ("ms_meta", "ms_stage0", "ms_scatter", "ms_detect", "ms_total")}
        times[name].append((time.perf_counter() - t0) / args.steps * 1e3)
        last[name] = ({k: v / args.steps for k, v in acc.items()}, st["host_syncs"], st["n_anomalies"])
print("%s: %d rows / %d keys / %d buckets, %d rounds x %d jobs, alternating" % (args.config, n, K, T, args.rounds, args.steps))
for name, _ in jobs:
    dev, syncs, rows = last[name]
    print("  %-12s median %.4f  min %.4f  max %.4f ms/job | device %.4f (meta %.3f, stage0 %.3f of which pass B %.3f, detect+emit %.3f) host syncs %d, rows %d"
          % (name, statistics.median(times[name]), min(times[name]), max(times[name]), dev["ms_total"], dev["ms_meta"], dev["ms_stage0"], dev["ms_scatter"],
             dev["ms_detect"], syncs, rows))
