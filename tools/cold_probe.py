#!/usr/bin/env python3
"""What a controller sees (controller.go:499-523: every CR is a new job on new data): the FIRST job of an engine, then jobs that
bring DIFFERENT column buffers every time (K tables of one shape, different seeds, round robin), then the same columns repeated.

usage: python tools/cold_probe.py [--config c2|c4] [--plan histogram=exact] [--tables 4] [--jobs 40]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import SYNTH_SEED  # noqa: E402

CONFIGS = {"c2": dict(algo="EWMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c4": dict(algo="DBSCAN", rows=100_000_000, keys=1_000_000, buckets=100, agg="")}

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
ap.add_argument("--plan", default="")
ap.add_argument("--tables", type=int, default=4)
ap.add_argument("--jobs", type=int, default=40)
args = ap.parse_args()
cfg = CONFIGS[args.config]
plan = {}
for kv in filter(None, args.plan.split(",")):
    k, _, v = kv.partition("=")
    plan[k] = int(v) if v.lstrip("-").isdigit() else v

t0 = time.perf_counter()
eng = TadEngine(device=0, plan=plan)
t_create = (time.perf_counter() - t0) * 1e3
n, K, T = cfg["rows"], cfg["keys"], cfg["buckets"]
tables = [eng.synth(0, n, K, T, seed=SYNTH_SEED + 977 * i) for i in range(args.tables)]
jobs = [eng.prepare(cfg["algo"], c[0], c[1], c[2], K, agg_flow=cfg["agg"], out="device") for c in tables]


def one(j):
    t = time.perf_counter()
    res = j.run()
    st = res.stats
    res.close()
    return (time.perf_counter() - t) * 1e3, st


print("%s, plan {%s}: engine create %.2f ms" % (args.config, args.plan, t_create))
ms, st = one(jobs[0])
print("  first job of the engine: %.3f ms wall (device %.3f, host syncs %d)" % (ms, st["ms_total"], st["host_syncs"]))
ms, st = one(jobs[1 % len(jobs)])
print("  second job (other columns): %.3f ms wall (device %.3f, host syncs %d)" % (ms, st["ms_total"], st["host_syncs"]))
fresh, passb, syncs = [], [], []
for i in range(args.jobs):
    ms, st = one(jobs[(i + 2) % len(jobs)])
    fresh.append(ms)
    passb.append(st["ms_scatter"])
    syncs.append(st["host_syncs"])
print("  fresh columns every job (%d tables round robin, %d jobs): median %.4f  mean %.4f  max %.4f ms/job; pass B median %.3f; host syncs %s"
      % (len(jobs), args.jobs, statistics.median(fresh), statistics.mean(fresh), max(fresh), statistics.median(passb), sorted(set(syncs))))
same, passb = [], []
for i in range(args.jobs):
    ms, st = one(jobs[0])
    same.append(ms)
    passb.append(st["ms_scatter"])
print("  same columns every job (%d jobs): median %.4f  mean %.4f  max %.4f ms/job; pass B median %.3f; host syncs %d"
      % (args.jobs, statistics.median(same), statistics.mean(same), max(same), statistics.median(passb), st["host_syncs"]))
