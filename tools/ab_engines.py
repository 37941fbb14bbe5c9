#!/usr/bin/env python3
"""N engines with the SAME plan on ONE table in one process, alternating: does the placement of an engine's own buffers move pass B?
(profiles/r4_*ab_c4*: pass B of the C4 job is 0.61 ms on some engine instances and 0.70 on others of the same process.)
usage: python tools/ab_engines.py --config c4 --engines 8 [--rounds 3 --steps 10]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theia_amd import TadEngine  # noqa: E402

CONFIGS = {"c2": dict(algo="EWMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c4": dict(algo="DBSCAN", rows=100_000_000, keys=1_000_000, buckets=100, agg="")}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
ap.add_argument("--engines", type=int, default=8)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
cfg = CONFIGS[args.config]
eng0 = TadEngine(device=0)
n, K, T = cfg["rows"], cfg["keys"], cfg["buckets"]
cols = eng0.synth(0, n, K, T)
jobs = []
for i in range(args.engines):
    e = TadEngine(device=0)
    j = e.prepare(cfg["algo"], cols[0], cols[1], cols[2], K, agg_flow=cfg["agg"], out="device")
    for _ in range(3):
        j.run().close()
    jobs.append((e, j))
pb = [[] for _ in jobs]
for r in range(args.rounds):
    for i, (e, j) in enumerate(jobs):
        acc = 0.0
        for _ in range(args.steps):
            res = j.run()
            acc += res.stats["ms_scatter"]
            res.close()
        pb[i].append(acc / args.steps)
print("%s: pass B per engine instance (ms, median of %d rounds x %d jobs): %s" % (args.config, args.rounds, args.steps,
                                                                                 " ".join("%.3f" % statistics.median(x) for x in pb)))
