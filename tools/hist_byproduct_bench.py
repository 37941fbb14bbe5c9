#!/usr/bin/env python3
"""C4 on the INGEST path: the rows' keys go through tad_factorize_hist (as they do when they come from ClickHouse), the DBSCAN job then sizes
pass B's regions from the factorisation's key-bin histogram instead of counting the key column again in pass A (tad.h:tad_key_hist).
Same process, alternating: the job with and without the histogram; prints ms per job, the lattice / histogram pass (ms_meta) and whether the
rows agree.  `--only with|without --jobs N`: N jobs of one kind (for rocprofv3 --pmc / --kernel-trace passes).
usage: python tools/hist_byproduct_bench.py [--rows 100000000 --keys 1000000 --buckets 100 --algo DBSCAN]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import TadEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--keys", type=int, default=1_000_000)
ap.add_argument("--buckets", type=int, default=100)
ap.add_argument("--algo", default="DBSCAN")
ap.add_argument("--agg", default="")
ap.add_argument("--only", default="", choices=["", "with", "without"])
ap.add_argument("--jobs", type=int, default=20)
ap.add_argument("--rounds", type=int, default=6)
args = ap.parse_args()
eng = TadEngine(device=0)
n, K, T = args.rows, args.keys, args.buckets
raw, t, v = eng.synth(0, n, K, T)                      # the synthetic table's key column stands in for the raw key tuples
t0 = time.perf_counter()
key, _, first, hist = eng.factorize([raw], with_hist=True)
t_fz = time.perf_counter() - t0
t0 = time.perf_counter()
key2, _, first2 = eng.factorize([raw])
t_fz_plain = time.perf_counter() - t0
nk = first.n
print("%d rows, %d keys, %d buckets: tad_factorize_hist %.2f ms (first call), tad_factorize %.2f ms; histogram %d workgroups x %d bins (shift %d)"
      % (n, nk, T, t_fz * 1e3, t_fz_plain * 1e3, hist.c.workgroups, hist.c.nbins, hist.c.shift), flush=True)
jobs = {"with": eng.prepare(args.algo, key, t, v, nk, agg_flow=args.agg, out="device", key_hist=hist),
        "without": eng.prepare(args.algo, key, t, v, nk, agg_flow=args.agg, out="device")}
if args.only:
    for _ in range(args.jobs):
        r = jobs[args.only].run()
        st = r.stats
        r.close()
    print(args.only, "hist_sampled", st["hist_sampled"], "ms_meta %.3f ms_total %.3f" % (st["ms_meta"], st["ms_total"]))
    sys.exit(0)
rows = {}
for name, j in jobs.items():
    for _ in range(3):
        r = j.run()
        rows[name] = (r.n_rows, r.stats["n_points"], r.stats["hist_sampled"])
        r.close()
assert rows["with"][:2] == rows["without"][:2] and rows["with"][2] == 2 and rows["without"][2] != 2, rows
times = {k: [] for k in jobs}
meta = {k: [] for k in jobs}
for rnd in range(args.rounds):
    for name in (("with", "without") if rnd % 2 == 0 else ("without", "with")):
        t0 = time.perf_counter()
        for _ in range(args.jobs):
            r = jobs[name].run()
            st = r.stats
            r.close()
        times[name].append((time.perf_counter() - t0) / args.jobs * 1e3)
        meta[name].append(st["ms_meta"])
for name in jobs:
    print("  %-8s median %.4f  min %.4f  max %.4f ms/job | lattice + histogram pass %.3f ms | %d anomaly rows, %d points"
          % (name, statistics.median(times[name]), min(times[name]), max(times[name]), statistics.median(meta[name]), rows[name][0], rows[name][1]))
print("  with vs without: %+.2f %%" % ((statistics.median(times["with"]) / statistics.median(times["without"]) - 1) * 100))
