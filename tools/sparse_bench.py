#!/usr/bin/env python3
"""Timing of the sparse Stage-0 path (stage0_path 4 / 6: hand-written LSD radix sort + reduction; 8 / 9: key-block partition pass + LDS sorts, big tables; tad_sparse.hip), device-resident
columns.  Default shapes = those of tests/test_gpu_sparse.py: (a) 3e6 rows / 2e4 keys with second-resolution timestamps over a day
(gcd 1: the dense grid would be 15.6 GB), (b) the same with two keys of 20 000 points under a 256 MB workspace (length classes).
`--rows N` adds the scale run of the reference's default mode (agg_flow=None keys per connection, anomaly_detection.py:52-61,
109-116): N rows = N/100 connections x ~33 second-resolution points x 3 rows per point.  Run under rocprofv3 by
tools/gpu_measure_r4.sh for the kernel stats / HBM counters under profiles/.
usage: python tools/sparse_bench.py [--steps S] [--rows N] [--only-scale] [--sorts auto,lsd,partition]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--rows", type=int, default=0, help="rows of the scale run (0 = none), e.g. 100000000")
ap.add_argument("--only-scale", action="store_true")
ap.add_argument("--algos", default="EWMA,DBSCAN")
ap.add_argument("--sorts", default="auto", help="comma list of tad_plan.sparse_sort values for the scale run: auto, lsd, partition")
ap.add_argument("--order", default="arbitrary", choices=["arbitrary", "time"],
                help="scale run: rows in arbitrary order, or sorted by flowEndSeconds with the ids handed out in order of first appearance "
                     "(what a read of `flows`, ORDER BY (timeInserted, flowEndSeconds), followed by a dictionary encode delivers)")
ap.add_argument("--lifetime", type=int, default=86400, help="scale run: seconds of the day within which a connection's points fall")
args = ap.parse_args()


def mix64(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def day_table(K, pts_per_key, rows_per_point, seed, long_keys=0, long_len=0):
    rng = np.random.default_rng(seed)
    n_k = np.full(K, pts_per_key)
    if long_keys:
        n_k[rng.choice(K, size=long_keys, replace=False)] = long_len
    pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
    pt = 1660202814 + np.concatenate([np.sort(rng.choice(86400, size=n, replace=False)) for n in n_k]).astype(np.int64)
    base = 1_000_000_000 + (mix64(pk + np.uint64(3)) % np.uint64(3_000_000_000)).astype(np.int64)
    k, t = np.repeat(pk, rows_per_point), np.repeat(pt, rows_per_point)
    v = (np.repeat(base, rows_per_point) + rng.integers(-1_000_000, 1_000_000, size=k.size)).astype(np.uint64)
    order = rng.permutation(k.size)
    return k[order], t[order], v[order]


def connection_table(rows, seed):
    """rows / 100 connections, ~33 points each at random seconds of a day, 3 rows per point, arbitrary row order (vectorised:
    a few seconds of numpy for 1e8 rows)"""
    rng = np.random.default_rng(seed)
    K = max(1, rows // 100)
    P = rows // 3
    pk = rng.integers(0, K, size=P, dtype=np.int64).astype(np.uint64)
    life = max(1, min(args.lifetime, 86400))
    born = rng.integers(0, 86400 - life + 1, size=K, dtype=np.int64)
    pt = 1660202814 + born[pk.astype(np.int64)] + rng.integers(0, life, size=P, dtype=np.int64)
    k, t = np.repeat(pk, 3), np.repeat(pt, 3)
    v = (1_000_000_000 + (mix64(k + np.uint64(3)) % np.uint64(3_000_000_000))).astype(np.uint64) + rng.integers(0, 2_000_000, size=k.size, dtype=np.int64).astype(np.uint64)
    order = rng.permutation(k.size)
    k, t, v = k[order], t[order], v[order]
    if args.order == "time":
        o = np.argsort(t, kind="stable")
        k, t, v = k[o], t[o], v[o]
        _, first = np.unique(k, return_index=True)
        present = np.unique(k)
        newid = np.zeros(K, dtype=np.uint64)
        newid[present.astype(np.int64)[np.argsort(first, kind="stable")]] = np.arange(present.size, dtype=np.uint64)
        k = newid[k.astype(np.int64)]
    return k, t, v, K


def time_jobs(label, eng, k, t, v, K):
    dk, dt, dv = (DeviceArray.from_host(eng, x) for x in (k, t, v))
    for algo in args.algos.split(","):
        for _ in range(2):
            eng.run(algo, dk, dt, dv, K, agg_flow="svc", out="device").close()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = eng.run(algo, dk, dt, dv, K, agg_flow="svc", out="device")
            st = r.stats
            r.close()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print("%s | %s: %d rows, %d points, %d anomalies, stage0_path %d, attempts %d: %.3f ms/job (stage0 %.3f ms = %.2e rows/s = %.0f GB/s on the 24 B/row "
              "of the columns = %.3f of the 8 TB/s peak; detect %.3f) = %.2e rows/s per job"
              % (label, algo, k.size, st["n_points"], st["n_anomalies"], st["stage0_path"], st["stage0_attempts"], ms, st["ms_stage0"], k.size / st["ms_stage0"] * 1e3,
                 24 * k.size / st["ms_stage0"] / 1e6, 24 * k.size / st["ms_stage0"] / 1e6 / 8000.0, st["ms_detect"], k.size / ms * 1e3), flush=True)
    for a in (dk, dt, dv):
        a.free()


if not args.only_scale:
    for label, K, kw, ws in (("gcd-1 day, 2e4 keys x 50 points x 3 rows", 20000, dict(pts_per_key=50, rows_per_point=3, seed=1), 0),
                             ("skewed: + 2 keys of 20000 points, 256 MB workspace", 20000, dict(pts_per_key=50, rows_per_point=3, seed=2, long_keys=2, long_len=20000), 256 << 20)):
        eng = TadEngine(device=0, workspace_limit=ws)
        k, t, v = day_table(K, **kw)
        time_jobs(label, eng, k, t, v, K)
        eng.close()
if args.rows:
    k, t, v, K = connection_table(args.rows, seed=7)
    for sort in args.sorts.split(","):
        eng = TadEngine(device=0, plan={} if sort == "auto" else {"sparse_sort": sort})
        time_jobs("per-connection keys [sparse_sort %s, rows in %s order, lifetime %d s]: %d connections x ~33 second-resolution points x 3 rows"
                  % (sort, args.order, args.lifetime, K), eng, k, t, v, K)
        eng.close()
