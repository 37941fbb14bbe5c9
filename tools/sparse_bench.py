#!/usr/bin/env python3
"""Timing of the sparse Stage-0 path (stage0_path 4 / 6) on the two table shapes of tests/test_gpu_sparse.py, device-resident
columns: (a) 3e6 rows / 2e4 keys with second-resolution timestamps over a day (gcd 1: the dense grid would be 15.6 GB),
(b) the same with two keys of 20 000 points under a 256 MB workspace (length classes).  Run under rocprofv3 by
tools/gpu_measure_r3.sh for the kernel stats / HBM counters of profiles/r3_*_sparse_*.   usage: python tools/sparse_bench.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


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


for label, K, kw, ws in (("gcd-1 day, 2e4 keys x 50 points x 3 rows", 20000, dict(pts_per_key=50, rows_per_point=3, seed=1), 0),
                         ("skewed: + 2 keys of 20000 points, 256 MB workspace", 20000, dict(pts_per_key=50, rows_per_point=3, seed=2, long_keys=2, long_len=20000), 256 << 20)):
    eng = TadEngine(device=0, workspace_limit=ws)
    k, t, v = day_table(K, **kw)
    dk, dt, dv = (DeviceArray.from_host(eng, x) for x in (k, t, v))
    for algo in ("EWMA", "DBSCAN"):
        for _ in range(2):
            eng.run(algo, dk, dt, dv, K, agg_flow="svc", out="device").close()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = eng.run(algo, dk, dt, dv, K, agg_flow="svc", out="device")
            st = r.stats
            r.close()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("%s | %s: %d rows, %d points, %d anomalies, stage0_path %d: %.3f ms/job (stage0 %.3f, detect %.3f) = %.2e rows/s, %.0f B of HBM column data per row at 8 TB/s would be %.4f ms"
              % (label, algo, k.size, st["n_points"], st["n_anomalies"], st["stage0_path"], ms, st["ms_stage0"], st["ms_detect"], k.size / ms * 1e3, 24, 24 * k.size / 8e12 * 1e3))
    eng.close()
