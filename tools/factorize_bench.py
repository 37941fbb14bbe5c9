#!/usr/bin/env python3
"""tad_factorize at scale, device-resident key columns: (a) one key column, 1e5 distinct values (svc / external mode), (b) the six key
columns of the reference's default mode (sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, protocolIdentifier,
flowStartSeconds: anomaly_detection.py:52-61) with rows/100 distinct connections, (c) pod mode: two tuples of two columns per row.
For comparison the pandas factorisation of the same columns on ONE host core (a 1e7-row sample).
usage: python tools/factorize_bench.py [--rows 100000000] [--steps 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--pandas-rows", type=int, default=10_000_000)
args = ap.parse_args()
n = args.rows
rng = np.random.default_rng(1)
eng = TadEngine(device=0)


def connections(n, K):
    conn = rng.integers(0, K, size=n)
    h = (conn.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15))
    return [(h >> np.uint64(40)).astype(np.int64) % 5000, (h >> np.uint64(7)).astype(np.int64) % 60000, (h >> np.uint64(23)).astype(np.int64) % 3000,
            (h >> np.uint64(13)).astype(np.int64) % 1000, (conn % 3).astype(np.int64) * 11 + 6, 1660000000 + (conn // 7).astype(np.int64)]


shapes = [("1 column, 1e5 distinct (svc / external mode)", [rng.integers(0, 100_000, size=n).astype(np.int64)], None),
          ("6 columns, rows/100 connections (default mode)", connections(n, max(1, n // 100)), None),
          ("pod mode: 2 x 2 columns, 2e4 pods", [rng.integers(0, 40, size=n).astype(np.int64), rng.integers(0, 20_000, size=n).astype(np.int64)],
           [rng.integers(0, 40, size=n).astype(np.int64), rng.integers(0, 20_000, size=n).astype(np.int64)])]
for label, cols, colsb in shapes:
    da = [DeviceArray.from_host(eng, c) for c in cols]
    db = [DeviceArray.from_host(eng, c) for c in colsb] if colsb else None
    for _ in range(2):
        k1, k2, fr = eng.factorize(da, None, db, None, max_keys=1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        k1, k2, fr = eng.factorize(da, None, db, None, max_keys=1)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    nk = int(k1.to_host().max()) + 1 if not colsb else int(max(k1.to_host().max(), k2.to_host().max())) + 1
    import pandas as pd
    m = min(n, args.pandas_rows)
    t0 = time.perf_counter()
    if len(cols) > 1:
        pd.MultiIndex.from_arrays([c[:m] for c in cols]).factorize()
    else:
        pd.factorize(cols[0][:m])
    ps = time.perf_counter() - t0
    nb = 8 * len(cols) * (2 if colsb else 1)
    print("%s | %d rows, %d keys: %.2f ms = %.2e rows/s (%d B/row of key columns read twice + %d B/row of ids written: %.0f GB/s of 8000) | pandas, one core, %d rows: %.2e rows/s"
          % (label, n, nk, ms, n / ms * 1e3, nb, 8 * (2 if colsb else 1), (2 * nb + 8 * (2 if colsb else 1)) * n / ms / 1e6, m, m / ps), flush=True)
    for a in da + (db or []):
        a.free()
eng.close()
