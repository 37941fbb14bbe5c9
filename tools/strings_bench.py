#!/usr/bin/env python3
"""tad_encode_strings at scale: an Arrow string column (pod names of ~25 bytes) -> dictionary codes on the GPU, (a) device-resident offsets +
bytes (the kernel rate), (b) the same column handed over in host memory (PCIe-inclusive: what theia_amd.clickhouse.query_columns(engine=...)
pays per chunk), for a low-cardinality column (2e4 pods: the small L2-resident table) and a high-cardinality one (rows/50 distinct values:
the table grows to 2 n slots on the second attempt).  For comparison Arrow's dictionary_encode of the same column on ONE host core.
usage: python tools/strings_bench.py [--rows 100000000] [--steps 3]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.compute as pc  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--arrow-rows", type=int, default=10_000_000)
ap.add_argument("--host-rows", type=int, default=20_000_000, help="rows of the host-memory (PCIe-inclusive) run")
ap.add_argument("--shapes", default="low,high", help="low = 2e4 distinct values, high = rows/50 distinct values")
args = ap.parse_args()
n = args.rows
rng = np.random.default_rng(5)
eng = TadEngine(device=0)


def column(n, distinct):
    """n rows drawn from `distinct` pod-like names (deployment-replicaset-pod: 18..34 bytes), Arrow large_string (int64 offsets)"""
    ids = np.arange(distinct)
    h = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
    vocab = pa.array(["%s-%010x-%05x" % (("antrea-agent", "coredns", "flow-aggregator", "clickhouse-shard0", "web")[i % 5], int(h[i]), i % 1048576)
                      for i in range(distinct)], pa.large_string())
    return vocab.take(pa.array(rng.integers(0, distinct, size=n)))


for label, distinct in [sh for sh, tag in ((("2e4 distinct pod names (small table)", 20_000), "low"), (("rows/50 distinct values (table grows)", max(1, n // 50)), "high"))
                        if tag in args.shapes.split(",")]:
    arr = column(n, distinct)
    _, obuf, dbuf = arr.buffers()
    offsets = np.frombuffer(obuf, dtype=np.int64)[: n + 1]
    data = np.frombuffer(dbuf, dtype=np.uint8)
    nbytes = int(offsets[-1])
    d_off = DeviceArray.from_host(eng, offsets)
    d_data = DeviceArray.from_host(eng, np.concatenate([data[:nbytes], np.zeros(-nbytes % 8, np.uint8)]))
    for _ in range(2):
        codes, first = eng.encode_strings((d_off, d_data), max_values=1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        codes, first = eng.encode_strings((d_off, d_data), max_values=1)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    nv = int(codes.to_host().max()) + 1
    alg = nbytes + 8 * (n + 1) + 8 * n          # bytes + offsets in, codes out
    # host memory, PCIe-inclusive
    m = min(n, args.host_rows)
    sl = arr.slice(0, m)
    eng.encode_strings(sl, max_values=1)
    t0 = time.perf_counter()
    eng.encode_strings(sl, max_values=1)
    hs = time.perf_counter() - t0
    # Arrow on one core
    a = min(n, args.arrow_rows)
    pa.set_cpu_count(1)
    t0 = time.perf_counter()
    pc.dictionary_encode(arr.slice(0, a))
    ps = time.perf_counter() - t0
    print("%s | %d rows, %.1f B/row of strings, %d values: %.2f ms = %.2e rows/s (%.0f GB/s on the %.1f B/row of offsets + bytes in and codes out = %.3f of 8 TB/s) | "
          "host memory, %d rows: %.1f ms = %.2e rows/s | Arrow dictionary_encode, one core, %d rows: %.2e rows/s"
          % (label, n, nbytes / n, nv, ms, n / ms * 1e3, alg / ms / 1e6, alg / n, alg / ms / 1e6 / 8000, m, hs * 1e3, m / hs, a, a / ps), flush=True)
    d_off.free(); d_data.free(); codes.free(); first.free()
    del arr, offsets, data
eng.close()
