"""CPU baseline driver — TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg runs it as a
subprocess; nothing under theia_amd/ may import it).

Times the numpy oracle (oracle/tad_oracle.py, the restatement of plugins/anomaly-detection/anomaly_detection.py:647-710;
ARIMA: oracle/arima_exact.c) on a bounded sample of the synthetic table, two ways:

  * one process — the scalar port, `cores` = 1 (on `--single-rows` rows);
  * key-sharded over P processes, P = ALL host cores by default — the way the reference job runs on a host: Spark
    `local[*]` hash-partitions the grouped series over the cores and runs the per-key UDFs in parallel
    (anomaly_detection.py:664-710: groupby(key) -> UDF per series).  The shuffle is a real two-phase exchange through
    shared memory and is inside the timed region: phase 1, worker w buckets ITS slice of the rows by owner = key mod P
    (every row is read once); phase 2, worker o gathers its bucket from every slice and runs the whole oracle job on
    it.  Wall time = both phases.  Pool start-up and table generation are outside the timed region (the reference's
    JVM / executor start-up and the table sitting in ClickHouse are not counted either).

Prints ONE JSON line.  Run in its own process so that it never forks a process that has the HIP runtime loaded.
"""
import argparse
import json
import mmap
import multiprocessing as mp
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # one thread per process: the parallelism is across keys
    os.environ.setdefault(_v, "1")

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tad_oracle as orc   # noqa: E402

_DATA = {}


def _shared(n, dtype):
    """anonymous MAP_SHARED array: written by forked workers, visible to all of them and to the parent"""
    nbytes = max(1, n) * np.dtype(dtype).itemsize
    return np.frombuffer(mmap.mmap(-1, nbytes), dtype=dtype, count=n)


def _noop(_):
    return os.getpid()


def _bounds(n, procs, w):
    return n * w // procs, n * (w + 1) // procs


def _bucket(w):
    """phase 1 (map side of the shuffle): rows [lo, hi) grouped by owner, in place in the shared exchange buffers"""
    procs = _DATA["procs"]
    k, t, v = _DATA["k"], _DATA["t"], _DATA["v"]
    lo, hi = _bounds(k.size, procs, w)
    owner = (k[lo:hi] % np.uint64(procs)).astype(np.int64)
    order = np.argsort(owner, kind="stable")
    _DATA["xk"][lo:hi] = (k[lo:hi] // np.uint64(procs))[order]
    _DATA["xt"][lo:hi] = t[lo:hi][order]
    _DATA["xv"][lo:hi] = v[lo:hi][order]
    _DATA["counts"][w * procs:(w + 1) * procs] = np.bincount(owner, minlength=procs)
    return hi - lo


def _job(o):
    """phase 2 (reduce side): gather bucket o of every slice, run the job on it"""
    procs, algo, agg = _DATA["procs"], _DATA["algo"], _DATA["agg"]
    counts = _DATA["counts"].reshape(procs, procs)
    n = _DATA["k"].size
    parts = []
    for w in range(procs):
        lo, _ = _bounds(n, procs, w)
        a = lo + int(counts[w, :o].sum())
        parts.append((a, a + int(counts[w, o])))
    kk = np.concatenate([_DATA["xk"][a:b] for a, b in parts])
    tt = np.concatenate([_DATA["xt"][a:b] for a, b in parts])
    vv = np.concatenate([_DATA["xv"][a:b] for a, b in parts])
    r = orc.run_job(algo, kk, tt, vv, agg_flow=agg)
    return int(r["n_anomalies"]), int(kk.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="EWMA")
    ap.add_argument("--rows", type=int, default=30_000_000)
    ap.add_argument("--keys", type=int, default=30_000)
    ap.add_argument("--buckets", type=int, default=250)
    ap.add_argument("--agg", default="svc")
    ap.add_argument("--procs", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--single-rows", type=int, default=0, help="rows of the one-process run (a prefix-shaped sample with the same "
                                                                "rows per key); 0 = --rows")
    ap.add_argument("--skip-single", action="store_true")
    a = ap.parse_args()
    procs = a.procs or (os.cpu_count() or 1)
    k, t, v = orc.synth_rows_parallel(a.rows, a.keys, a.buckets, procs=min(procs, 32))
    out = {"rows": a.rows, "keys": a.keys, "buckets": a.buckets, "algo": a.algo, "host_cores": os.cpu_count()}
    if not a.skip_single:
        srows = a.single_rows or a.rows
        if srows < a.rows:      # same rows-per-key: a table of srows rows over proportionally fewer keys
            skeys = max(1, int(a.keys * srows / a.rows))
            sk, st_, sv = orc.synth_rows(0, srows, skeys, a.buckets)
        else:
            srows, sk, st_, sv = a.rows, k, t, v
        t0 = time.perf_counter()
        r = orc.run_job(a.algo, sk, st_, sv, agg_flow=a.agg)
        out["single_s"] = time.perf_counter() - t0
        out["single_rows"] = srows
        out["single_anomalies"] = int(r["n_anomalies"])
    if procs > 1:
        _DATA.update(k=k, t=t, v=v, procs=procs, algo=a.algo, agg=a.agg,
                     xk=_shared(k.size, np.uint64), xt=_shared(k.size, np.int64), xv=_shared(k.size, np.uint64),
                     counts=_shared(procs * procs, np.int64))
        ctx = mp.get_context("fork")
        with ctx.Pool(procs) as pool:
            pool.map(_noop, range(procs))          # workers are up before the clock starts
            t0 = time.perf_counter()
            pool.map(_bucket, range(procs), chunksize=1)
            t1 = time.perf_counter()
            res = pool.map(_job, range(procs), chunksize=1)
            t2 = time.perf_counter()
        out["multi_s"] = t2 - t0
        out["shuffle_s"] = t1 - t0
        out["procs"] = procs
        out["multi_anomalies"] = sum(r[0] for r in res)
        out["multi_rows"] = sum(r[1] for r in res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
