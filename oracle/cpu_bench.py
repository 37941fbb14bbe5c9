"""CPU baseline driver — TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg runs it as a
subprocess; nothing under theia_amd/ may import it).

Times the numpy oracle (oracle/tad_oracle.py, the restatement of plugins/anomaly-detection/anomaly_detection.py:647-710)
on a bounded sample of the synthetic table, two ways:

  * one process — the scalar port, `cores` = 1;
  * key-sharded over P processes — the way the reference job runs on a host: Spark `local[*]` hash-partitions the
    grouped series over all cores and runs the per-key UDFs in parallel (anomaly_detection.py:664-710: groupby(key) →
    UDF per series).  Worker w takes the rows with key mod P == w (the selection is inside the timed region — it is the
    shuffle), runs the whole oracle job on them; wall time = slowest worker.  Pool start-up is outside the timed region
    (the reference's JVM / executor start-up is not counted either).

Prints ONE JSON line.  Run in its own process so that it never forks a process that has the HIP runtime loaded.
"""
import argparse
import json
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


def _noop(_):
    return os.getpid()


def _shard(job):
    w, procs, algo, agg = job
    k, t, v = _DATA["k"], _DATA["t"], _DATA["v"]
    t0 = time.perf_counter()
    sel = np.flatnonzero(k % np.uint64(procs) == np.uint64(w))
    r = orc.run_job(algo, k[sel] // np.uint64(procs), t[sel], v[sel], agg_flow=agg)
    return int(r["n_anomalies"]), int(sel.size), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="EWMA")
    ap.add_argument("--rows", type=int, default=30_000_000)
    ap.add_argument("--keys", type=int, default=30_000)
    ap.add_argument("--buckets", type=int, default=250)
    ap.add_argument("--agg", default="svc")
    ap.add_argument("--procs", type=int, default=0, help="0 = all host cores (capped at 64)")
    ap.add_argument("--skip-single", action="store_true")
    a = ap.parse_args()
    k, t, v = orc.synth_rows(0, a.rows, a.keys, a.buckets)
    out = {"rows": a.rows, "keys": a.keys, "buckets": a.buckets, "algo": a.algo, "host_cores": os.cpu_count()}
    if not a.skip_single:
        t0 = time.perf_counter()
        r = orc.run_job(a.algo, k, t, v, agg_flow=a.agg)
        out["single_s"] = time.perf_counter() - t0
        out["single_anomalies"] = int(r["n_anomalies"])
    procs = a.procs or min(os.cpu_count() or 1, 64)
    if procs > 1:
        _DATA.update(k=k, t=t, v=v)
        ctx = mp.get_context("fork")
        with ctx.Pool(procs) as pool:
            pool.map(_noop, range(procs))          # workers are up before the clock starts
            t0 = time.perf_counter()
            res = pool.map(_shard, [(w, procs, a.algo, a.agg) for w in range(procs)], chunksize=1)
            out["multi_s"] = time.perf_counter() - t0
        out["procs"] = procs
        out["multi_anomalies"] = sum(r[0] for r in res)
        out["multi_rows"] = sum(r[1] for r in res)
        out["slowest_worker_s"] = max(r[2] for r in res)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
