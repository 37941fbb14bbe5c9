#!/usr/bin/env python3
"""bench.py — flow-records/s of the Throughput Anomaly Detection hot path on MI355X.

One "step" = one full job (tad_run through the C ABI) over one synthetic batch that is already
resident in HBM: Stage-0 GROUP BY (key, flowEndSeconds) -> per-key stddev_samp -> detector ->
compaction of the anomalous points into device-resident result columns.

Workload at N=1 (`--config c2`, the default) = BASELINE.json configs[1]: EWMA on 1e8 rows / 1e5 flow keys / 250 time
buckets, sum(throughput) (mode svc), the deterministic synthetic table of SURVEY.md §8d.  The same JSON line carries
`other_configs`: C4 (configs[3]: DBSCAN, 1e6 keys x 100 buckets, max(throughput) = mode None) and C3 (configs[2]: ARIMA
on the C2 table), each timed the same way (W warm-up steps, K timed steps bracketed by synchronisation) with its own
roofline — so that those rates are measured by whoever runs this file, not quoted.
N>1 (`python bench.py --gpus N` starts the N ranks itself; under an external torchrun — WORLD_SIZE set — it is one of them;
one rank per GPU): weak scaling — every rank owns the key shard `key mod N == rank`
(1e8 rows / 1e5 keys per rank, pre-sharded by key as SURVEY.md §8e allows), no data-path collective;
per step ONE RCCL all-gather of 9 doubles per rank: the counters [anomalies, keys, points, rows, ...] (the global
`count() == 0` sentinel decision, anomaly_detection.py:395) and the (n, mean, M2) moments (global sigma).
`--config c5` = BASELINE.json configs[4]: the 1e9-row / 1e6-key table split over the N ranks (1e9/N rows, 1e6/N keys
per rank — strong scaling), one step = the EWMA job THEN the ARIMA job on the rank's shard; `--ingest rows` makes every
rank start from an arbitrary row slice and ship partial points to the key owners with one all-to-all(v) first.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (Stage-0 pass B, the row partition pass):
achieved = 24 B/row x rows per launch / the kernel's average duration measured with HIP events on the
engine's stream (tad_stats.ms_scatter); `traffic` = that kernel's HBM bytes per launch from the committed
rocprofv3 PMC passes (profiles/pmc_latest.json: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE;
separate --pmc runs of this same command).  `roofline.frac` prices the DOMINANT KERNEL; `roofline.frac_whole_run` is SURVEY.md 8d's own definition — (24 B x rows + 40 B x anomalies)
per job over the WALL time of a step (`ms_per_step`, launch gaps and host round trips included) against the same peak.
`value` is what a controller sees from a warm engine (controller.go:499-523: every CR is a new job on new data): the timed steps go
ROUND ROBIN over `config.tables` (4) synthetic tables of the workload's shape with different seeds in different device buffers, so
no step meets the columns of the step before it; `same_columns` is the old loop (one table, every step) beside it, `cold` the FIRST
job of the engine in the process (allocations included).  `concurrency` (N = 1, default line): aggregate rows/s with 1 / 2 / 4 host
threads submitting C2 jobs to the ONE engine (job contexts = stream pool, tad.h ABI 12), and the latency of C2 EWMA jobs submitted
while a C3 ARIMA job runs.  `row_orders` (N = 1, default line): the C2 job on the SAME rows in the orders a caller can bring them in —
ids handed out in order of first appearance (any dictionary encoder), rows by time (`flows` is ORDER BY (timeInserted, flowEndSeconds),
create_table.sh:85), by key (a GROUP BY result), keys alive for a tenth of the table — and with an --end-time window; the synthetic
table's rows are in arbitrary order with hashed ids.
`cpu_baseline` = the oracle (numpy port of the reference
job) timed on this box's host cores on a bounded sample.  ARIMA lines add `arima`: fits/s and the
FP64 flop rate from the engine's Kalman-step counter (16 flop per step of the recursion the kernel executes; the
60-flop-equivalent of SURVEY.md 8d's three-state model beside it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
FLOP_PER_KALMAN_STEP_3STATE = 60   # SURVEY.md 8d's model (one predict + update of the textbook three-state filter): rounds 1-2 priced on it
# the contract since round 3 (tad_arima.hip:kfc_step4, the collapsed recursion): 13 flops per chain and step (v, g, w, q fma, product,
# a1 = add + mul + fma, p' fma, p - p') plus a quarter of the batched inversion (9 multiplications + 1 division per four chains) are
# what is EXECUTED; `frac` is priced on those flops, the 60-flop-equivalent fraction is reported next to it so that the rounds
# stay comparable
FLOP_PER_KALMAN_STEP = 16
BYTES_PER_ROW = 24      # key_id u64 + flow_end_s i64 + value u64, read once
BYTES_PER_ANOMALY = 40  # key_id, flow_end_s, throughput, algo_calc, stddev

CONFIGS = {   # BASELINE.json configs[1..4]
    "c2": dict(algos=("EWMA",), rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
    "c3": dict(algos=("ARIMA",), rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
    "c4": dict(algos=("DBSCAN",), rows=100_000_000, keys=1_000_000, buckets=100, agg=""),
    "c5": dict(algos=("EWMA", "ARIMA"), rows=1_000_000_000, keys=1_000_000, buckets=250, agg="svc"),   # totals, split over the ranks
}
STAGE0_KERNEL = {2: "k_partition (Stage-0 v2, row partition pass, sort-by-tile)",
                 3: "k_partition_wc (Stage-0 v2, row partition pass, write-combining)",
                 5: "k_partition_wc (Stage-0 v2 two-level, level-1 row partition pass, write-combining)"}


def cpu_baseline(algo, rows, keys, buckets, agg, single_rows=0):
    """The oracle (numpy restatement of the reference job; ARIMA: oracle/arima_exact.c) on a bounded sample of the same
    workload, in its own process (oracle/cpu_bench.py): one process, and key-sharded over ALL host cores the way Spark
    local[*] runs the per-key UDFs (two-phase shuffle through shared memory, timed).  `value` is the faster of the two,
    `cores` the processes it used.  kind "port": the reference's PySpark job cannot run here (no JVM / pyspark /
    statsmodels in the image) — a GPU-over-numpy ratio says nothing about kernel quality; the roofline fractions do."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--algo", algo, "--rows", str(rows), "--keys", str(keys),
           "--buckets", str(buckets), "--agg", agg]
    if single_rows and single_rows < rows:
        cmd += ["--single-rows", str(single_rows)]
    r = json.loads(subprocess.run(cmd, check=True, capture_output=True, timeout=1200, text=True).stdout.strip().splitlines()[-1])
    single = r["single_rows"] / r["single_s"] if "single_s" in r else None
    multi = rows / r["multi_s"] if "multi_s" in r else None
    use_multi = multi is not None and (single is None or multi > single)
    out = {"value": multi if use_multi else single, "unit": "flow-records/s", "cores": r["procs"] if use_multi else 1, "kind": "port",
           "sample": "%s, %d rows / %d keys / %d buckets of the same synthetic table (same rows-per-key as the GPU workload), numpy "
                     "oracle; %s" % (algo, rows if use_multi else r["single_rows"], keys, buckets,
                                     "key-sharded over %d processes = all host cores (two-phase shuffle by key mod P through shared "
                                     "memory %.2f s + per-shard jobs, both timed), %.1f s" % (r["procs"], r["shuffle_s"], r["multi_s"])
                                     if use_multi else "single process, %.1f s" % r["single_s"]),
           "host_cores": r["host_cores"], "anomalies": r.get("multi_anomalies", r.get("single_anomalies"))}
    # the reference job evaluates its lineage twice (the `.count()` action of anomaly_detection.py:395, then the JDBC
    # write, :713-726): `value` is the de-duplicated (1x) rate, this is the as-written one
    out["as_written_2x_value"] = out["value"] / 2.0
    if single is not None:
        out["single_core_value"] = single
    if multi is not None:
        out["all_cores_value"] = multi
    ref = reference_functions(algo)
    if ref is not None:
        out["reference_functions"] = ref
    return out


def reference_functions(algo):
    """The reference's OWN per-series functions (calculate_ewma[_anomaly], calculate_dbscan[_anomaly]) timed by oracle/ref_baseline.py
    in the BUILD CONTAINER — /root/reference does not exist on the GPU box, so the committed figures are copied into the line,
    labelled with where they come from (they are not measured by this run)."""
    path = os.path.join(ROOT, "profiles", "r3_reference_functions_cpu.json")
    shape = {"EWMA": "c2_shape", "DBSCAN": "c4_shape"}.get(algo)
    if shape is None or not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    m = d[shape]
    return {"measured": "NOT by this run: committed figures of oracle/ref_baseline.py, %s, %d cores (%s)" % (d["where"], d["cores"], d["cpu"]),
            "what": "reference functions imported from /root/reference, unmodified, on %d rows / %d keys / %d buckets of the synthetic table; "
                    "pandas Stage 0 + series assembly included" % (m["rows"], m["keys"], m["buckets"]),
            "rows_per_s_one_core": m["rows_per_s_one_core_with_pandas_stage0"], "rows_per_s_all_cores": m["rows_per_s_all_cores_with_pandas_stage0"],
            "rows_per_s_one_core_udf_only": m["rows_per_s_one_core_udf_only"], "cores": d["cores"]}


def cpu_sample(algo, rows, keys, cores):
    """(rows, keys, single_rows) of the bounded CPU sample: ~10-30 s of host work, same rows-per-key as the workload."""
    rpk = max(1, rows // keys)
    if algo == "ARIMA":        # ~22 keys/s per core (245 fits of ~125 points each per key)
        ck = min(keys, max(8, 12 * cores))
        return ck * rpk, ck, min(ck, 24) * rpk
    cr = rows if cores >= 64 else min(rows, 30_000_000)     # the full table on a many-core host (a few seconds there)
    return cr, max(1, cr // rpk), min(cr, 20_000_000)


def pmc_traffic(kernel, name="pmc_latest.json"):
    """HBM bytes per launch of `kernel` from the committed PMC summary (separate rocprofv3 --pmc passes), or None.  `job_*`: the sum over
    every kernel of the job (each launches once per job; the table generator is not part of it)."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        once = ("k_synth",)     # the table generator is not part of the job
        jf = sum(v["fetch_bytes"] for a, v in d["kernels"].items() if a not in once)
        jw = sum(v["write_bytes"] for a, v in d["kernels"].items() if a not in once)
        return {"bytes": k["fetch_bytes"] + k["write_bytes"], "fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
                "job_bytes": jf + jw, "job_fetch_bytes": jf, "job_write_bytes": jw, "source": d["source"]}
    except Exception:
        return None


def concurrency_leg(eng, levels, serial_ms):
    """Jobs in flight on ONE engine (tad.h ABI 12: job contexts = a pool of streams + workspaces; controller.go:199-201 runs four
    workers).  (i) c host threads, each submitting C2 EWMA jobs on a table of its own: aggregate rows/s; (ii) C2 EWMA jobs submitted
    while a C3 ARIMA job (~0.27 s of FP64, low-priority stream) runs: their latency, and what they cost the ARIMA job."""
    import statistics
    import threading
    from theia_amd.engine import SYNTH_SEED
    c2 = CONFIGS["c2"]
    n, K, T = c2["rows"], c2["keys"], c2["buckets"]
    nt = max(levels + [2])
    tabs = [eng.synth(0, n, K, T, seed=SYNTH_SEED + 977 * (i + 11)) for i in range(nt)]
    jobs = [eng.prepare("EWMA", t[0], t[1], t[2], K, agg_flow=c2["agg"], out="device") for t in tabs]
    out = {"what": "c host threads submit C2 EWMA jobs (1e8 rows each, a table per thread) to one engine; value = aggregate rows/s over the "
                   "wall time of all threads; serial_ms_per_step = this line's ms_per_step", "serial_ms_per_step": serial_ms, "levels": {}}
    steps = 12
    for c in levels:
        ctxs = set()
        for j in jobs[:c]:          # every context that will be used has its buffers
            j.run().close()
        bar = threading.Barrier(c + 1)

        def work(i):
            bar.wait()
            for _ in range(steps):
                r = jobs[i].run()
                ctxs.add(r.stats["job_context"])
                r.close()
        if c > 1:       # warm the c contexts: c jobs at once
            ths = [threading.Thread(target=work, args=(i,)) for i in range(c)]
            for th in ths:
                th.start()
            bar.wait()
            for th in ths:
                th.join()
            bar = threading.Barrier(c + 1)
            ctxs.clear()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(c)]
        for th in ths:
            th.start()
        bar.wait()
        t0 = time.perf_counter()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        out["levels"][str(c)] = {"value": c * steps * n / dt, "ms_per_job_wall": dt * 1e3 / steps, "ms_per_job_amortised": dt * 1e3 / (steps * c),
                                 "job_contexts_used": sorted(ctxs)}
    base = out["levels"].get("1", {}).get("value")
    if base:
        for c, v in out["levels"].items():
            v["vs_one_in_flight"] = v["value"] / base
    # (ii) a short job next to a long one
    arima = eng.prepare("ARIMA", tabs[0][0], tabs[0][1], tabs[0][2], K, agg_flow=c2["agg"], out="device")
    t0 = time.perf_counter()
    arima.run().close()          # alone (also: allocates the ARIMA workspace)
    t0 = time.perf_counter()
    arima.run().close()
    alone_s = time.perf_counter() - t0
    def beside(gap_s):
        lat, box = [], {}

        def long_job():
            t = time.perf_counter()
            r = arima.run()
            box["s"] = time.perf_counter() - t
            box["ctx"], box["relaunches"] = r.stats["job_context"], r.stats["arima_relaunches"]
            r.close()
        th = threading.Thread(target=long_job)
        th.start()
        time.sleep(0.03)
        while th.is_alive():
            t = time.perf_counter()
            r = jobs[1].run()
            lat.append((time.perf_counter() - t) * 1e3)
            r.close()
            if gap_s:
                time.sleep(gap_s)
        th.join()
        if len(lat) > 1:
            lat = lat[:-1]        # the last one may have outlived the ARIMA job
        q = sorted(lat)
        return {"ewma_jobs": len(lat), "ewma_ms_p50": statistics.median(lat) if lat else None, "ewma_ms_p90": q[int(0.9 * (len(q) - 1))] if q else None,
                "ewma_ms_max": max(lat) if lat else None, "arima_s": box.get("s"), "arima_relaunches": box.get("relaunches"), "arima_job_context": box.get("ctx")}
    out["short_job_beside_long_job"] = {
        "what": "C2 EWMA jobs submitted by one thread while a C3 ARIMA job (same shape, another table) runs on another context's low-priority "
                "stream: every 20 ms (`paced`), and back to back (`saturated`: the fit time-slices with them — 2 ms at most per wait, then ~1 ms "
                "of fit whatever arrives).  While a whole-CU job is in flight the fit kernel suspends its fits at the end of the running optimiser "
                "cycle and its wavefronts retire; the host relaunches it and every wavefront takes its lanes back (tad_stats.arima_relaunches; "
                "results bit-identical).  Without that the EWMA job waited for the fit's whole grid: 212 ms (profiles/r6_a2_bench_default_line.json)",
        "ewma_ms_alone": serial_ms, "arima_s_alone": alone_s, "paced": beside(0.02), "saturated": beside(0.0)}
    return out


def row_orders_leg(eng, dev):
    """The C2 job on one table in the orders a caller can bring its rows in (tools/order_bench.py has the longer list).  Integer sum / max do
    not depend on the order, so every permutation must give the arbitrary order's rows bit for bit (checked); what changes is which queues
    of pass B fill.  Median ms per job over 10 jobs each, device-resident columns, result on the device."""
    import statistics
    import torch
    c2 = CONFIGS["c2"]
    n, K, T = c2["rows"], c2["keys"], c2["buckets"]
    key = torch.empty(n, dtype=torch.int64, device=dev)
    tend = torch.empty(n, dtype=torch.int64, device=dev)
    val = torch.empty(n, dtype=torch.int64, device=dev)
    eng.synth(0, n, K, T, into=(key, tend, val))
    out = {"what": "C2 EWMA job, 1e8 rows, median ms per job of 10 and the warm engine's FIRST job on the table (stage0_attempts 2 = Stage 0 was redone "
                   "with the exact histogram); `identical`: the result rows equal the arbitrary order's bit for bit"}
    fields = ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")

    def time_job(k, t, v, **kw):
        torch.cuda.synchronize()      # (the columns were written on torch's stream, the engine reads them on its own)
        ms, st, first = [], None, None
        for i in range(12):
            t0 = time.perf_counter()
            r = eng.run("EWMA", k, t, v, K, agg_flow=c2["agg"], out="device", **kw)
            st = r.stats
            r.close()
            dt = (time.perf_counter() - t0) * 1e3
            if i == 0:
                first = {"ms": dt, "stage0_attempts": st["stage0_attempts"]}      # every job of a controller is the first on its table
            if i >= 2:
                ms.append(dt)
        return {"ms_per_job": statistics.median(ms), "first_job_on_this_table": first, "pass_b_ms": st["ms_scatter"], "stage0_attempts": st["stage0_attempts"],
                "hist_sampled": st["hist_sampled"]}

    def rows_of(k, t, v):
        torch.cuda.synchronize()
        r = eng.run("EWMA", k, t, v, K, agg_flow=c2["agg"])
        got = {f: r[f].copy() for f in fields}
        r.close()
        return got
    want = rows_of(key, tend, val)      # (also warms the engine: buffers, code objects)
    out["arbitrary"] = time_job(key, tend, val)
    # ids in order of first appearance: the same series under other ids
    first = torch.full((K,), n, dtype=torch.int64, device=dev).scatter_reduce(0, key, torch.arange(n, device=dev), "amin")
    newid = torch.empty(K, dtype=torch.int64, device=dev)
    newid[torch.sort(first).indices] = torch.arange(K, device=dev)
    k2 = newid[key].contiguous()
    timed = time_job(k2, tend, val)     # (timed before the rows are fetched: its first job is the engine's first on this table)
    got = rows_of(k2, tend, val)
    old = torch.sort(newid).indices.cpu().numpy()[got["key_id"].astype(np.int64)]
    perm = old.argsort(kind="stable")
    same = got["key_id"].size == want["key_id"].size and (old[perm] == want["key_id"].astype(np.int64)).all() and \
        all((got[f][perm] == want[f]).all() for f in fields[1:])
    out["ids_by_first_appearance"] = dict(timed, identical=bool(same))
    del k2, first, newid

    def rows_sorted_by(name, column):
        o = torch.sort(column, stable=True).indices
        k, t, v = key[o].contiguous(), tend[o].contiguous(), val[o].contiguous()
        del o
        timed = time_job(k, t, v)
        got = rows_of(k, t, v)
        same = all(got[f].shape == want[f].shape and (got[f] == want[f]).all() for f in fields)
        out[name] = dict(timed, identical=bool(same))
    rows_sorted_by("by_time", tend)
    # keys that live for a tenth of the table, rows in time order: every workgroup of pass B sees a narrow range of ids
    i = torch.arange(n, device=dev)
    W = K // 10
    k_live = ((i.double() * ((K - W) / n)).long() + (key * 2654435761 % W)) % K
    t_live = tend.min() + 60 * ((i.double() * (T / n)).long())
    del i
    out["keys_alive_for_a_tenth_in_time_order"] = time_job(k_live, t_live, val)
    del k_live, t_live
    t_lo, t_hi = int(tend.min()), int(tend.max())
    out["end_time_keeps_80_percent"] = time_job(key, tend, val, end_time=t_lo + (t_hi - t_lo) * 4 // 5)
    rows_sorted_by("by_key", key)      # last: a sorted table leaves "this shape needs the exact histogram" behind in the job context
    del key, tend, val
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """Run this same command line as `n` ranks of one node (python -m torch.distributed.run, one process per GPU) and return
    the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS), help="BASELINE.json config; --algo/--rows/--keys/--buckets/--agg override it")
    ap.add_argument("--algo", default=None, choices=["EWMA", "DBSCAN", "ARIMA"])
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU")
    ap.add_argument("--keys", type=int, default=None, help="keys per GPU")
    ap.add_argument("--buckets", type=int, default=None)
    ap.add_argument("--agg", default=None)
    ap.add_argument("--hint-lattice", action="store_true", help="pass the time lattice instead of deriving it")
    ap.add_argument("--ingest", default="keys", choices=["keys", "rows"],
                    help="keys: every rank holds the rows of its own keys (no data-path collective, the default); "
                         "rows: every rank holds an arbitrary slice of the rows -> pre-aggregate (tad_aggregate), one "
                         "all-to-all(v) of the partial points to the key owners, detect on the owners")
    ap.add_argument("--host-input", action="store_true",
                    help="hand the columns over as pinned HOST buffers (tad_columns.memory = TAD_MEM_HOST): the PCIe-inclusive "
                         "rate DESIGN.md quotes; never the headline value")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with ONE rank: create the process group anyway and run the job's all-gather / all-to-all(v) through it "
                         "(backend nccl = RCCL on device tensors) — the single-GPU check of the N>1 RCCL path")
    ap.add_argument("--plan", default="", help="tad_plan overrides for A/B runs, e.g. histogram=exact,partition_pass=sort (default: the engine decides)")
    ap.add_argument("--tables", type=int, default=4, help="synthetic tables of the workload's shape the timed steps go round robin over "
                                                           "(different seeds, different device buffers); 1 = the same columns every step")
    ap.add_argument("--c5-shape", default="", help="N > 1: run the other_configs.c5 leg (BASELINE configs[4], key- and row-sharded) on a table of "
                                                     "ROWS,KEYS,BUCKETS in total instead of 1e9,1e6,250, whatever the headline is (reduced-size tests)")
    ap.add_argument("--no-row-orders", action="store_true", help="N = 1 default line: skip the row_orders leg")
    ap.add_argument("--concurrency", default="1,2,4", help="N = 1 default line: host threads submitting C2 jobs to the one engine (job contexts); '' = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the CPU sample (0 = chosen from the host's core count)")
    ap.add_argument("--dump-rows", default="", help="after the timed region run the step once more and save every rank's anomaly rows "
                                                     "(GLOBAL key ids) to <path>.rank<r>.npz — used by the N-rank == 1-rank parity test")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks here (one process per GPU under
        # torch.distributed.run, RCCL rendezvous on 127.0.0.1) and hand the same command line to each of them; the ranks inherit
        # stdout, rank 0 alone prints the JSON line, last.  Under an external torchrun WORLD_SIZE is set and this is skipped.
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from theia_amd import TadEngine
    from theia_amd import distributed as td
    from theia_amd.engine import SYNTH_SEED

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # backend "nccl" is RCCL on ROCm.  TAD_BENCH_BACKEND=gloo exists to exercise the N>1 code path on a box with
    # fewer GPUs than ranks (ranks then share devices and the collectives run on host tensors).
    backend = os.environ.get("TAD_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > max(1, torch.cuda.device_count()):
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s): RCCL needs one device per rank (TAD_BENCH_BACKEND=gloo lets ranks "
                         "share a device for testing the N>1 code path)" % (world, torch.cuda.device_count()))
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    grouped = world > 1 or args.force_collectives
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if (grouped and backend == "nccl") else None
    ranks_seen = 1
    if grouped:     # evidence that the collectives of this line really crossed `world` processes: every rank contributes its rank id
        ids = [torch.zeros(1, dtype=torch.int64, device=coll_dev if coll_dev is not None else "cpu") for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=coll_dev if coll_dev is not None else "cpu"))
        ranks_seen = len({int(x.item()) for x in ids})

    cfg = dict(CONFIGS[args.config])
    strong = args.config == "c5"
    if strong:      # the fixed 1e9-row / 1e6-key table split over the ranks
        cfg["rows"], cfg["keys"] = cfg["rows"] // world, cfg["keys"] // world
    if args.algo:
        cfg["algos"] = (args.algo,)
    for f in ("rows", "keys", "buckets", "agg"):
        if getattr(args, f) is not None:
            cfg[f] = getattr(args, f)
    headline_is_c2 = (args.config == "c2" and not args.algo and all(getattr(args, f) is None for f in ("rows", "keys", "buckets", "agg")))

    plan = {}
    for item in filter(None, args.plan.split(",")):
        name, _, val = item.partition("=")
        plan[name.strip()] = int(val) if val.strip().lstrip("-").isdigit() else val.strip()
    eng = TadEngine(device=dev.index, plan=plan)
    reducer = td.JobReducer(device=coll_dev)

    def make_table(n, K, T, ingest, index=0):
        key = torch.empty(n, dtype=torch.int64, device=dev)
        tend = torch.empty(n, dtype=torch.int64, device=dev)
        val = torch.empty(n, dtype=torch.int64, device=dev)
        # ingest=keys: local key ids 0..K-1 of this rank's shard (global key = local * world + rank);
        # ingest=rows: an arbitrary slice of the rows, global key ids over all K * world keys
        # index > 0: another table of the same shape (another seed, other buffers) for the fresh-columns loop
        eng.synth(rank * n, n, K * (world if ingest == "rows" else 1), T, seed=SYNTH_SEED + 977 * index, into=(key, tend, val))
        return key, tend, val

    def run_config(algos, n, K, T, agg, steps, warmup, ingest="keys", host_input=False, hint=False, tables=1, same_columns_steps=0):
        """W warm-up steps, then exactly `steps` timed steps between synchronisation points; one step = one job per algo.  With
        tables > 1 step i runs on table i mod tables (different rows in different device buffers: no step meets the columns of the
        step before it); same_columns_steps > 0 adds a second timed loop on table 0 alone."""
        tabs = [make_table(n, K, T, ingest, i) for i in range(tables)]
        lattice = (1660202814, 60, T) if hint else None
        if host_input:
            htabs = [tuple(x.cpu().pin_memory() for x in tab) for tab in tabs]
        pending = [None]
        rccl = {"allgather_us": [], "alltoall_ms": [], "alltoall_bytes": 0, "exchanges": 0}

        dump = {}

        prepared = {}   # the device-resident job of a step, its two C structs built once (TadEngine.prepare): a step = the bare tad_run call

        def one_job(algo, ti, k_, t_, v_):
            if host_input:
                return eng.run(algo, *htabs[ti], K, agg_flow=agg, lattice=lattice, out="host")
            if dump.get("on") or ingest == "rows":
                return eng.run(algo, k_, t_, v_, K, agg_flow=agg, lattice=lattice, out="host" if dump.get("on") else "device")
            if (algo, ti) not in prepared:
                prepared[(algo, ti)] = eng.prepare(algo, k_, t_, v_, K, agg_flow=agg, lattice=lattice, out="device")
            return prepared[(algo, ti)].run()

        def step(ti=0):
            stats, glob = [], None
            key, tend, val = tabs[ti]
            if ingest == "rows":
                # Stage 0 on the local slice -> partial points; all-to-all(v) to the owners (RCCL over xGMI); the owners run the
                # job(s) on the partials (re-aggregating sums of sums is bit-exact)
                pts = eng.aggregate(key, tend, val, K * world, agg_flow=agg, lattice=lattice, out="device")
                ptr = pts.device_pointers()
                cols = [torch.as_tensor(td.DeviceColumn(ptr[f], pts.n_points), device=dev) for f in ("key_id", "flow_end_s", "value")]
                # bucketed by owner on the GPU (tad_shard_rows), shipped with one all-to-all(v) per column
                tx = time.perf_counter()
                lk, lt, lv = td.exchange_rows_device(eng, cols[0], cols[1], cols[2], world, rank,
                                                     host_collective=(coll_dev is None and grouped))
                torch.cuda.synchronize()
                rccl["alltoall_ms"].append((time.perf_counter() - tx) * 1e3)
                rccl["alltoall_bytes"] = 24 * int(pts.n_points)      # sent by this rank per exchange (three 8-byte columns per partial point)
                rccl["exchanges"] += 1
                pts.close()
            else:
                lk, lt, lv = key, tend, val
            for algo in algos:
                res = one_job(algo, ti, lk, lt, lv)
                st = res.stats
                # RCCL over xGMI: one 9-double all-gather (counters + moments) per job, started now and collected after the
                # NEXT job has been issued, so its latency hides behind that job; the last one is collected inside the timed region
                if grouped:
                    nxt = reducer.start(st)
                    if pending[0] is not None:
                        tg = time.perf_counter()
                        glob = pending[0].result()
                        rccl["allgather_us"].append((time.perf_counter() - tg) * 1e6)
                    pending[0] = nxt
                if dump.get("on"):
                    h = res.to_host()
                    dump[algo] = {f: (h[f] * np.uint64(world) + np.uint64(rank) if f == "key_id" else h[f]).copy()
                                  for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
                res.close()
                stats.append(st)
            return stats, glob

        def drain():
            g = None
            if pending[0] is not None:
                tg = time.perf_counter()
                g = pending[0].result()
                rccl["allgather_us"].append((time.perf_counter() - tg) * 1e6)
            pending[0] = None
            return g

        def timed_loop(count, table_of):
            acc = [{"ms_meta": 0.0, "ms_stage0": 0.0, "ms_scatter": 0.0, "ms_detect": 0.0, "ms_total": 0.0} for _ in algos]
            glob, stats = None, None
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(count):
                stats, g2 = step(table_of(i))
                glob = g2 or glob
                for a, st in zip(acc, stats):
                    for f in a:
                        a[f] += st[f]
            if grouped:
                glob = drain() or glob          # the last job's reduction completes inside the timed region
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            dt = time.perf_counter() - t0
            dt_rank = dt
            if grouped:
                tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else "cpu")
                lo = tt.clone()
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dt, dt_rank = float(tt.item()), float(lo.item())
            return dt, dt_rank, acc, stats, glob

        glob = None
        cold = None
        for w in range(warmup):
            if w == 0:      # the first job of this shape in this process: allocations, three host synchronisations
                torch.cuda.synchronize()
                tc = time.perf_counter()
            stats, glob = step(w % tables)
            if w == 0:
                torch.cuda.synchronize()
                cold = {"ms_first_step": (time.perf_counter() - tc) * 1e3, "host_syncs_per_job": [st.get("host_syncs") for st in stats],
                        "stage0_attempts": [st.get("stage0_attempts") for st in stats],
                        "what": "the first step of this shape in the process (untimed warm-up step 1): the job context's buffer allocations "
                                "included (the library's code objects are loaded by tad_engine_create)"}
        for ti in range(min(warmup, 1) * tables):      # every table's prepared job exists before the timed region (a struct build, no GPU work)
            if not host_input and ingest == "keys" and all((a, ti) in prepared for a in algos):
                continue
            step(ti)
        if grouped:
            glob = drain() or glob
        rccl["allgather_us"].clear()
        rccl["alltoall_ms"].clear()
        # (the LAST timed step lands on table 0, the default-seed table the CPU baseline and the parity tests use: `result` is comparable)
        dt, dt_min_rank, acc, stats, g2 = timed_loop(steps, lambda i: (i + 1 - steps) % tables)
        glob = g2 or glob
        same = None
        if same_columns_steps:
            sdt, _, _, _, _ = timed_loop(same_columns_steps, lambda i: 0)
            same = {"ms_per_step": sdt * 1e3 / same_columns_steps, "steps": same_columns_steps,
                    "what": "the same loop on ONE table (every step meets the previous step's columns): the rounds-1-5 headline loop"}
        if not grouped:
            glob = td.JobReducer().reduce(stats[-1])
        if args.dump_rows:
            dump["on"] = True
            step(0)
            if grouped:
                drain()
            np.savez(args.dump_rows + ".rank%d.npz" % rank, **{"%s_%s" % (a, f): v for a in algos for f, v in dump[a].items()})
        del tabs
        return dict(dt=dt, dt_min_rank=dt_min_rank, stats=stats, acc=acc, glob=glob, steps=steps, warmup=warmup, n=n, K=K, T=T, agg=agg, algos=algos, cold=cold,
                    tables=tables, same=same, rccl=rccl)

    def describe(r, host_input=False, ingest="keys", hint=False):
        """the JSON fields of one measured config (rank 0)"""
        n, K, T, steps, algos = r["n"], r["K"], r["T"], r["steps"], r["algos"]
        ms_step = r["dt"] * 1e3 / steps
        st0, a0 = r["stats"][0], r["acc"][0]            # Stage 0 of the first job of the step (every job of a step re-runs it)
        scatter_ms = a0["ms_scatter"] / steps
        achieved = BYTES_PER_ROW * n / (scatter_ms * 1e-3) / 1e9
        dev_ms = sum(a["ms_total"] for a in r["acc"]) / steps
        A = sum(st["n_anomalies"] for st in r["stats"])
        out = {
            "value": world * n * steps / r["dt"], "unit": "flow-records/s", "steps": steps, "warmup": r["warmup"], "ms_per_step": ms_step,
            "config": {"workload": "%s detector%s, %d rows / %d flow keys / %d time buckets per GPU, %s(throughput) (agg_flow=%s), "
                                   "deterministic synthetic flow table (SURVEY.md 8d), %s"
                                   % (" then ".join(algos), "s (one step = both jobs)" if len(algos) > 1 else "", n, K, T,
                                      "sum" if r["agg"] else "max", r["agg"] or "None",
                                      "inputs in pinned HOST memory, results copied back (PCIe-inclusive)" if host_input
                                      else "inputs and outputs resident in HBM"),
                       "algo": "+".join(algos), "rows_per_gpu": n, "keys_per_gpu": K, "buckets": T,
                       "lattice": "hinted" if hint else "derived by the engine (extra pass over flow_end_s)",
                       "parallelism": ("key-sharded x%d, no data-path collective; one 9-double all-gather per job (counters + moments)" % world)
                       if ingest == "keys" else
                       ("row-sharded x%d: tad_aggregate on the local slice, one all-to-all(v) of partial points, job on the owners; "
                        "one 9-double all-gather per job" % world)},
            "roofline": {"bound": "hbm", "kernel": STAGE0_KERNEL.get(st0["stage0_path"], "k_scatter (Stage-0 v1, direct atomics)"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": BYTES_PER_ROW * n, "avg_kernel_ms": scatter_ms,
                         # SURVEY.md 8d's definition on the run: (24 N + 40 A) per job over the WALL time of a step
                         "achieved_whole_run": (len(algos) * BYTES_PER_ROW * n + BYTES_PER_ANOMALY * A) / (ms_step * 1e-3) / 1e9,
                         "frac_whole_run": (len(algos) * BYTES_PER_ROW * n + BYTES_PER_ANOMALY * A) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "pipeline": {"ms_meta": a0["ms_meta"] / steps, "ms_stage0_clear_plus_scatter": a0["ms_stage0"] / steps,
                         "ms_detect_and_emit": sum(a["ms_detect"] for a in r["acc"]) / steps, "ms_device_total": dev_ms,
                         "host_syncs_per_job": [st.get("host_syncs") for st in r["stats"]],
                         "hbm_frac_whole_job": (len(algos) * BYTES_PER_ROW * n + BYTES_PER_ANOMALY * A) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        out["config"]["tables"] = r.get("tables", 1)
        out["config"]["columns"] = ("step i runs on table i mod %d: tables of this shape with different seeds in different device buffers — no step "
                                    "meets the columns of the step before it" % r["tables"]) if r.get("tables", 1) > 1 else "one table, every step"
        if r.get("cold"):
            out["cold"] = r["cold"]
        if r.get("same"):
            out["same_columns"] = dict(r["same"], ratio_to_ms_per_step=r["same"]["ms_per_step"] / ms_step)
        if grouped:
            import statistics
            rc = r["rccl"]
            out["rccl"] = {"backend": "%s (%s)" % (backend, "RCCL, device tensors" if backend == "nccl" else "host tensors: ranks share GPUs"),
                           "world": world, "ranks_seen": ranks_seen,
                           "allgather_us_p50": statistics.median(rc["allgather_us"]) if rc["allgather_us"] else None,
                           "allgather_what": "host wait for the 9-double all-gather of a job (counters + moments), collected after the NEXT job "
                                             "was issued; %d collected in the timed region" % len(rc["allgather_us"]),
                           "alltoall_bytes": rc["alltoall_bytes"] if ingest == "rows" else 0,
                           "alltoall_ms": statistics.median(rc["alltoall_ms"]) if rc["alltoall_ms"] else None,
                           "alltoall_what": "tad_shard_rows + one all-to-all(v) per column of the partial points (24 B each), wall time on rank 0"
                                            if ingest == "rows" else "none: rows arrive key-sharded",
                           "per_rank_ms_per_step": {"min": r["dt_min_rank"] * 1e3 / steps, "max": ms_step,
                                                    "what": "timed region of the fastest and of the slowest rank (key-size imbalance)"}}
        g = r["glob"]
        out["result"] = {"anomalies": g["n_anomalies"], "keys": g["n_keys"], "points": g["n_points"], "rows_used": g["rows_used"],
                         "global_mean": g["global_mean"], "global_sigma": g["global_sigma"]}
        for st, a, algo in zip(r["stats"], r["acc"], algos):
            if algo == "ARIMA":
                sec = a["ms_detect"] / steps * 1e-3
                flops = FLOP_PER_KALMAN_STEP * st["kalman_steps"]
                eq60 = FLOP_PER_KALMAN_STEP_3STATE * st["kalman_steps"] / sec / 1e12
                out["arima"] = {"fits_per_launch": st["arima_fits"], "kalman_steps_per_launch": st["kalman_steps"],
                                "fits_per_s": st["arima_fits"] / sec, "nan_fits": st.get("arima_nan_fits", 0),
                                "bound": "fp64 vector ALU: instruction issue + dependency latency",
                                "achieved_tflops": flops / sec / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                "frac": flops / sec / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "ms_detect": a["ms_detect"] / steps,
                                "tflops_60flop_equivalent": eq60, "frac_60flop_equivalent": eq60 / FP64_VECTOR_PEAK_TFLOPS,
                                "flop_model": "16 flop per time-step of the collapsed recursion (executed) x the engine's kalman_steps counter; "
                                              "60-flop-equivalent = SURVEY.md 8d's three-state model on the same counter (rounds 1-2 reported that)"}
                if len(algos) == 1:     # the detector, not Stage 0, is this config's dominant kernel
                    out["roofline"] = {"bound": "fp64-vector", "kernel": "k_arima_fit (per-lane L-BFGS-B over the ARIMA(1,1,1) likelihood, four recursions per lane)",
                                       "achieved": flops / sec / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": flops / sec / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                                       "algorithmic_flops_per_launch": flops, "avg_kernel_ms": a["ms_detect"] / steps,
                                       "frac_60flop_equivalent": eq60 / FP64_VECTOR_PEAK_TFLOPS,
                                       "hbm_frac_of_24B_per_row": BYTES_PER_ROW * n / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "frac_whole_run": flops / (ms_step * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS}
        return out

    # the fresh-columns loop wherever four tables of the shape are cheap (<= 1e8 rows: 9.6 GB); C5 keeps one table
    multi = args.tables > 1 and not strong and not args.host_input and args.ingest == "keys" and cfg["rows"] <= 100_000_000 and not args.dump_rows
    head = run_config(cfg["algos"], cfg["rows"], cfg["keys"], cfg["buckets"], cfg["agg"], args.steps, args.warmup,
                      ingest=args.ingest, host_input=args.host_input, hint=args.hint_lattice, tables=args.tables if multi else 1,
                      same_columns_steps=max(1, args.steps // 2) if multi else 0)
    out = None
    if rank == 0:
        d = describe(head, args.host_input, args.ingest, args.hint_lattice)
        out = {"metric": "flow-records/sec", "value": d["value"], "unit": d["unit"], "n_gpus": world, "steps": d["steps"], "warmup": d["warmup"],
               "ms_per_step": d["ms_per_step"], "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
               "dtype": "u64 (aggregates) + f64 (detectors)", "data": "synthetic", "config": d["config"], "roofline": d["roofline"],
               "pipeline": d["pipeline"], "result": d["result"]}
        for f in ("cold", "same_columns", "rccl"):
            if f in d:
                out[f] = d[f]
        out["config"]["baseline_config"] = args.config
        if "arima" in d:
            out["arima"] = d["arima"]
        if headline_is_c2:   # the PMC passes were taken on this workload
            out["roofline"]["traffic"] = pmc_traffic({2: "k_partition", 3: "k_partition_wc"}.get(head["stats"][0]["stage0_path"], "k_scatter"))
    cores = os.cpu_count() or 1
    if world == 1 and headline_is_c2 and not args.no_other_configs and not args.host_input and args.ingest == "keys":
        # BASELINE.json configs[3] and configs[2], measured in the same run (their tables replace the C2 table in HBM)
        others = {}
        # C5 at N = 1 is the strong-scaling base of BASELINE.json configs[4] (the whole 1e9-row / 1e6-key table on one GPU)
        for name, steps, warmup in (("c4", 5, 2), ("c3", 1, 1), ("c5", 1, 1)):
            c = CONFIGS[name]
            try:
                r = run_config(c["algos"], c["rows"], c["keys"], c["buckets"], c["agg"], steps, warmup)
            except Exception as exc:      # the headline line must still be printed
                others[name] = {"baseline_config": name, "error": repr(exc)[:300]}
                torch.cuda.empty_cache()
                continue
            d = describe(r)
            d["baseline_config"] = name
            if name == "c4":     # PMC passes of `bench.py --config c4` (profiles/README.md)
                d["roofline"]["traffic"] = pmc_traffic({2: "k_partition", 3: "k_partition_wc"}.get(r["stats"][0]["stage0_path"], "k_scatter"),
                                                       "pmc_latest_c4.json")
            if name == "c5":     # PMC passes of `bench.py --config c5 --algo EWMA` (the Stage 0 both jobs of the step run)
                d["roofline"]["traffic"] = pmc_traffic({2: "k_partition", 3: "k_partition_wc"}.get(r["stats"][0]["stage0_path"], "k_scatter"),
                                                       "pmc_latest_c5.json")
                d["scaling"] = "strong (this is the N = 1 base: `bench.py --config c5 --gpus N` splits the same table over N ranks)"
            elif not args.no_cpu_baseline:
                cr, ck, sr = cpu_sample(c["algos"][0], c["rows"], c["keys"], cores)
                try:
                    d["cpu_baseline"] = cpu_baseline(c["algos"][0], cr, ck, c["buckets"], c["agg"], sr)
                    d["cpu_baseline"]["gpu_over_cpu"] = d["value"] / d["cpu_baseline"]["value"]
                except Exception as exc:      # the headline line must still be printed
                    d["cpu_baseline"] = {"error": repr(exc)[:200]}
            others[name] = d
        out["other_configs"] = others
    if world > 1 and (headline_is_c2 or args.c5_shape) and not args.no_other_configs and not args.host_input and args.ingest == "keys":
        # BASELINE.json configs[4], the only config it names for 8 GPUs: the 1e9-row / 1e6-key table split over the N ranks (strong
        # scaling), one step = EWMA then ARIMA on the rank's shard — key-sharded (no data-path collective) and row-sharded
        # (tad_aggregate + one all-to-all(v) of the partial points to the key owners: the shuffle of anomaly_detection.py:664-684)
        c = dict(CONFIGS["c5"])
        if args.c5_shape:
            c["rows"], c["keys"], c["buckets"] = (int(x) for x in args.c5_shape.split(","))
        c5 = {}
        for ing in ("keys", "rows"):
            try:
                r = run_config(c["algos"], c["rows"] // world, c["keys"] // world, c["buckets"], c["agg"], 1, 1, ingest=ing)
                if rank == 0:
                    d = describe(r, ingest=ing)
                    d["baseline_config"] = "c5"
                    d["scaling"] = "strong: %d rows / %d keys in total, 1/%d of them per rank" % (c["rows"], c["keys"], world)
                    c5[ing] = d
            except Exception as exc:      # the headline line must still be printed
                c5[ing] = {"baseline_config": "c5", "error": repr(exc)[:300]}
                torch.cuda.empty_cache()
        if rank == 0:
            out["other_configs"] = {"c5": c5}
    if world == 1 and headline_is_c2 and args.concurrency and not args.no_other_configs and not args.host_input and args.ingest == "keys":
        try:
            out["concurrency"] = concurrency_leg(eng, [int(x) for x in args.concurrency.split(",") if x], out["ms_per_step"])
        except Exception as exc:
            out["concurrency"] = {"error": repr(exc)[:300]}
    if world == 1 and headline_is_c2 and not args.no_other_configs and not args.host_input and args.ingest == "keys" and not args.no_row_orders:
        try:
            eng_o = TadEngine(device=dev.index, plan=plan)     # (an engine of its own: its contexts learn from the sorted tables)
            out["row_orders"] = row_orders_leg(eng_o, dev)
            eng_o.close()
        except Exception as exc:
            out["row_orders"] = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            algo = cfg["algos"][-1] if len(cfg["algos"]) > 1 else cfg["algos"][0]
            cr, ck, sr = cpu_sample(algo, cfg["rows"], cfg["keys"], cores)
            if args.cpu_rows:
                cr, ck = args.cpu_rows, max(1, int(cfg["keys"] * args.cpu_rows / cfg["rows"]))
                sr = min(sr, cr)
            out["cpu_baseline"] = cpu_baseline(algo, cr, ck, cfg["buckets"], cfg["agg"], sr)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    eng.close()
    if grouped:
        dist.destroy_process_group()
    if rank == 0:      # the ONE JSON line, last on stdout (after the process group is gone: RCCL may print on teardown)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
