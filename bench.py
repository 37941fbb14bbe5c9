#!/usr/bin/env python3
"""bench.py — flow-records/s of the Throughput Anomaly Detection hot path on MI355X.

One "step" = one full job (tad_run through the C ABI) over one synthetic batch that is already
resident in HBM: Stage-0 GROUP BY (key, flowEndSeconds) -> per-key stddev_samp -> detector ->
compaction of the anomalous points into device-resident result columns.

Workload at N=1 = BASELINE.json configs[1]: EWMA on 1e8 rows / 1e5 flow keys / 250 time buckets,
sum(throughput) (mode svc), the deterministic synthetic table of SURVEY.md §8d.
N>1 (torchrun, one rank per GPU): weak scaling — every rank owns the key shard `key mod N == rank`
(1e8 rows / 1e5 keys per rank, pre-sharded by key as SURVEY.md §8e allows), no data-path collective;
per step ONE RCCL all-gather of 9 doubles per rank: the counters [anomalies, keys, points, rows, ...] (the global
`count() == 0` sentinel decision, anomaly_detection.py:395) and the (n, mean, M2) moments (global sigma).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (Stage-0 pass B, k_partition):
achieved = 24 B/row x rows per launch / the kernel's average duration measured with HIP events on the
engine's stream (tad_stats.ms_scatter); `traffic` = that kernel's HBM bytes per launch from the committed
rocprofv3 PMC passes (profiles/pmc_latest.json: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE;
separate --pmc runs of this same command).  `cpu_baseline` = the oracle (numpy port of the reference
job) timed on this box's host cores on a bounded sample.  The ARIMA line adds `arima`: fits/s and the
FP64 flop rate from the engine's Kalman-step counter (60 flop per 3-state predict+update).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
FLOP_PER_KALMAN_STEP = 60
BYTES_PER_ROW = 24      # key_id u64 + flow_end_s i64 + value u64, read once
BYTES_PER_ANOMALY = 40  # key_id, flow_end_s, throughput, algo_calc, stddev


def cpu_baseline(algo, rows, keys, buckets, agg):
    """The oracle (numpy restatement of the reference job) on a bounded sample of the same workload, in its own process
    (oracle/cpu_bench.py): one process, and key-sharded over all host cores the way Spark local[*] runs the per-key
    UDFs.  `value` is the faster of the two, `cores` the processes it used."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), "--algo", algo, "--rows", str(rows), "--keys", str(keys),
           "--buckets", str(buckets), "--agg", agg]
    if algo == "ARIMA":
        cmd += ["--skip-single"] if (os.cpu_count() or 1) > 1 else []
    r = json.loads(subprocess.run(cmd, check=True, capture_output=True, timeout=900, text=True).stdout.strip().splitlines()[-1])
    single = rows / r["single_s"] if "single_s" in r else None
    multi = rows / r["multi_s"] if "multi_s" in r else None
    use_multi = multi is not None and (single is None or multi > single)
    out = {"value": multi if use_multi else single, "unit": "flow-records/s", "cores": r["procs"] if use_multi else 1, "kind": "port",
           "sample": "%s, %d rows / %d keys / %d buckets of the same synthetic table (same rows-per-key as the GPU workload), numpy "
                     "oracle; %s" % (algo, rows, keys, buckets,
                                     "key-sharded over %d processes (rows with key mod P == w per worker, selection timed), %.1f s"
                                     % (r["procs"], r["multi_s"]) if use_multi else "single process, %.1f s" % r["single_s"]),
           "host_cores": r["host_cores"], "anomalies": r.get("multi_anomalies", r.get("single_anomalies"))}
    # the reference job evaluates its lineage twice (the `.count()` action of anomaly_detection.py:395, then the JDBC
    # write, :713-726): `value` is the de-duplicated (1x) rate, this is the as-written one
    out["as_written_2x_value"] = out["value"] / 2.0
    if single is not None:
        out["single_core_value"] = single
    if multi is not None:
        out["all_cores_value"] = multi
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (separate rocprofv3 --pmc passes), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        return {"bytes": k["fetch_bytes"] + k["write_bytes"], "fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
                "source": d["source"]}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--algo", default="EWMA", choices=["EWMA", "DBSCAN", "ARIMA"])
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--keys", type=int, default=100_000)
    ap.add_argument("--buckets", type=int, default=250)
    ap.add_argument("--agg", default="svc")
    ap.add_argument("--hint-lattice", action="store_true", help="pass the time lattice instead of deriving it")
    ap.add_argument("--ingest", default="keys", choices=["keys", "rows"],
                    help="keys: every rank holds the rows of its own keys (no data-path collective, the default); "
                         "rows: every rank holds an arbitrary slice of the rows -> pre-aggregate (tad_aggregate), one "
                         "all-to-all(v) of the partial points to the key owners, detect on the owners")
    ap.add_argument("--host-input", action="store_true",
                    help="hand the columns over as pinned HOST buffers (tad_columns.memory = TAD_MEM_HOST): the PCIe-inclusive "
                         "rate DESIGN.md quotes; never the headline value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=30_000_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from theia_amd import TadEngine
    from theia_amd import distributed as td

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # backend "nccl" is RCCL on ROCm.  TAD_BENCH_BACKEND=gloo exists to exercise the N>1 code path on a box with
    # fewer GPUs than ranks (ranks then share devices and the collectives run on host tensors).
    backend = os.environ.get("TAD_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if (world > 1 and backend == "nccl") else None

    eng = TadEngine(device=dev.index)
    n, K, T = args.rows, args.keys, args.buckets
    # rank r's shard: its own 1e8 rows of the table, local key ids 0..K-1 (global key = local * world + rank)
    key = torch.empty(n, dtype=torch.int64, device=dev)
    tend = torch.empty(n, dtype=torch.int64, device=dev)
    val = torch.empty(n, dtype=torch.int64, device=dev)
    # ingest=keys: local key ids 0..K-1 of this rank's shard; ingest=rows: global key ids over all K * world keys
    eng.synth(rank * n, n, K * (world if args.ingest == "rows" else 1), T, into=(key, tend, val))
    lattice = (1660202814, 60, T) if args.hint_lattice else None

    reducer = td.JobReducer(device=coll_dev)
    if args.host_input:
        hkey, htend, hval = (x.cpu().pin_memory() for x in (key, tend, val))

    def step():
        if args.ingest == "rows":
            # Stage 0 on the local slice -> partial points; all-to-all(v) to the owners (RCCL over xGMI); the owners run the
            # job on the partials (re-aggregating sums of sums is bit-exact)
            pts = eng.aggregate(key, tend, val, K * world, agg_flow=args.agg, lattice=lattice, out="device")
            ptr = pts.device_pointers()
            cols = [torch.as_tensor(td.DeviceColumn(ptr[f], pts.n_points), device=dev) for f in ("key_id", "flow_end_s", "value")]
            if coll_dev is None and world > 1:
                cols = [c.cpu() for c in cols]                  # gloo test mode: host tensors
            lk, lt, lv = td.exchange_points_torch(cols[0], cols[1], cols[2], world, rank)
            pts.close()
            lk, lt, lv = (c.to(dev) for c in (lk, lt, lv))
            res = eng.run(args.algo, lk, lt, lv, K, agg_flow=args.agg, lattice=lattice, out="device")
        elif args.host_input:
            res = eng.run(args.algo, hkey, htend, hval, K, agg_flow=args.agg, lattice=lattice, out="host")
        else:
            res = eng.run(args.algo, key, tend, val, K, agg_flow=args.agg, lattice=lattice, out="device")
        st = res.stats
        # RCCL over xGMI: one 9-double all-gather (counters + moments) per job, started now and collected after the NEXT
        # job has been issued, so its latency hides behind that job; the last one is collected inside the timed region
        glob = None
        if world > 1:
            nxt = reducer.start(st)
            if pending[0] is not None:
                glob = pending[0].result()
            pending[0] = nxt
        res.close()
        return st, glob

    pending = [None]

    def drain():
        g = pending[0].result() if pending[0] is not None else None
        pending[0] = None
        return g

    for _ in range(args.warmup):
        st, glob = step()
    if world > 1:
        glob = drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    acc = {"ms_meta": 0.0, "ms_stage0": 0.0, "ms_scatter": 0.0, "ms_detect": 0.0, "ms_total": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, glob = step()
        for f in acc:
            acc[f] += st[f]
    if world > 1:
        glob = drain()          # the last job's reduction completes inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    else:
        glob = td.JobReducer().reduce(st)
    g_counts = [glob["n_anomalies"], glob["n_keys"], glob["n_points"], glob["rows_used"]]

    if rank == 0:
        steps = args.steps
        ms_step = dt * 1e3 / steps
        A = st["n_anomalies"]
        scatter_ms = acc["ms_scatter"] / steps
        achieved = BYTES_PER_ROW * n / (scatter_ms * 1e-3) / 1e9
        out = {
            "metric": "flow-records/sec", "value": world * n * steps / dt, "unit": "flow-records/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (aggregates) + f64 (detectors)", "data": "synthetic",
            "config": {"workload": "%s detector, %d rows / %d flow keys / %d time buckets per GPU, sum(throughput) (agg_flow=%s), "
                                   "deterministic synthetic flow table (SURVEY.md 8d), %s"
                                   % (args.algo, n, K, T, args.agg, "inputs in pinned HOST memory, results copied back (PCIe-inclusive)"
                                      if args.host_input else "inputs and outputs resident in HBM"),
                       "algo": args.algo, "rows_per_gpu": n, "keys_per_gpu": K, "buckets": T,
                       "lattice": "hinted" if args.hint_lattice else "derived by the engine (extra pass over flow_end_s)",
                       "parallelism": ("key-sharded x%d, no data-path collective; one 9-double all-gather per job (counters + moments)" % world)
                       if args.ingest == "keys" else
                       ("row-sharded x%d: tad_aggregate on the local slice, one all-to-all(v) of partial points, job on the owners; "
                        "one 9-double all-gather per job" % world)},
            "roofline": {"bound": "hbm", "kernel": {2: "k_partition (Stage-0 v2, row partition pass, sort-by-tile)", 3: "k_partition_wc (Stage-0 v2, row partition pass, write-combining)"}.get(st["stage0_path"], "k_scatter (Stage-0 v1, direct atomics)"), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": BYTES_PER_ROW * n, "avg_kernel_ms": scatter_ms},
            "pipeline": {"ms_meta": acc["ms_meta"] / steps, "ms_stage0_clear_plus_scatter": acc["ms_stage0"] / steps,
                         "ms_detect_and_emit": acc["ms_detect"] / steps, "ms_device_total": acc["ms_total"] / steps,
                         "hbm_frac_whole_job": (BYTES_PER_ROW * n + BYTES_PER_ANOMALY * A) / (acc["ms_total"] / steps * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "result": {"anomalies": g_counts[0], "keys": g_counts[1], "points": g_counts[2], "rows_used": g_counts[3],
                       "global_mean": glob["global_mean"], "global_sigma": glob["global_sigma"]},
        }
        if (n, K, T, args.algo) == (100_000_000, 100_000, 250, "EWMA"):   # the PMC passes were taken on this workload
            out["roofline"]["traffic"] = pmc_traffic({2: "k_partition", 3: "k_partition_wc"}.get(st["stage0_path"], "k_scatter"))
        if args.algo == "ARIMA":
            sec = acc["ms_detect"] / steps * 1e-3
            flops = FLOP_PER_KALMAN_STEP * st["kalman_steps"]
            out["arima"] = {"fits_per_launch": st["arima_fits"], "kalman_steps_per_launch": st["kalman_steps"],
                            "fits_per_s": st["arima_fits"] / sec, "bound": "fp64 vector ALU / dependency latency",
                            "achieved_tflops": flops / sec / 1e12, "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                            "frac": flops / sec / 1e12 / FP64_VECTOR_PEAK_TFLOPS, "ms_detect": acc["ms_detect"] / steps}
        if world == 1 and not args.no_cpu_baseline:
            crow = min(args.cpu_rows, n)
            ckeys = max(1, int(K * crow / n))
            out["cpu_baseline"] = cpu_baseline(args.algo, crow, ckeys, T, args.agg)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
