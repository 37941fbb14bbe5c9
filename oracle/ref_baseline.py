"""Times the reference's OWN per-series functions — MEASUREMENT INFRASTRUCTURE, build container only.

SURVEY.md 8d / BASELINE.md 3 ask for the reference's as-written CPU path next to the GPU numbers.  The PySpark job cannot
run here (no JVM, no pyspark), but the functions Spark maps over the grouped series can: this script loads
/root/reference/plugins/anomaly-detection/anomaly_detection.py through oracle/ref_loader.py (stub modules for pyspark /
statsmodels, the functions run unmodified from where they lie), builds the synthetic table of SURVEY.md 8d at a bounded size,
does Stage 0 + the series assembly with pandas (what ClickHouse and Spark's groupby / collect_list / stddev_samp do), and then
calls, per key, exactly what plot_anomaly's rdd.map calls (anomaly_detection.py:440-443):

    EWMA   : calculate_ewma(series)  and  calculate_ewma_anomaly(series, stddev)     (:146-212; the EWMA is computed twice)
    DBSCAN : calculate_dbscan(series) and calculate_dbscan_anomaly(series, stddev)   (:312-349; sklearn per key)

in one process and over all cores of this container (multiprocessing, keys split evenly — Spark local[*] on the same
functions).  ARIMA is absent: calculate_arima needs statsmodels.  /root/reference does not exist on the GPU box, so the figures
are COMMITTED (profiles/r3_reference_functions_cpu.json) and bench.py copies them into `cpu_baseline.reference_functions`,
labelled with where they were measured.   usage: python oracle/ref_baseline.py [rows] > profiles/r3_reference_functions_cpu.json
"""
import json
import multiprocessing as mp
import os
import sys
import time
from decimal import Decimal

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader, tad_oracle as orc  # noqa: E402

_REF = None
_SERIES = None


def _ref():
    global _REF
    if _REF is None:
        _REF = ref_loader.load_reference_job()
    return _REF


def _run_keys(arg):
    algo, lo, hi = arg
    ref = _ref()
    n_anom = 0
    for vals, sd in _SERIES[lo:hi]:
        if algo == "EWMA":
            ref.calculate_ewma(vals)
            verdict = ref.calculate_ewma_anomaly(vals, sd)
        else:
            ref.calculate_dbscan(vals)
            verdict = ref.calculate_dbscan_anomaly(vals, sd)
        n_anom += sum(1 for b in verdict if b)
    return n_anom


def measure(algo, rows, K, T, op, cores):
    global _SERIES
    k, t, v = orc.synth_rows(0, rows, K, T)
    t0 = time.perf_counter()
    df = pd.DataFrame({"k": k, "t": t, "v": v})
    pts = df.groupby(["k", "t"], sort=True)["v"].agg("sum" if op == "sum" else "max").reset_index()   # Stage 0 (ClickHouse's GROUP BY)
    grp = pts.groupby("k", sort=True)["v"]
    sd = grp.std(ddof=1)                                                                                 # stddev_samp
    # the reference's UDFs receive lists of decimal.Decimal (clickhouse-jdbc -> Spark DecimalType)
    series = [([Decimal(int(x)) for x in g.to_numpy()], (None if np.isnan(sd[key]) else float(sd[key]))) for key, g in grp]
    t_stage = time.perf_counter() - t0
    _SERIES = series
    _ref()                                                  # module load outside the timed region
    _run_keys((algo, 0, min(8, len(series))))               # first-call warm-up
    t0 = time.perf_counter()
    a1 = _run_keys((algo, 0, len(series)))
    t_one = time.perf_counter() - t0
    bounds = [(algo, len(series) * i // cores, len(series) * (i + 1) // cores) for i in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_run_keys, [(algo, 0, 0)] * cores)          # workers up, reference module loaded
        t0 = time.perf_counter()
        a2 = sum(pool.map(_run_keys, bounds))
        t_all = time.perf_counter() - t0
    assert a1 == a2
    return {"algo": algo, "rows": rows, "keys": K, "buckets": T, "points": int(len(pts)), "anomalies": int(a1),
            "stage0_and_series_assembly_pandas_s": round(t_stage, 3), "udf_one_core_s": round(t_one, 3), "udf_all_cores_s": round(t_all, 3),
            "rows_per_s_one_core_udf_only": rows / t_one, "rows_per_s_all_cores_udf_only": rows / t_all,
            "rows_per_s_one_core_with_pandas_stage0": rows / (t_one + t_stage), "rows_per_s_all_cores_with_pandas_stage0": rows / (t_all + t_stage)}


if __name__ == "__main__":
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    cores = os.cpu_count() or 1
    out = {"what": "the reference's own calculate_ewma / calculate_ewma_anomaly / calculate_dbscan / calculate_dbscan_anomaly "
                   "(plugins/anomaly-detection/anomaly_detection.py:146-212, 312-349), imported from /root/reference and run unmodified on "
                   "the synthetic table of SURVEY.md 8d; Stage 0 + series assembly by pandas; per key both functions as plot_anomaly's rdd.map does",
           "where": "build container (no GPU box has /root/reference)", "cores": cores,
           "cpu": open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip() if os.path.exists("/proc/cpuinfo") else "",
           "c2_shape": measure("EWMA", rows, max(1, rows // 1000), 250, "sum", cores),           # C2's rows-per-key ratio (1000 rows / key, ~4 rows / point)
           "c4_shape": measure("DBSCAN", rows, max(1, rows // 100), 100, "max", cores)}          # C4's (100 rows / key, ~1 row / point)
    print(json.dumps(out, indent=1))
