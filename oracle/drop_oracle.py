"""Abnormal-traffic-drop detector oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates /root/reference/snowflake/udfs/udfs/drop_detection/drop_detection_udf.py:21-56 (DropDetection.end_partition):
per (endpoint, direction) partition of daily drop counts — skip when fewer than 3 samples (:44-45); mean and sample
standard deviation of the counts (:47-48, pandas Series.mean / Series.std, ddof = 1); a day is anomalous when its
count lies outside mean +- 3 std (:49-52); one result row per anomalous day carrying mean and std (:53-56).

The arithmetic lives in pandas / numpy (not under /root/reference): Series.mean = sum / n and Series.std =
sqrt(sum((mean - x)^2) / (n - 1)) with numpy's float64 add-reduce, i.e. PAIRWISE summation (8 interleaved
accumulators up to 128 elements, recursive halving above).  `pairwise_sum` restates that order so that the GPU kernel
can be held to the same bits; tests/test_oracle_drop.py pins this file against the reference UDF itself (run from
where it lies in the build container -> tests/golden/drop_outputs.json) and against the reference's own golden
(drop_detection_udf_test.py:130-139: avg 8.0, stdev 21.7037469479108, anomaly 2022-01-05 = 100).
"""
import importlib.util
import os

import numpy as np

REF_ROOT = os.environ.get("THEIA_REFERENCE", "/root/reference")
UDF_FILE = os.path.join(REF_ROOT, "snowflake", "udfs", "udfs", "drop_detection", "drop_detection_udf.py")
N_SIGMA = 3.0       # drop_detection_udf.py:49-50
MIN_SAMPLES = 3     # :44


def reference_available():
    return os.path.isfile(UDF_FILE)


def load_reference_udf():
    """The reference module, imported from where it lies (pandas only).  Build container only."""
    spec = importlib.util.spec_from_file_location("theia_ref_drop_detection_udf", UDF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pairwise_sum(a):
    """numpy's pairwise_sum_DOUBLE on a contiguous float64 vector (numpy/core/src/umath/loops_utils.h)."""
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r += float(v)
        return r
    if n <= 128:
        r = [float(v) for v in a[:8]]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += float(a[i + j])
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += float(a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum(a[:n2]) + pairwise_sum(a[n2:])


def drop_stats(x):
    """(mean, std) of the series with pandas' arithmetic; x = float64 counts in date order."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    mean = pairwise_sum(x) / n
    sq = (mean - x) ** 2
    var = pairwise_sum(sq) / (n - 1)
    return float(mean), float(np.sqrt(var))


def drop_detection_series(x, n_sigma=N_SIGMA, min_samples=MIN_SAMPLES):
    """-> None (too few samples) or (mean, std, verdict bool[n])."""
    x = np.asarray(x, dtype=np.float64)
    if x.size < min_samples:
        return None
    mean, std = drop_stats(x)
    upper, lower = mean + n_sigma * std, mean - n_sigma * std
    with np.errstate(invalid="ignore"):
        verdict = (x > upper) | (x < lower)
    return mean, std, verdict


def run_job(key_id, day_s, drop_number, n_sigma=N_SIGMA, min_samples=MIN_SAMPLES):
    """Columnar batch -> anomalous (key, day) rows, ordered by (key, day).  Stage 0 = SUM(dropNumber) GROUP BY
    endpoint, direction, date (snowflake/cmd/dropDetection.go:151-162) through the TAD oracle's integer group-by."""
    from . import tad_oracle as orc
    pk, pt, pv = orc.stage0(key_id, day_s, drop_number, "sum")
    keys, ptr = orc.series_offsets(pk)
    xf = orc.u64_to_f64(pv)
    sel, means, stds = [], [], []
    skipped = 0
    for a, b in zip(ptr[:-1], ptr[1:]):
        r = drop_detection_series(xf[a:b], n_sigma, min_samples)
        if r is None:
            skipped += 1
            continue
        mean, std, verdict = r
        idx = np.flatnonzero(verdict) + a
        sel.append(idx)
        means.append(np.full(idx.size, mean))
        stds.append(np.full(idx.size, std))
    sel = np.concatenate(sel) if sel else np.zeros(0, dtype=np.int64)
    return {"key_id": pk[sel], "flow_end_s": pt[sel], "throughput": xf[sel],
            "algo_calc": np.concatenate(means) if means else np.zeros(0),
            "stddev": np.concatenate(stds) if stds else np.zeros(0),
            "n_anomalies": int(sel.size), "n_keys": int(keys.size), "n_points": int(pk.size), "keys_no_result": skipped}
