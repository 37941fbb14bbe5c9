"""CPU oracle for the Throughput Anomaly Detection path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of what the reference job computes
(/root/reference/plugins/anomaly-detection/anomaly_detection.py), used only as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
(theia_amd/ -> libtad_mi355x.so) never imports this package.

Pinned against the reference's own golden vectors (anomaly_detection_test.py:199-402) and against
outputs of the reference's pure functions run in the build container (tests/golden/*.json, made by
oracle/make_golden.py).  What is NOT pinned by any reference unit test and is therefore defined
here: series order = ascending flowEndSeconds per key; stddev_samp = Spark's
CentralMomentAgg streaming update in that order, null for n = 1; ARIMA-None -> key yields no rows;
sentinel row when the global anomaly count is 0.

Each function cites the reference lines it restates.
"""
import numpy as np

U64 = np.uint64
MASK64 = (1 << 64) - 1

# ----------------------------------------------------------------------------------------------
# Deterministic synthetic flow table.  Mirrors theia_amd/csrc/tad_synth.hip.
# ----------------------------------------------------------------------------------------------
SYNTH_SEED = 0x7AD05EED
SYNTH_T_BASE = 1660202814   # 2022-08-11T07:26:54Z, test/e2e/throughputanomalydetection_test.go:402-403
SYNTH_T_STEP = 60           # one point per minute, :440
_GOLDEN = U64(0x9E3779B97F4A7C15)


def mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping)."""
    z = np.asarray(z, dtype=U64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> U64(30))) * U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> U64(27))) * U64(0x94D049BB133111EB)
        return z ^ (z >> U64(31))


def _h(seed, stream, i):
    with np.errstate(over="ignore"):
        return mix64(U64(seed) + _GOLDEN * (i * U64(8) + U64(stream)))


def synth_rows(first_row, n_rows, num_keys, n_buckets, seed=SYNTH_SEED):
    """Rows [first_row, first_row+n_rows) of the synthetic table -> (key_id u64, flow_end_s i64, value u64)."""
    i = np.arange(first_row, first_row + n_rows, dtype=U64)
    key = _h(seed, 1, i) % U64(num_keys)
    bucket = _h(seed, 2, i) % U64(n_buckets)
    t = (np.int64(SYNTH_T_BASE) + np.int64(SYNTH_T_STEP) * bucket.astype(np.int64))
    with np.errstate(over="ignore"):
        base = U64(1000000000) + mix64(U64(seed) ^ (_GOLDEN * (key + U64(3)))) % U64(3000000000)
        j = base // U64(1000)
        v = base + (_h(seed, 4, i) % (U64(2) * j + U64(1))) - j
        r = _h(seed, 5, i)
        spike = (r & U64(8191)) == 0
        dip = (~spike) & (((r >> U64(13)) & U64(16383)) == 0)
        v = np.where(spike, v * (U64(2) + (r >> U64(20)) % U64(10)), v)
        v = np.where(dip, v // (U64(2) + (r >> U64(40)) % U64(18)), v)
    return key, t, v.astype(U64)


# ----------------------------------------------------------------------------------------------
# Stage 0: the GROUP BY the reference pushes into ClickHouse (anomaly_detection.py:507-614)
# ----------------------------------------------------------------------------------------------
KEY_SKIP = U64(MASK64)


def stage0(key_id, flow_end_s, value, op, key_id2=None, flow_start_s=None, start_time=0, end_time=0):
    """GROUP BY (key, flowEndSeconds) with max() or sum() over UInt64.

    op: "max" (mode None, :52-61) or "sum" (pod/svc/external, :63-106).  sum wraps mod 2^64, max is
    unsigned (ClickHouse UInt64, create_table.sh:74).  Row filters: flowStartSeconds >= start
    (:581-583), flowEndSeconds < end (:584-586); KEY_SKIP models every string predicate the host
    evaluated.  key_id2: pod mode's UNION ALL (:556-565) — a row also counts for its second key.
    Returns points sorted by (key, t): key u64[P], t i64[P], v u64[P].
    """
    key_id = np.asarray(key_id, dtype=U64)
    t = np.asarray(flow_end_s, dtype=np.int64)
    v = np.asarray(value, dtype=U64)
    keep = np.ones(key_id.shape, dtype=bool)
    if start_time and flow_start_s is not None:
        keep &= np.asarray(flow_start_s, dtype=np.int64) >= start_time
    if end_time:
        keep &= t < end_time
    ks = [key_id]
    if key_id2 is not None:
        ks.append(np.asarray(key_id2, dtype=U64))
    kk = np.concatenate([k[keep & (k != KEY_SKIP)] for k in ks])
    tt = np.concatenate([t[keep & (k != KEY_SKIP)] for k in ks])
    vv = np.concatenate([v[keep & (k != KEY_SKIP)] for k in ks])
    if kk.size == 0:
        return kk, tt, vv
    order = np.lexsort((tt, kk))
    kk, tt, vv = kk[order], tt[order], vv[order]
    new = np.ones(kk.size, dtype=bool)
    new[1:] = (kk[1:] != kk[:-1]) | (tt[1:] != tt[:-1])
    starts = np.flatnonzero(new)
    if op == "sum":
        with np.errstate(over="ignore"):
            agg = np.add.reduceat(vv, starts)     # uint64: wraps
    elif op == "max":
        agg = np.maximum.reduceat(vv, starts)
    else:
        raise ValueError(op)
    return kk[starts], tt[starts], agg.astype(U64)


def series_offsets(pkey):
    """CSR offsets of the per-key series in a (key, t)-sorted point list: keys u64[K'], ptr i64[K'+1]."""
    if pkey.size == 0:
        return pkey, np.zeros(1, dtype=np.int64)
    new = np.ones(pkey.size, dtype=bool)
    new[1:] = pkey[1:] != pkey[:-1]
    starts = np.flatnonzero(new)
    return pkey[starts], np.append(starts, pkey.size).astype(np.int64)


def _padded(values_f64, ptr):
    """[K', maxn] matrix + length vector, so that per-key SEQUENTIAL recurrences run in lockstep."""
    n = np.diff(ptr)
    maxn = int(n.max()) if n.size else 0
    idx = ptr[:-1, None] + np.arange(maxn)[None, :]
    valid = np.arange(maxn)[None, :] < n[:, None]
    mat = np.where(valid, values_f64[np.minimum(idx, values_f64.size - 1)], 0.0)
    return mat, valid, n


# ----------------------------------------------------------------------------------------------
# Stage 1: stddev_samp per key (anomaly_detection.py:674-684)
# ----------------------------------------------------------------------------------------------
def stddev_samp_series(x):
    """Spark CentralMomentAgg, single partition, in series order.

    n += 1; d = x - avg; dn = d / n; avg += dn; m2 += d * (d - dn); result sqrt(m2 / (n - 1));
    None when n < 2 (Spark >= 3.1: null).  x: float64 values (u64 -> double, correctly rounded).
    """
    x = np.asarray(x, dtype=np.float64)
    if x.size < 2:
        return None
    n = 0.0
    avg = 0.0
    m2 = 0.0
    for xv in x.tolist():
        n += 1.0
        d = xv - avg
        dn = d / n
        avg = avg + dn
        m2 = m2 + d * (d - dn)
    return float(np.sqrt(np.float64(m2 / (n - 1.0))))


def stddev_samp_all(pv_f64, ptr):
    """Vectorised-over-keys version of stddev_samp_series (identical operation order per key).
    Returns (sigma f64[K'], has_sigma bool[K'])."""
    mat, valid, n = _padded(pv_f64, ptr)
    K = mat.shape[0]
    cnt = np.zeros(K)
    avg = np.zeros(K)
    m2 = np.zeros(K)
    for j in range(mat.shape[1]):
        m = valid[:, j]
        xv = mat[:, j]
        c1 = cnt + 1.0
        d = xv - avg
        dn = d / c1
        avg = np.where(m, avg + dn, avg)
        m2 = np.where(m, m2 + d * (d - dn), m2)
        cnt = np.where(m, c1, cnt)
    has = n >= 2
    with np.errstate(invalid="ignore", divide="ignore"):
        sigma = np.sqrt(m2 / (cnt - 1.0))
    return np.where(has, sigma, 0.0), has


# ----------------------------------------------------------------------------------------------
# Stage 2: detectors
# ----------------------------------------------------------------------------------------------
def calculate_ewma(throughput_list, alpha=0.5):
    """anomaly_detection.py:146-165: e_t = (1 - alpha) * e_{t-1} + alpha * float(x_t), e_{-1} = 0."""
    prev = 0.0
    out = []
    for ele in throughput_list:
        cur = (1 - alpha) * prev + alpha * float(ele)
        prev = cur
        out.append(float(cur))
    return out


def calculate_ewma_anomaly(throughput_row, stddev, alpha=0.5):
    """anomaly_detection.py:168-212: |float(x_t) - e_t| > float(stddev), strict; stddev None -> False."""
    e = calculate_ewma(throughput_row, alpha)
    if stddev is None:
        return [False] * len(e)
    s = float(stddev)
    return [abs(float(x) - ev) > s for x, ev in zip(throughput_row, e)]


def ewma_all(pv_f64, ptr, alpha=0.5):
    """EWMA for every key in lockstep (same per-key op order as calculate_ewma). Returns f64[P]."""
    mat, valid, n = _padded(pv_f64, ptr)
    e = np.zeros(mat.shape[0])
    out = np.zeros_like(mat)
    for j in range(mat.shape[1]):
        e = (1 - alpha) * e + alpha * mat[:, j]
        out[:, j] = e
    return out[valid]  # row-major order == (key, t) order


def dbscan_noise_1d(x, eps=250000000.0, min_samples=4):
    """anomaly_detection.py:325-349 — DBSCAN(min_samples=4, eps=250000000).fit_predict(x.reshape(-1,1)) == -1.

    Restated semantics (checked against sklearn in tests/test_oracle.py):
    i is core iff #{j : |x_i - x_j| <= eps} >= min_samples (self counted); noise iff not core and no
    core point within eps.  Distances in float64 after u64 -> double conversion.
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n == 0:
        return np.zeros(0, dtype=bool)
    d = np.abs(x[:, None] - x[None, :]) <= eps
    core = d.sum(axis=1) >= min_samples
    reach = (d & core[None, :]).any(axis=1)
    return ~(core | reach)


def calculate_dbscan_anomaly(throughput_row, stddev=None, eps=250000000.0, min_samples=4):
    return dbscan_noise_1d(np.array([float(v) for v in throughput_row]), eps, min_samples).tolist()


def calculate_dbscan(throughput_list):
    """anomaly_detection.py:312-322: algoCalc placeholder 0.0 per point."""
    return [0.0] * len(throughput_list)


# ----------------------------------------------------------------------------------------------
# Whole job (anomaly_detection.py:647-710 + 352-421): columnar batch -> anomalous points
# ----------------------------------------------------------------------------------------------
def u64_to_f64(v):
    """float(Decimal(x)) of anomaly_detection.py:161,231: correctly rounded uint64 -> double."""
    return np.asarray(v, dtype=U64).astype(np.float64)


def run_job(algo, key_id, flow_end_s, value, op=None, agg_flow="", key_id2=None, flow_start_s=None,
            start_time=0, end_time=0, alpha=0.5, eps=250000000.0, min_samples=4, arima_fn=None):
    """Returns dict with the anomalous points ordered by (key, t) and the counters tad_stats carries.

    algo in {"EWMA", "ARIMA", "DBSCAN"}.  op defaults to max for agg_flow "" / None and sum otherwise.
    """
    if op is None:
        op = "sum" if agg_flow else "max"
    pk, pt, pv = stage0(key_id, flow_end_s, value, op, key_id2, flow_start_s, start_time, end_time)
    keys, ptr = series_offsets(pk)
    pvf = u64_to_f64(pv)
    sigma, has_sigma = stddev_samp_all(pvf, ptr) if pk.size else (np.zeros(0), np.zeros(0, bool))
    n = np.diff(ptr)
    sig_pt = np.repeat(sigma, n)
    has_pt = np.repeat(has_sigma, n)
    keys_no_result = 0
    arima_counters, arima_results = {}, None
    if algo == "EWMA":
        calc = ewma_all(pvf, ptr, alpha) if pk.size else np.zeros(0)
        with np.errstate(invalid="ignore"):
            anomaly = has_pt & (np.abs(pvf - calc) > sig_pt)
    elif algo == "DBSCAN":
        calc = np.zeros(pk.size)
        anomaly = np.zeros(pk.size, dtype=bool)
        for a, b in zip(ptr[:-1], ptr[1:]):
            anomaly[a:b] = dbscan_noise_1d(pvf[a:b], eps, min_samples)
    elif algo == "ARIMA":
        if arima_fn is None:
            from oracle.arima_oracle import calculate_arima_exact as arima_fn   # the fixed-arithmetic restatement
        calc = np.zeros(pk.size)
        anomaly = np.zeros(pk.size, dtype=bool)
        arima_results = []
        for k, (a, b) in enumerate(zip(ptr[:-1], ptr[1:])):
            try:
                pred = arima_fn(pv[a:b], counters=arima_counters)
            except TypeError:         # a plain callable(series) was passed in
                pred = arima_fn(pv[a:b])
            arima_results.append(pred)
            if pred is None:          # :284-287 + arrays_zip/explode of a null array: no rows
                keys_no_result += 1
                continue
            calc[a:b] = pred
            if has_sigma[k]:
                anomaly[a:b] = np.abs(pvf[a:b] - calc[a:b]) > sigma[k]
    else:
        raise ValueError(algo)
    sel = np.flatnonzero(anomaly)
    return {
        "key_id": pk[sel], "flow_end_s": pt[sel], "throughput": pvf[sel], "algo_calc": calc[sel],
        "stddev": sig_pt[sel], "n_anomalies": int(sel.size), "n_keys": int(keys.size),
        "n_points": int(pk.size), "keys_no_result": keys_no_result,
        # everything, for deeper comparisons
        "points": (pk, pt, pv), "sigma": sigma, "has_sigma": has_sigma, "keys": keys, "ptr": ptr,
        "calc_all": calc, "anomaly_all": anomaly,
        "arima_results": arima_results, "kalman_steps": int(arima_counters.get("kalman_steps", 0)),
    }


# ----------------------------------------------------------------------------------------------
# Full-size helpers (BASELINE C2 / C4 tables, 1e8 rows): the same semantics as stage0 / dbscan_noise_1d by a
# DIFFERENT, faster route, for the per-point parity tests at the benchmarked sizes (tests/test_gpu_fullsize.py).
# tests/test_oracle.py checks them against stage0 / dbscan_noise_1d on small inputs.
# ----------------------------------------------------------------------------------------------
def stage0_dense(key_id, flow_end_s, value, op, num_keys, t0, step, n_buckets):
    """GROUP BY (key, flowEndSeconds) on a dense lattice without sorting.  sum: the 64-bit values are split into 32-bit
    halves whose per-cell float64 bincount sums are exact (< 2^53 for < 2^21 rows per cell) and recombined mod 2^64;
    max: numpy.maximum.at on uint64.  Returns the points in (key, t) order like stage0."""
    key_id = np.asarray(key_id, dtype=U64)
    t = np.asarray(flow_end_s, dtype=np.int64)
    v = np.asarray(value, dtype=U64)
    bucket = (t - np.int64(t0)) // np.int64(step)
    assert ((t - np.int64(t0)) % np.int64(step) == 0).all() and (bucket >= 0).all() and (bucket < n_buckets).all()
    cell = key_id.astype(np.int64) * np.int64(n_buckets) + bucket
    ncell = int(num_keys) * int(n_buckets)
    cnt = np.bincount(cell, minlength=ncell)
    assert cnt.max() < (1 << 21)
    if op == "sum":
        lo = np.bincount(cell, weights=(v & U64(0xFFFFFFFF)).astype(np.float64), minlength=ncell)
        hi = np.bincount(cell, weights=(v >> U64(32)).astype(np.float64), minlength=ncell)
        with np.errstate(over="ignore"):
            agg = (hi.astype(U64) << U64(32)) + lo.astype(U64)        # wraps mod 2^64 like ClickHouse's UInt64 sum
    elif op == "max":
        agg = np.zeros(ncell, dtype=U64)
        np.maximum.at(agg, cell, v)
    else:
        raise ValueError(op)
    present = np.flatnonzero(cnt)
    pk = (present // n_buckets).astype(U64)
    pt = np.int64(t0) + np.int64(step) * (present % n_buckets)
    return pk, pt, agg[present]


def dbscan_noise_all(pv_f64, ptr, eps=250000000.0, min_samples=4, chunk=4096):
    """dbscan_noise_1d for every key at once: padded [keys, maxn] matrix, pairwise tests in chunks of keys."""
    mat, valid, n = _padded(pv_f64, ptr)
    out = np.zeros_like(valid)
    for a in range(0, mat.shape[0], chunk):
        m, ok = mat[a:a + chunk], valid[a:a + chunk]
        near = (np.abs(m[:, :, None] - m[:, None, :]) <= eps) & ok[:, :, None] & ok[:, None, :]
        core = near.sum(axis=2) >= min_samples
        reach = (near & core[:, None, :]).any(axis=2)
        out[a:a + chunk] = ~(core | reach)
    return out[valid]


def _synth_chunk(args):
    return synth_rows(*args)


def synth_rows_parallel(n_rows, num_keys, n_buckets, procs=None, chunk=5_000_000):
    """synth_rows(0, n_rows, ...) generated by a process pool (the 1e8-row tables take ~50 s on one core)."""
    import concurrent.futures as cf
    import os
    jobs = [(i, min(chunk, n_rows - i), num_keys, n_buckets) for i in range(0, n_rows, chunk)]
    procs = procs or min(len(jobs), max(1, (os.cpu_count() or 1) - 2), 32)
    if procs <= 1:
        parts = [synth_rows(*j) for j in jobs]
    else:
        import multiprocessing as mp
        # spawn, not fork: the GPU tests call this from a process that has the HIP runtime loaded
        with cf.ProcessPoolExecutor(procs, mp_context=mp.get_context("spawn")) as ex:
            parts = list(ex.map(_synth_chunk, jobs))
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))
