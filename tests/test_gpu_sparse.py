"""GPU: the sparse Stage-0 path (tad_sparse.hip) — tables whose dense keys x time-lattice grid would be mostly empty or
would not fit: second-resolution timestamps with gcd 1 over a day, and a million short-lived per-connection keys (mode
None: the key contains flowStartSeconds, anomaly_detection.py:52-61, 109-116).  Results must equal the oracle bit for
bit like the dense path's, and the run time must follow the rows, not keys x lattice."""

import numpy as np
import pytest

from oracle import tad_oracle as orc
from theia_amd import TadError

pytestmark = pytest.mark.gpu


def check(engine, algo, k, t, v, K, agg_flow, paths=(4,), **kw):
    want = orc.run_job(algo, k, t, v, agg_flow=agg_flow, **kw)
    allp = engine.run(algo, k, t, v, K, agg_flow=agg_flow, emit_all=True, **kw)
    assert allp.stats["stage0_path"] in paths
    pk, pt, pv = want["points"]
    if algo == "ARIMA":
        keep = np.repeat(np.array([r is not None for r in want["arima_results"]]), np.diff(want["ptr"]))
    else:
        keep = np.ones(pk.size, dtype=bool)
    assert allp.n_rows == int(keep.sum())
    assert (allp["key_id"] == pk[keep]).all() and (allp["flow_end_s"] == pt[keep]).all()
    assert (allp["throughput"] == orc.u64_to_f64(pv)[keep]).all()
    assert (allp["stddev"] == np.repeat(want["sigma"], np.diff(want["ptr"]))[keep]).all()
    assert np.array_equal(allp["algo_calc"], want["calc_all"][keep], equal_nan=True)
    assert (allp["anomaly"].astype(bool) == want["anomaly_all"][keep]).all()
    res = engine.run(algo, k, t, v, K, agg_flow=agg_flow, **kw)
    assert res.stats["stage0_path"] in paths and res.n_rows == want["n_anomalies"]
    for f in ("key_id", "flow_end_s", "throughput", "stddev"):
        assert (res[f] == want[f]).all(), f
    assert np.array_equal(res["algo_calc"], want["algo_calc"], equal_nan=True)
    assert res.stats["n_keys"] == want["n_keys"] and res.stats["n_points"] == want["n_points"]
    return res, want


def day_table(K, pts_per_key, rows_per_point, seed, span=86400):
    """second-resolution timestamps anywhere in a day (gcd 1), a few rows per (key, second)"""
    rng = np.random.default_rng(seed)
    P = K * pts_per_key
    pk = np.repeat(np.arange(K, dtype=np.uint64), pts_per_key)
    pt = 1660202814 + rng.integers(0, span, size=P).astype(np.int64)
    base = 1_000_000_000 + (orc.mix64(pk + np.uint64(3)) % np.uint64(3_000_000_000)).astype(np.int64)
    k = np.repeat(pk, rows_per_point)
    t = np.repeat(pt, rows_per_point)
    v = (np.repeat(base, rows_per_point) + rng.integers(-1_000_000, 1_000_000, size=k.size)).astype(np.uint64)
    spike = rng.random(k.size) < 2e-3
    v = np.where(spike, v * np.uint64(7), v)
    order = rng.permutation(k.size)
    return k[order], t[order], v[order]


@pytest.mark.parametrize("algo", ["EWMA", "DBSCAN"])
def test_gcd1_timestamps_over_a_day(engine, algo):
    K = 20000
    k, t, v = day_table(K, 50, 3, seed=1)                       # 3e6 rows; dense grid would be 20000 x 86400 cells = 15.6 GB
    res, want = check(engine, algo, k, t, v, K, "svc")
    assert res.stats["step"] == 1 and res.stats["n_buckets"] > 86000
    print("%s: %d rows, %d points, sparse job %.2f ms" % (algo, k.size, want["n_points"], res.stats["ms_total"]))
    assert res.stats["ms_total"] < 60.0


def test_gcd1_arima(engine):
    K = 300
    k, t, v = day_table(K, 40, 2, seed=2)
    k, t, v = k[:], t[:], v[:]
    with engine.plan(sparse="always"):
        check(engine, "ARIMA", k, t, v, K, "svc")


def test_a_million_short_lived_connections(engine):
    # mode None: one key per connection, 2..7 points each at minute resolution somewhere in a day, max(throughput)
    rng = np.random.default_rng(5)
    K = 1_000_000
    n_k = rng.integers(2, 8, size=K)
    pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
    start = rng.integers(0, 1430, size=K)
    pos = np.concatenate([np.arange(n) for n in n_k])
    pt = (1660202814 + 60 * (np.repeat(start, n_k) + pos)).astype(np.int64)
    v = (rng.integers(1, 4_000_000_000, size=pk.size)).astype(np.uint64)
    dup = rng.random(pk.size) < 0.3                              # some (key, time) points have two rows: max() picks one
    k = np.concatenate([pk, pk[dup]]); t = np.concatenate([pt, pt[dup]])
    vv = np.concatenate([v, (v[dup] // np.uint64(2))])
    order = rng.permutation(k.size)
    k, t, vv = k[order], t[order], vv[order]
    res, want = check(engine, "DBSCAN", k, t, vv, K, "", paths=(8,))    # 5.8e6 rows: pass A ran, the key-block partition pass + LDS sorts (tests/test_gpu_sparse_partition.py)
    with engine.plan(sparse_sort="lsd"):
        check(engine, "DBSCAN", k, t, vv, K, "")
    # n_k < 4 = min_samples -> no core point -> every point of such a key is an anomaly
    assert want["n_anomalies"] > 1_000_000
    print("1e6 connections: %d rows, %d points, %d anomalies, sparse job %.2f ms" % (k.size, want["n_points"], want["n_anomalies"], res.stats["ms_total"]))
    assert res.stats["ms_total"] < 80.0                          # a dense 1e6 x 1440 grid alone would be 13 GB to clear and walk


def test_sparse_with_second_key_time_window_and_aggregate(engine):
    k, t, v = day_table(400, 30, 2, seed=3, span=7200)
    rng = np.random.default_rng(4)
    k2 = (orc.mix64(k + np.uint64(99)) % np.uint64(400)).astype(np.uint64)
    k2 = np.where(k2 % np.uint64(5) == 0, orc.KEY_SKIP, k2)
    k = np.where(k % np.uint64(7) == 0, orc.KEY_SKIP, k)
    ts = t - rng.integers(0, 600, size=t.size)
    with engine.plan(sparse="always"):
        check(engine, "EWMA", k, t, v, 400, "pod", key_id2=k2, flow_start_s=ts, start_time=int(t.min()) + 100, end_time=int(t.max()) - 300)
        pts = engine.aggregate(k, t, v, 400, agg_flow="pod", key_id2=k2)
        pk, pt, pv = orc.stage0(k, t, v, "sum", k2)
        assert pts.stats["stage0_path"] == 4 and pts.n_points == pk.size
        assert (pts["key_id"] == pk).all() and (pts["flow_end_s"] == pt).all() and (pts["value"] == pv).all()
        # wrap-around sums and values beyond 2^53 are exact here too (plain u64 arithmetic, no packed records)
        big = np.where(rng.random(v.size) < 0.05, rng.integers(2**62, 2**64 - 1, size=v.size, dtype=np.uint64), v)
        check(engine, "EWMA", k, t, big, 400, "svc")
        with pytest.raises(TadError):
            engine.run("EWMA", np.array([500], dtype=np.uint64), t[:1], v[:1], 400, agg_flow="svc")        # key id out of range


def skewed_table(K, long_keys, long_len, seed, span=86400):
    """a few keys with a point at (almost) every second of a day, the rest with 1..12 points: the rank grid of the sparse path
    would be K x long_len cells for ~K * 6 points"""
    rng = np.random.default_rng(seed)
    n_k = rng.integers(1, 13, size=K)
    n_k[rng.choice(K, size=long_keys, replace=False)] = long_len
    n_k[rng.choice(K, size=K // 50, replace=False)] = 0          # keys without rows
    pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
    pt = np.concatenate([np.sort(rng.choice(span, size=n, replace=False)) for n in n_k]).astype(np.int64) + 1660202814
    base = 1_000_000_000 + (orc.mix64(pk + np.uint64(3)) % np.uint64(3_000_000_000)).astype(np.int64)
    v = (base + rng.integers(-1_000_000, 1_000_000, size=pk.size)).astype(np.uint64)
    v = np.where(rng.random(pk.size) < 3e-3, v * np.uint64(9), v)
    dup = rng.random(pk.size) < 0.2                                # some points have two rows
    k = np.concatenate([pk, pk[dup]]); t = np.concatenate([pt, pt[dup]]); v = np.concatenate([v, v[dup] // np.uint64(3)])
    order = rng.permutation(k.size)
    return k[order], t[order], v[order]


def check_classes(engine, algo, k, t, v, K, agg_flow):
    want = orc.run_job(algo, k, t, v, agg_flow=agg_flow)
    allp = engine.run(algo, k, t, v, K, agg_flow=agg_flow, emit_all=True)
    assert allp.stats["stage0_path"] == 6
    pk, pt, pv = want["points"]
    keep = np.ones(pk.size, dtype=bool)
    if algo == "ARIMA":
        keep = np.repeat(np.array([r is not None for r in want["arima_results"]]), np.diff(want["ptr"]))
    assert allp.n_rows == int(keep.sum())
    assert (allp["key_id"] == pk[keep]).all() and (allp["flow_end_s"] == pt[keep]).all()          # merged back in (key, time) order
    assert (allp["throughput"] == orc.u64_to_f64(pv)[keep]).all()
    assert (allp["stddev"] == np.repeat(want["sigma"], np.diff(want["ptr"]))[keep]).all()
    assert np.array_equal(allp["algo_calc"], want["calc_all"][keep], equal_nan=True)
    assert (allp["anomaly"].astype(bool) == want["anomaly_all"][keep]).all()
    assert allp.stats["n_anomalies"] == int(want["anomaly_all"][keep].sum())
    res = engine.run(algo, k, t, v, K, agg_flow=agg_flow)
    assert res.stats["stage0_path"] == 6 and res.n_rows == want["n_anomalies"] == res.stats["n_anomalies"]
    for f in ("key_id", "flow_end_s", "throughput", "stddev"):
        assert (res[f] == want[f]).all(), f
    assert np.array_equal(res["algo_calc"], want["algo_calc"], equal_nan=True)
    assert res.stats["n_keys"] == want["n_keys"] and res.stats["n_points"] == want["n_points"] and res.stats["rows_used"] == k.size
    gm = orc.u64_to_f64(pv).mean()
    assert abs(res.stats["pts_mean"] - gm) <= 1e-12 * gm
    return res, want


@pytest.mark.parametrize("algo,agg", [("EWMA", "svc"), ("DBSCAN", "")])
def test_skewed_series_lengths_run_as_length_classes(algo, agg):
    """one key with a day of seconds next to thousands of short-lived ones: the K x longest-series rank grid does not fit
    the workspace -> the keys run as length classes (stage0_path 6), rows identical to the oracle's"""
    from theia_amd import TadEngine
    K = 3000
    k, t, v = skewed_table(K, long_keys=2, long_len=20000, seed=11)
    eng = TadEngine(device=0, workspace_limit=256 << 20)          # rank grid: 3000 x 20000 cells x 17 B = 1 GB
    try:
        res, want = check_classes(eng, algo, k, t, v, K, agg)
        print("%s: %d rows, %d points in length classes, %.2f ms" % (algo, k.size, want["n_points"], res.stats["ms_total"]))
        # Stage 0 alone needs no grid at all: the sorted unique points are the result (stage0_path 7)
        pts = eng.aggregate(k, t, v, K, agg_flow=agg)
        pk, pt, pv = orc.stage0(k, t, v, "max" if agg == "" else "sum")
        assert pts.stats["stage0_path"] == 7 and pts.n_points == pk.size == pts.stats["n_points"]
        assert (pts["key_id"] == pk).all() and (pts["flow_end_s"] == pt).all() and (pts["value"] == pv).all()
        assert pts.stats["n_keys"] == np.unique(pk).size and pts.stats["rows_used"] == k.size
        gm = orc.u64_to_f64(pv).mean()
        assert abs(pts.stats["pts_mean"] - gm) <= 1e-12 * gm
    finally:
        eng.close()


def test_length_classes_forced_small_tables(engine):
    # every class boundary (16 / 64 / 256 points), keys without points, ARIMA's no-result keys, a single class
    with engine.plan(sparse="always", sparse_classes="always"):
        rng = np.random.default_rng(3)
        n_k = np.array([0, 1, 2, 3, 4, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300, 0, 5, 40], dtype=np.int64)
        K = n_k.size
        pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
        pt = np.concatenate([np.sort(rng.choice(5000, size=n, replace=False)) for n in n_k]).astype(np.int64) + 1660202814
        v = (2_000_000_000 + rng.integers(-3_000_000, 3_000_000, size=pk.size)).astype(np.uint64)
        v = np.where(rng.random(pk.size) < 0.02, v * np.uint64(5), v)
        order = rng.permutation(pk.size)
        k, t, v = pk[order], pt[order], v[order]
        for algo, agg in (("EWMA", "svc"), ("DBSCAN", ""), ("ARIMA", "svc")):
            check_classes(engine, algo, k, t, v, K, agg)
        k1, t1, v1 = day_table(40, 10, 2, seed=9, span=3000)      # all keys in one class
        check_classes(engine, "EWMA", k1, t1, v1, 40, "svc")


def test_grids_that_do_not_fit_while_the_points_do():
    """1000 rows over 100 000 keys x 250 buckets under a 1 MB workspace: neither the dense grid (225 MB) nor the rank grid fits,
    the points do -> length classes; 200 000 rows over 1000 keys: the one class the keys form does not fit either -> refused"""
    from theia_amd import TadEngine
    small = TadEngine(device=0, workspace_limit=1 << 20)
    try:
        k, t, v = orc.synth_rows(0, 1000, 100000, 250)
        res = small.run("EWMA", k, t, v, 100000, agg_flow="svc")
        want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
        assert res.stats["stage0_path"] == 6 and res.n_rows == want["n_anomalies"] and res.stats["n_points"] == want["n_points"]
        for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
            assert (res[f] == want[f]).all(), f
        with pytest.raises(TadError) as ei:
            small.run("EWMA", *orc.synth_rows(0, 200000, 1000, 250), 1000)
        assert ei.value.code == -6
    finally:
        small.close()
