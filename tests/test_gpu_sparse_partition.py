"""GPU: the sparse Stage 0 of BIG tables — key-block partition pass + one LDS sort per key sub-range (tad_sparse.hip: launch_sparse_sort,
stage0_path 8 / 9 / 10) instead of the LSD radix sort (4 / 6 / 7).  Same contract as tests/test_gpu_sparse.py: the job's rows equal the
oracle's bit for bit (GROUP BY key, flowEndSeconds with max / wrapping sum: anomaly_detection.py:52-61, 507-614).  The engine takes this form
when pass A ran with its key-bin histogram (>= 2^22 rows); the small tables here force it with tad_plan (stage0 = v2, sparse_sort = partition).
Everything the 8-byte records cannot hold — a heavy key whose bin exceeds one LDS round, a value of 2^36 and more, a lattice too long for the
record's cell bits — must end in the LSD sort with the same rows."""

import numpy as np
import pytest

from oracle import tad_oracle as orc
from test_gpu_sparse import day_table, skewed_table

pytestmark = pytest.mark.gpu

PART = dict(stage0="v2", sparse="always", sparse_sort="partition")


def check(engine, algo, k, t, v, K, agg_flow, paths=(8,), **kw):
    want = orc.run_job(algo, k, t, v, agg_flow=agg_flow, **kw)
    allp = engine.run(algo, k, t, v, K, agg_flow=agg_flow, emit_all=True, **kw)
    assert allp.stats["stage0_path"] in paths, allp.stats["stage0_path"]
    pk, pt, pv = want["points"]
    assert allp.n_rows == pk.size
    assert (allp["key_id"] == pk).all() and (allp["flow_end_s"] == pt).all()
    assert (allp["throughput"] == orc.u64_to_f64(pv)).all()
    assert (allp["stddev"] == np.repeat(want["sigma"], np.diff(want["ptr"]))).all()
    assert np.array_equal(allp["algo_calc"], want["calc_all"], equal_nan=True)
    assert (allp["anomaly"].astype(bool) == want["anomaly_all"]).all()
    res = engine.run(algo, k, t, v, K, agg_flow=agg_flow, **kw)
    assert res.stats["stage0_path"] in paths and res.n_rows == want["n_anomalies"]
    for f in ("key_id", "flow_end_s", "throughput", "stddev"):
        assert (res[f] == want[f]).all(), f
    assert np.array_equal(res["algo_calc"], want["algo_calc"], equal_nan=True)
    assert res.stats["n_keys"] == want["n_keys"] and res.stats["n_points"] == want["n_points"]
    return res, want


@pytest.mark.parametrize("algo,agg", [("EWMA", "svc"), ("DBSCAN", "")])
def test_day_of_seconds_through_the_partition_pass(engine, algo, agg):
    K = 3000
    k, t, v = day_table(K, 20, 3, seed=21)                      # 1.8e5 rows, 86 400 buckets: 750 key blocks of 4 keys
    with engine.plan(**PART):
        res, want = check(engine, algo, k, t, v, K, agg)
        assert res.stats["step"] == 1 and res.stats["n_buckets"] > 86000 and res.stats["stage0_attempts"] == 1


def test_key_blocks_of_several_rounds(engine):
    """key blocks whose records do not fit one LDS round (14 336 records): the block's bins are split into rounds at bin boundaries,
    every round stages its points behind the earlier ones; blocks without rows, a block with one key only"""
    rng = np.random.default_rng(22)
    K = 4000                                                     # 1000 blocks of 4 keys (bins of one key)
    n_k = rng.integers(0, 6, size=K)
    n_k[8:16] = 5000                                             # two blocks of 4 x 5000 points x ~1.5 rows = 30 000 records each: 3 rounds
    n_k[100] = 9000                                              # one key alone fills most of a round
    n_k[3996:] = 0
    pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
    pt = np.concatenate([np.sort(rng.choice(40000, size=n, replace=False)) for n in n_k]).astype(np.int64) + 1660202814
    v = (2_000_000_000 + rng.integers(-3_000_000, 3_000_000, size=pk.size)).astype(np.uint64)
    v = np.where(rng.random(pk.size) < 0.01, v * np.uint64(6), v)
    dup = rng.random(pk.size) < 0.5
    k = np.concatenate([pk, pk[dup]]); t = np.concatenate([pt, pt[dup]]); v = np.concatenate([v, v[dup] // np.uint64(3)])
    order = rng.permutation(k.size)
    k, t, v = k[order], t[order], v[order]
    with engine.plan(**PART):
        check(engine, "EWMA", k, t, v, K, "svc")
        check(engine, "DBSCAN", k, t, v, K, "")
        pts = engine.aggregate(k, t, v, K, agg_flow="svc")
        qk, qt, qv = orc.stage0(k, t, v, "sum")
        assert pts.stats["stage0_path"] in (8, 10) and pts.n_points == qk.size
        assert (pts["key_id"] == qk).all() and (pts["flow_end_s"] == qt).all() and (pts["value"] == qv).all()


def test_second_key_time_window_skipped_keys_and_a_coarse_lattice(engine):
    k, t, v = day_table(2500, 12, 2, seed=23, span=7200)
    t = 1660202814 + (t - 1660202814) * 7                        # a lattice of 7-second steps: bucket * step is the time that comes out
    rng = np.random.default_rng(24)
    k2 = (orc.mix64(k + np.uint64(99)) % np.uint64(2500)).astype(np.uint64)
    k2 = np.where(k2 % np.uint64(5) == 0, orc.KEY_SKIP, k2)
    k = np.where(k % np.uint64(7) == 0, orc.KEY_SKIP, k)
    ts = t - rng.integers(0, 600, size=t.size)
    with engine.plan(**PART):
        res, _ = check(engine, "EWMA", k, t, v, 2500, "pod", key_id2=k2, flow_start_s=ts, start_time=int(t.min()) + 700, end_time=int(t.max()) - 2100)
        assert res.stats["step"] == 7
        pts = engine.aggregate(k, t, v, 2500, agg_flow="pod", key_id2=k2)
        pk, pt, pv = orc.stage0(k, t, v, "sum", k2)
        assert pts.stats["stage0_path"] in (8, 10) and pts.n_points == pk.size
        assert (pts["key_id"] == pk).all() and (pts["flow_end_s"] == pt).all() and (pts["value"] == pv).all()


def test_lattice_of_exactly_a_power_of_two_buckets(engine):
    # T = 4096: the largest cell of a block is T * KP - 1; the record's all-ones cell is reserved for the fillers
    rng = np.random.default_rng(25)
    K = 2000
    pk = np.repeat(np.arange(K, dtype=np.uint64), 8)
    pt = rng.integers(0, 4096, size=pk.size).astype(np.int64)
    pt[0], pt[-1] = 0, 4095
    pk[-1] = K - 1
    v = rng.integers(1, 3_000_000_000, size=pk.size).astype(np.uint64)
    t = 1660202814 + pt
    with engine.plan(**PART):
        res, _ = check(engine, "DBSCAN", pk, t, v, K, "")
        assert res.stats["n_buckets"] == 4096


def test_what_the_records_cannot_hold_goes_to_the_lsd_sort(engine):
    # (a) a key with more records than one LDS round holds; (b) values of 2^36 and more (36 value bits at 28 cell bits; here fewer cell bits,
    # so take 2^62); (c) a lattice whose bucket index needs more bits than the record's cell has room for
    K = 3000
    k, t, v = skewed_table(K, long_keys=1, long_len=20000, seed=26)
    with engine.plan(**PART):
        res, _ = check(engine, "EWMA", k, t, v, K, "svc", paths=(4,))
        assert res.stats["stage0_attempts"] == 2
        k, t, v = day_table(K, 10, 2, seed=27)
        big = np.where(np.arange(v.size) % 1000 == 0, np.uint64(2**62) + v, v)
        res, _ = check(engine, "EWMA", k, t, big, K, "svc", paths=(4,))
        assert res.stats["stage0_attempts"] == 2
        t_long = 1660202814 + (t - 1660202814) * 3001 + (np.arange(t.size) % 2)       # gcd 1 over 2.6e8 seconds: 28 time bits alone
        check(engine, "EWMA", k, t_long, v, K, "svc", paths=(4,))


def test_length_classes_after_the_partition_sort():
    from theia_amd import TadEngine
    K = 3000
    k, t, v = skewed_table(K, long_keys=2, long_len=6000, seed=28)     # 6000 points x 1.2 rows: one bin still fits a round
    eng = TadEngine(device=0, workspace_limit=128 << 20, plan=PART)   # rank grid: 3000 x 6000 cells x 17 B = 306 MB
    try:
        want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
        res = eng.run("EWMA", k, t, v, K, agg_flow="svc")
        assert res.stats["stage0_path"] == 9 and res.n_rows == want["n_anomalies"]
        for f in ("key_id", "flow_end_s", "throughput", "stddev", "algo_calc"):
            assert (res[f] == want[f]).all(), f
        pts = eng.aggregate(k, t, v, K, agg_flow="svc")
        pk, pt, pv = orc.stage0(k, t, v, "sum")
        assert pts.stats["stage0_path"] == 10 and pts.n_points == pk.size
        assert (pts["key_id"] == pk).all() and (pts["flow_end_s"] == pt).all() and (pts["value"] == pv).all()
    finally:
        eng.close()


def test_big_table_takes_the_partition_sort_by_itself(engine):
    """5e6 rows of per-connection keys on second-resolution timestamps: pass A runs (>= 2^22 rows), the table is sparse, no plan override"""
    K = 100_000
    k, t, v = day_table(K, 17, 3, seed=29)
    res, want = check(engine, "DBSCAN", k, t, v, K, "")
    print("5.1e6 rows, %d points: job %.2f ms, stage 0 %.2f ms" % (want["n_points"], res.stats["ms_total"], res.stats["ms_stage0"]))
