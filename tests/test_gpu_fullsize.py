"""GPU: per-point parity at the BENCHMARKED sizes (BASELINE.json configs[1..4]; VERDICT r1 "next" #2, r4 "next" #1).

The Stage-0 plan the bench line runs (C2: 782 partitions of 128 keys, 128-byte lines, two parallel bucket rounds) is
only ever chosen at full size, so here the numpy oracle runs on the COMPLETE 1e8-row tables and every point, sigma,
EWMA value, DBSCAN / EWMA / ARIMA verdict is compared bit for bit:
  C2  EWMA,   1e8 rows / 1e5 keys / 250 buckets, sum   — all 2.45e7 points
  C3  ARIMA,  the same table                            — every prediction and verdict of 200 sampled keys (~4.9e4 fits)
  C4  DBSCAN, 1e8 rows / 1e6 keys / 100 buckets, max   — all points
  C5  EWMA + ARIMA on 1e9 rows / 1e6 keys / 250 buckets (BASELINE configs[4]):
        the shard one of 8 GPUs gets (`bench.py --config c5 --gpus 8`: 1.25e8 rows / 1.25e5 keys) — every point (EWMA) and 200
        sampled keys (ARIMA), exactly like C2 / C3;
        the whole table on ONE GPU (the strong-scaling base, `other_configs.c5`) — size-independent properties (sortedness, checksum
        of checksums against the host generator, idempotence) + the oracle on 200 sampled keys of the full table, both detectors.
The oracle side uses oracle.tad_oracle.stage0_dense / dbscan_noise_all (a different route to the same semantics as
stage0 / dbscan_noise_1d, checked against them in tests/test_oracle.py) so that the whole module runs in a few minutes."""
import numpy as np
import pytest

from oracle import arima_oracle as ao
from oracle import tad_oracle as orc

pytestmark = pytest.mark.gpu


def oracle_points(N, K, T, op):
    k, t, v = orc.synth_rows_parallel(N, K, T)
    pk, pt, pv = orc.stage0_dense(k, t, v, op, K, orc.SYNTH_T_BASE, orc.SYNTH_T_STEP, T)
    return (k, t, v), (pk, pt, pv)


def device_table(engine, rows, N, K, T):
    """the device-resident table the bench uses (tad_synth_generate) — and it must be the oracle's table"""
    dk, dt, dv = engine.synth(0, N, K, T)
    for d, h in zip((dk, dt, dv), rows):
        assert (d.to_host() == h).all()
    return dk, dt, dv


SHAPES = {"c2": (100_000_000, 100_000, 250),
          "c5_shard_of_8": (125_000_000, 125_000, 250)}      # rank 0's table of `bench.py --config c5 --gpus 8` (rows r*n .. (r+1)*n, local keys)


@pytest.fixture(scope="module", params=list(SHAPES))
def c2(engine, request):
    N, K, T = SHAPES[request.param]
    rows, pts = oracle_points(N, K, T, "sum")
    dev = device_table(engine, rows, N, K, T)
    del rows
    yield dict(N=N, K=K, T=T, dev=dev, pts=pts)
    for d in dev:
        d.free()


def check_points(h, pk, pt, pv, sigma, ptr):
    assert h["key_id"].size == pk.size
    assert (h["key_id"] == pk).all() and (h["flow_end_s"] == pt).all()
    assert (h["throughput"] == orc.u64_to_f64(pv)).all()                 # integer aggregates, bit-exact
    assert (h["stddev"] == np.repeat(sigma, np.diff(ptr))).all()          # stddev_samp per key, bit-exact


def test_c2_ewma_every_point(engine, c2):
    pk, pt, pv = c2["pts"]
    dk, dt, dv = c2["dev"]
    res = engine.run("EWMA", dk, dt, dv, c2["K"], agg_flow="svc", emit_all=True, out="device")
    st = res.stats
    assert st["stage0_path"] == 3                                         # the write-combining partition pass of the bench line
    assert st["rows_used"] == c2["N"] and st["n_keys"] == c2["K"] and st["n_points"] == pk.size == res.n_rows
    h = res.to_host()
    keys, ptr = orc.series_offsets(pk)
    pvf = orc.u64_to_f64(pv)
    sigma, has = orc.stddev_samp_all(pvf, ptr)
    check_points(h, pk, pt, pv, sigma, ptr)
    calc = orc.ewma_all(pvf, ptr)
    assert (h["algo_calc"] == calc).all()                                 # EWMA, bit-exact
    verdict = np.repeat(has, np.diff(ptr)) & (np.abs(pvf - calc) > np.repeat(sigma, np.diff(ptr)))
    assert (h["anomaly"].astype(bool) == verdict).all()
    # the filtered job (what the bench times) emits exactly the flagged rows
    res2 = engine.run("EWMA", dk, dt, dv, c2["K"], agg_flow="svc", out="device")
    assert res2.stats["stage0_path"] == 3 and res2.n_rows == int(verdict.sum())
    h2 = res2.to_host()
    for f, want in (("key_id", pk), ("flow_end_s", pt), ("throughput", pvf), ("algo_calc", calc)):
        assert (h2[f] == want[verdict]).all(), f


def test_c3_arima_sampled_keys_of_the_full_table(engine, c2):
    pk, pt, pv = c2["pts"]
    dk, dt, dv = c2["dev"]
    res = engine.run("ARIMA", dk, dt, dv, c2["K"], agg_flow="svc", emit_all=True, out="device")
    st = res.stats
    assert st["n_points"] == pk.size and st["arima_fits"] == pk.size - 3 * c2["K"] and st["keys_no_result"] == 0
    h = res.to_host()
    assert (h["key_id"] == pk).all() and (h["flow_end_s"] == pt).all()
    keys, ptr = orc.series_offsets(pk)
    pvf = orc.u64_to_f64(pv)
    rng = np.random.default_rng(3)
    sample = np.sort(rng.choice(c2["K"], size=200, replace=False))
    checked = 0
    for kk in sample:
        a, b = ptr[kk], ptr[kk + 1]
        want = np.array(ao.calculate_arima_exact(pv[a:b]))
        got = h["algo_calc"][a:b]
        fin = np.isfinite(want)
        assert (np.isfinite(got) == fin).all()
        assert (got[fin].view(np.uint64) == want[fin].view(np.uint64)).all(), int(kk)      # every prediction, bit for bit
        sd = orc.stddev_samp_series(pvf[a:b])
        assert sd == h["stddev"][a]
        with np.errstate(invalid="ignore"):
            assert (h["anomaly"][a:b].astype(bool) == (np.abs(pvf[a:b] - want) > sd)).all()       # zero verdict flips
        checked += b - a
    assert checked > 45000


def test_c4_dbscan_every_point(engine):
    N, K, T = 100_000_000, 1_000_000, 100
    rows, (pk, pt, pv) = oracle_points(N, K, T, "max")
    dk, dt, dv = device_table(engine, rows, N, K, T)
    del rows
    res = engine.run("DBSCAN", dk, dt, dv, K, agg_flow="", emit_all=True, out="device")      # mode None: max(throughput)
    st = res.stats
    assert st["stage0_path"] in (2, 3)
    assert st["rows_used"] == N and st["n_keys"] == K and st["n_points"] == pk.size == res.n_rows
    h = res.to_host()
    keys, ptr = orc.series_offsets(pk)
    pvf = orc.u64_to_f64(pv)
    sigma, has = orc.stddev_samp_all(pvf, ptr)
    check_points(h, pk, pt, pv, sigma, ptr)
    assert (h["algo_calc"] == 0.0).all()
    noise = orc.dbscan_noise_all(pvf, ptr)
    assert (h["anomaly"].astype(bool) == noise).all()
    res2 = engine.run("DBSCAN", dk, dt, dv, K, agg_flow="", out="device")
    assert res2.n_rows == int(noise.sum())
    h2 = res2.to_host()
    assert (h2["key_id"] == pk[noise]).all() and (h2["flow_end_s"] == pt[noise]).all() and (h2["throughput"] == pvf[noise]).all()


# ------------------------------------------------------------------ C5 on ONE GPU: BASELINE configs[4]'s table, the strong-scaling base
def _c5_chunk(args):
    """host generator over one chunk of the table: the u64 sum of the values and the rows of the sampled keys"""
    first, n, K, T, sample = args
    k, t, v = orc.synth_rows(first, n, K, T)
    sel = np.isin(k, sample)
    return int(v.sum(dtype=np.uint64)), k[sel], t[sel], v[sel]


def test_c5_whole_table_on_one_gpu(engine):
    import multiprocessing as mp
    import os
    N, K, T = 1_000_000_000, 1_000_000, 250
    dk, dt, dv = engine.synth(0, N, K, T)
    # the device table IS the host generator's (three 5e6-row windows: first, middle, last)
    for first in (0, N // 2 - 2_500_000, N - 5_000_000):
        want = orc.synth_rows(first, 5_000_000, K, T)
        got = engine.synth(first, 5_000_000, K, T)
        for g, w in zip(got, want):
            assert (g.to_host() == w).all()
            g.free()
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(K, size=200, replace=False)).astype(np.uint64)
    chunk = 5_000_000
    with mp.get_context("fork").Pool(min(os.cpu_count() or 1, 128)) as pool:
        parts = pool.map(_c5_chunk, [(i, min(chunk, N - i), K, T, sample) for i in range(0, N, chunk)], chunksize=1)
    total = sum(p[0] for p in parts) % 2**64
    sk, st_, sv = (np.concatenate([p[i] for p in parts]) for i in (1, 2, 3))
    pk, pt, pv = orc.stage0(sk, st_, sv, "sum")                                  # the sampled keys' points, by the plain (sort-based) oracle
    keys, ptr = orc.series_offsets(pk)
    assert (keys == sample).all()
    pvf = orc.u64_to_f64(pv)

    res = engine.run("EWMA", dk, dt, dv, K, agg_flow="svc", emit_all=True, out="device")
    st = res.stats
    print("C5 on one GPU: stage0_path %d, %d points" % (st["stage0_path"], st["n_points"]))
    assert st["stage0_path"] in (2, 3)                                           # a partition pass (sort-by-tile or write-combining), never the direct scatter
    assert st["rows_used"] == N and st["n_keys"] == K and st["step"] == 60 and st["n_buckets"] == T
    assert st["n_points"] == res.n_rows <= K * T
    h = res.to_host()
    res.close()
    # (key, time) strictly increasing = sorted and duplicate-free
    dkey = np.diff(h["key_id"].astype(np.int64))
    assert ((dkey > 0) | ((dkey == 0) & (np.diff(h["flow_end_s"]) > 0))).all()
    del dkey
    # checksum of checksums: sum over all points of sum(throughput) == sum over all rows (values < 2^40: every aggregate exact in f64)
    assert int(h["throughput"].astype(np.uint64).sum(dtype=np.uint64)) == total
    verdict = np.abs(h["throughput"] - h["algo_calc"]) > h["stddev"]
    assert (verdict == h["anomaly"].astype(bool)).all()
    # the oracle on the 200 sampled keys: aggregates, sigma, EWMA, verdicts, bit for bit
    sel = np.isin(h["key_id"], sample)
    assert (h["key_id"][sel] == pk).all() and (h["flow_end_s"][sel] == pt).all() and (h["throughput"][sel] == pvf).all()
    sigma, has = orc.stddev_samp_all(pvf, ptr)
    assert (h["stddev"][sel] == np.repeat(sigma, np.diff(ptr))).all()
    calc = orc.ewma_all(pvf, ptr)
    assert (h["algo_calc"][sel] == calc).all()
    assert (verdict[sel] == (np.repeat(has, np.diff(ptr)) & (np.abs(pvf - calc) > np.repeat(sigma, np.diff(ptr))))).all()
    # the filtered job (what the bench times) emits exactly the flagged rows; run twice: idempotent
    for _ in range(2):
        res2 = engine.run("EWMA", dk, dt, dv, K, agg_flow="svc", out="device")
        assert res2.n_rows == int(verdict.sum()) == res2.stats["n_anomalies"]
        h2 = res2.to_host()
        res2.close()
        for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
            assert (h2[f] == h[f][verdict]).all(), f
    del h2, verdict
    n_points = h["key_id"].size
    del h
    # ARIMA on the same table: every prediction and verdict of the sampled keys
    res = engine.run("ARIMA", dk, dt, dv, K, agg_flow="svc", emit_all=True, out="device")
    st = res.stats
    assert st["n_points"] == n_points and st["arima_fits"] == n_points - 3 * K and st["keys_no_result"] == 0
    h = res.to_host()
    res.close()
    sel = np.isin(h["key_id"], sample)
    got_all, an_all, sd_all = h["algo_calc"][sel], h["anomaly"][sel].astype(bool), h["stddev"][sel]
    assert (h["key_id"][sel] == pk).all() and (h["flow_end_s"][sel] == pt).all()
    del h
    for i in range(sample.size):
        a, b = ptr[i], ptr[i + 1]
        want = np.array(ao.calculate_arima_exact(pv[a:b]))
        got = got_all[a:b]
        fin = np.isfinite(want)
        assert (np.isfinite(got) == fin).all()
        assert (got[fin].view(np.uint64) == want[fin].view(np.uint64)).all(), int(sample[i])
        assert sd_all[a] == sigma[i]
        with np.errstate(invalid="ignore"):
            assert (an_all[a:b] == (np.abs(pvf[a:b] - want) > sigma[i])).all()
    for d in (dk, dt, dv):
        d.free()


def test_full_size_first_jobs_go_through_on_the_sampled_histogram():
    """Every job of a controller is the FIRST on its table (controller.go:499-523), so what the sampled pass A costs when it is wrong is paid by
    every job it is wrong for.  On the C2 table — hashed, with an `--end-time`, with `--start-time` + `--end-time` (regions of ~300 records: the
    low tail of the sampled count, where `sampled_capacity`'s constant decides) — and on a time-ordered table whose keys are alive for a tenth
    of it (keys that LEAVE while a workgroup's chunk is read: the estimate must not be biased against them), a fresh engine's first job
    settles on the sample in ONE attempt; rows sorted by key are sent to the exact histogram (two attempts, the first one cut short)."""
    import torch
    from theia_amd import TadEngine
    N, K, T = SHAPES["c2"]
    dev = torch.device("cuda", 0)
    key = torch.empty(N, dtype=torch.int64, device=dev)
    tend = torch.empty(N, dtype=torch.int64, device=dev)
    val = torch.empty(N, dtype=torch.int64, device=dev)
    eng = TadEngine(device=0)
    eng.synth(0, N, K, T, into=(key, tend, val))
    eng.close()
    lo, hi = int(tend.min()), int(tend.max())
    i = torch.arange(N, device=dev)
    W = K // 10
    k_live = ((i.double() * ((K - W) / N)).long() + (key * 2654435761 % W)) % K
    t_live = lo + 60 * ((i.double() * (T / N)).long())
    del i
    tstart = tend - 30
    by_key = torch.sort(key, stable=True).indices
    cases = [("hashed", (key, tend, val), {}, (1, 1)),
             ("end_time", (key, tend, val), dict(end_time=lo + (hi - lo) * 4 // 5), (1, 1)),
             ("start_and_end_time", (key, tend, val), dict(flow_start_s=tstart, start_time=lo + (hi - lo) // 5, end_time=lo + (hi - lo) * 4 // 5), (1, 1)),
             ("keys_alive_for_a_tenth_in_time_order", (k_live, t_live, val), {}, (1, 1)),
             ("rows_by_key", (key[by_key].contiguous(), tend[by_key].contiguous(), val[by_key].contiguous()), {}, (2, 0))]
    torch.cuda.synchronize()      # (the columns were written on torch's stream, the engine reads them on its own)
    rows = {}
    for name, (k, t, v), kw, want in cases:
        eng = TadEngine(device=0)
        try:
            r = eng.run("EWMA", k, t, v, K, agg_flow="svc", out="device", **kw)
            assert (r.stats["stage0_attempts"], r.stats["hist_sampled"]) == want, (name, r.stats["stage0_attempts"], r.stats["hist_sampled"])
            rows[name] = r.n_rows
            r.close()
        finally:
            eng.close()
    assert rows["rows_by_key"] == rows["hashed"]        # (the same rows in another order: the same anomalies)
