"""Parity of the opt-in experiments (env switches, default off, queued for measurement on the GPU) on the host emulator:
   TAD_META_PREFETCH=1      software-pipelined sampled histogram in pass A
   TAD_DBSCAN_TILESTATS=1   pass C leaves per-round key statistics, the DBSCAN scan settles keys from them
   TAD_DBSCAN_WAVELIST=1    exact DBSCAN pair tests with one wavefront per listed key (readlane broadcast, no LDS / barriers) and
                            the emit from the same work list (k_emit_dbscan_wave)
   TAD_ARIMA_FILTER=collapsed   ARIMA likelihood by the collapsed recursion (oracle switched with the same variable)
   TAD_EWMA_FUSED=1         EWMA job: sigma + detector + compaction + emit in one kernel with a decoupled look-back (k_ewma_fused)
Run:  python tools/hipemu/build.py && python tools/hipemu/check_experiments.py
Every case runs the whole job through the C ABI of the emulated library and compares all rows with the oracle."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["TAD_LIBRARY_PATH"] = os.path.join(HERE, "_build", "libtad_hipemu.so")

import numpy as np  # noqa: E402

from oracle import tad_oracle as orc  # noqa: E402
from theia_amd import TadEngine  # noqa: E402

FIELDS = ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")


def run(eng, algo, k, t, v, K, agg, env):
    old = {name: os.environ.get(name) for name in env}
    os.environ.update(env)
    try:
        return eng.run(algo, k, t, v, K, agg_flow=agg)
    finally:
        for name, val in old.items():
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val


def check(eng, label, algo, k, t, v, K, agg, env):
    want = orc.run_job(algo, k, t, v, agg_flow=agg)
    base = run(eng, algo, k, t, v, K, agg, {kk_: vv for kk_, vv in env.items() if not kk_.startswith(("TAD_META_PREFETCH", "TAD_DBSCAN_TILESTATS", "TAD_DBSCAN_WAVELIST"))})
    res = run(eng, algo, k, t, v, K, agg, env)
    assert res.n_rows == want["n_anomalies"] == base.n_rows, (label, res.n_rows, want["n_anomalies"])
    for f in FIELDS:
        assert (res[f] == want[f]).all(), (label, f)
    for f in ("n_points", "n_keys", "rows_used"):
        assert res.stats[f] == base.stats[f], (label, f)
    assert abs(res.stats["pts_mean"] - base.stats["pts_mean"]) <= 1e-12 * abs(base.stats["pts_mean"]), label
    assert abs(res.stats["pts_m2"] - base.stats["pts_m2"]) <= 1e-12 * abs(base.stats["pts_m2"]), label
    print("ok  %-58s rows %6d  path %d  sampled %d" % (label, res.n_rows, res.stats["stage0_path"], res.stats["hist_sampled"]))


def check_fused(eng):
    """k_ewma_fused: capacity from the previous job, pinned capacities (exact / too small -> classic emit), a tiny LDS capacity
    (overflow walk), device and host results, ragged keys (none / one point), the sparse rank grid.  (Workgroups run in
    order on the emulator, so the look-back never has to wait here; what is checked is everything else.)"""
    F = {"TAD_EWMA_FUSED": "1"}

    def one(label, k, t, v, K, env, want_path, out="host"):
        want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
        old = {name: os.environ.get(name) for name in env}
        os.environ.update(env)
        try:
            res = eng.run("EWMA", k, t, v, K, agg_flow="svc", out=out)
        finally:
            for name, val in old.items():
                if val is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = val
        h = res.to_host() if out == "device" else res
        assert res.n_rows == want["n_anomalies"], (label, res.n_rows, want["n_anomalies"])
        for f in FIELDS:
            assert (h[f] == want[f]).all(), (label, f)
        assert res.stats["n_keys"] == want["n_keys"] and res.stats["n_points"] == want["n_points"], label
        assert want_path is None or res.stats["detect_path"] == want_path, (label, res.stats["detect_path"])
        print("ok  fused     %-58s rows %6d  detect_path %d" % (label, res.n_rows, res.stats["detect_path"]))
        return res

    for n, K, T in ((3000, 40, 13), (100003, 100, 250), (60000, 300, 100), (5000, 65, 7)):
        k, t, v = orc.synth_rows(0, n, K, T)
        rows = one("%d rows / %d keys / %d buckets, first job" % (n, K, T), k, t, v, K, F, None).n_rows
        one("capacity from the previous job", k, t, v, K, F, 1)
        one("device result", k, t, v, K, F, 1, out="device")
        one("result block exactly the rows", k, t, v, K, dict(F, TAD_EWMA_FUSED_ROWS=str(max(rows, 1))), 1)
        one("result block too small -> classic emit", k, t, v, K, dict(F, TAD_EWMA_FUSED_ROWS=str(max(rows // 2, 1))), 0)
        one("tiny LDS capacity (overflow walk)", k, t, v, K, dict(F, TAD_EMIT_CAP=str(T + 1)), 1)
        one("Stage 0 v2", k, t, v, K, dict(F, TAD_STAGE0="v2"), 1)
    rng = np.random.default_rng(5)
    K = 150
    n_k = rng.integers(0, 40, size=K)
    n_k[::7] = 0
    n_k[3::11] = 1
    pk = np.repeat(np.arange(K, dtype=np.uint64), n_k)
    pt = np.concatenate([np.sort(rng.choice(60, size=n, replace=False)) for n in n_k]).astype(np.int64) * 60 + 1660202814
    v = (1_000_000_000 * np.exp(rng.normal(0, 0.8, size=pk.size))).astype(np.uint64)
    o = rng.permutation(pk.size)
    k, t, v = pk[o], pt[o], v[o]
    one("ragged noisy table", k, t, v, K, F, 1)
    one("ragged noisy table, tiny LDS capacity", k, t, v, K, dict(F, TAD_EMIT_CAP="64"), 1)
    one("sparse rank grid (timestamps from the grid)", k, t, v, K, dict(F, TAD_SPARSE="1"), 1)


def main():
    eng = TadEngine(device=0)
    rng = np.random.default_rng(17)
    v2 = {"TAD_STAGE0": "v2"}
    for partb in ("wc", "sort"):
        e = dict(v2, TAD_PARTB=partb)
        k, t, v = orc.synth_rows(0, 100003, 100, 250)
        check(eng, "prefetch  %s 1e5 rows / 100 keys (sampled histogram)" % partb, "EWMA", k, t, v, 100, "svc", dict(e, TAD_META_PREFETCH="1"))
        check(eng, "tilestats %s one round" % partb, "DBSCAN", k, t, v, 100, "svc", dict(e, TAD_DBSCAN_TILESTATS="1"))
        k, t, v = orc.synth_rows(0, 120000, 300, 250)
        check(eng, "tilestats %s two rounds (128-key blocks x 250 buckets)" % partb, "DBSCAN", k, t, v, 300, "svc",
              dict(e, TAD_DBSCAN_TILESTATS="1", TAD_KP_SHIFT_MIN="7"))
        check(eng, "tilestats %s max mode, three rounds" % partb, "DBSCAN", k, t, v, 300, "", dict(e, TAD_DBSCAN_TILESTATS="1", TAD_KP_SHIFT_MIN="8"))
        v2_ = v.copy()
        v2_[::997] = rng.integers(2**50, 2**63, size=v2_[::997].size, dtype=np.uint64)    # overflow-list records
        check(eng, "tilestats %s values beyond the packed record range" % partb, "DBSCAN", k, t, v2_, 300, "svc",
              dict(e, TAD_DBSCAN_TILESTATS="1", TAD_KP_SHIFT_MIN="7"))
    # TILESTATS=2: with one bucket round per partition pass C does not write the columns of the keys it sees are settled (fresh
    # "device" memory is 0xA5 here: a reader of such a column would show)
    for partb in ("wc", "sort"):
        e = dict(v2, TAD_PARTB=partb, TAD_DBSCAN_TILESTATS="2")
        k, t, v = orc.synth_rows(0, 100003, 100, 250)
        check(eng, "tilestats=2 %s one round: settled columns not written" % partb, "DBSCAN", k, t, v, 100, "svc", e)
        check(eng, "tilestats=2 %s + wavelist, max mode" % partb, "DBSCAN", k, t, v, 100, "", dict(e, TAD_DBSCAN_WAVELIST="1"))
        k, t, v = orc.synth_rows(0, 120000, 300, 250)
        check(eng, "tilestats=2 %s two rounds (columns written as before)" % partb, "DBSCAN", k, t, v, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="7"))
        k, t, v = orc.synth_rows(0, 60000, 300, 100)
        v3 = v.copy()
        v3[::997] = rng.integers(2**50, 2**63, size=v3[::997].size, dtype=np.uint64)
        check(eng, "tilestats=2 %s overflow-list records (skipping off for the job)" % partb, "DBSCAN", k, t, v3, 300, "svc", e)
        kk = np.where(k % np.uint64(5) == 0, k, np.where(rng.random(k.size) < 0.97, orc.KEY_SKIP, k))   # most keys with 0..3 points
        check(eng, "tilestats=2 %s keys with fewer than min_samples points" % partb, "DBSCAN", kk, t, v, 300, "svc", e)
    # ... and a partition that needs several LDS tiles is then split by KEY sub-range (whole series per tile) instead of by buckets
    for partb in ("wc", "sort"):
        e = dict(v2, TAD_PARTB=partb, TAD_DBSCAN_TILESTATS="2")
        k, t, v = orc.synth_rows(0, 120000, 300, 250)
        check(eng, "key rounds %s: 128-key blocks x 250 buckets -> 2 x 64 keys" % partb, "DBSCAN", k, t, v, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="7"))
        check(eng, "key rounds %s: 256-key blocks, max mode -> 4 x 64 keys" % partb, "DBSCAN", k, t, v, 300, "", dict(e, TAD_KP_SHIFT_MIN="8"))
        check(eng, "key rounds %s + wavelist" % partb, "DBSCAN", k, t, v, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="8", TAD_DBSCAN_WAVELIST="1"))
        check(eng, "key rounds %s, rounds one after the other" % partb, "DBSCAN", k, t, v, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="8", TAD_PAR_ROUNDS="0"))
        v3 = v.copy()
        v3[::997] = rng.integers(2**50, 2**63, size=v3[::997].size, dtype=np.uint64)
        check(eng, "key rounds %s overflow-list records" % partb, "DBSCAN", k, t, v3, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="8"))
        kk = np.where(k % np.uint64(5) == 0, k, np.where(rng.random(k.size) < 0.97, orc.KEY_SKIP, k))
        check(eng, "key rounds %s keys with fewer than min_samples points" % partb, "DBSCAN", kk, t, v, 300, "svc", dict(e, TAD_KP_SHIFT_MIN="8"))
    k, t, v = orc.synth_rows(0, 400000, 300, 64)
    k = np.where(rng.random(k.size) < 0.5, np.uint64(7), k)
    check(eng, "key rounds hot key (split partition)", "DBSCAN", k, t, v, 300, "svc", dict(v2, TAD_DBSCAN_TILESTATS="2", TAD_HIST_SAMPLE="0", TAD_KP_SHIFT_MIN="9"))
    k, t, v = orc.synth_rows(0, 300000, 250000, 100)        # many keys -> two-level plan, single-round 128-key tiles
    v2 = dict(v2, TAD_SPARSE="0")                           # (so few rows per key would otherwise take the sparse path)
    res_tl = run(eng, "DBSCAN", k, t, v, 250000, "", dict(v2, TAD_TWO_LEVEL="1"))
    if res_tl.stats["stage0_path"] == 5:
        check(eng, "tilestats=2 two-level plan, 250000 keys", "DBSCAN", k, t, v, 250000, "", dict(v2, TAD_TWO_LEVEL="1", TAD_DBSCAN_TILESTATS="2"))
        check(eng, "tilestats=1 two-level plan, 250000 keys", "DBSCAN", k, t, v, 250000, "", dict(v2, TAD_TWO_LEVEL="1", TAD_DBSCAN_TILESTATS="1"))
    else:
        print("--  two-level plan not taken at this size (path %d)" % res_tl.stats["stage0_path"])
    v2 = {"TAD_STAGE0": "v2"}
    # a hot key: its partition is split over several workgroups (partial tiles -> the detector walks the grid for its keys)
    k, t, v = orc.synth_rows(0, 400000, 300, 64)
    k = np.where(rng.random(k.size) < 0.5, np.uint64(7), k)
    check(eng, "tilestats hot key (split partition)", "DBSCAN", k, t, v, 300, "svc", dict(v2, TAD_DBSCAN_TILESTATS="1", TAD_HIST_SAMPLE="0"))
    check(eng, "prefetch  hot key", "EWMA", k, t, v, 300, "svc", dict(v2, TAD_META_PREFETCH="1"))
    # wavefront-per-key pair tests: 1..4 buckets per lane, keys with fewer than min_samples points, spikes and dips
    for n, K, T in ((3000, 40, 13), (60000, 300, 100), (100003, 100, 250), (40000, 200, 190), (20000, 70, 64), (30000, 90, 129)):
        k, t, v = orc.synth_rows(0, n, K, T)
        check(eng, "wavelist  %d rows / %d keys / %d buckets" % (n, K, T), "DBSCAN", k, t, v, K, "svc", dict(v2, TAD_DBSCAN_WAVELIST="1"))
        check(eng, "wavelist + tilestats, max mode, same table", "DBSCAN", k, t, v, K, "", dict(v2, TAD_DBSCAN_WAVELIST="1", TAD_DBSCAN_TILESTATS="1"))
    gold = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "reference_golden.json")))
    os.environ["TAD_DBSCAN_WAVELIST"] = "1"
    assert eng.series_dbscan_anomaly(gold["throughput_list"]).tolist() == gold["expected_dbscan_anomaly_list"]
    x = [1000000000, 1250000000, 1500000000, 1750000001, 5000000000, 5000000001, 5000000002, 5250000002]   # exact-eps chains
    os.environ["TAD_DBSCAN_WAVELIST"] = "0"
    ref = eng.series_dbscan_anomaly(x).tolist()
    os.environ["TAD_DBSCAN_WAVELIST"] = "1"
    assert eng.series_dbscan_anomaly(x).tolist() == ref == orc.dbscan_noise_1d(orc.u64_to_f64(np.array(x, dtype=np.uint64))).tolist()
    os.environ.pop("TAD_DBSCAN_WAVELIST")
    print("ok  wavelist  reference golden series + exact-eps chain")
    # collapsed ARIMA filter: the four-chain fit kernel at 2 / 3 / 4 wavefronts per SIMD against oracle/arima_exact.c in the
    # same mode (bit for bit: predictions, verdicts, Kalman-step counter), and the default contract untouched next to it
    from oracle import arima_oracle as ao
    k, t, v = orc.synth_rows(0, 3000, 20, 40)
    for flt in ("general", "collapsed"):
        os.environ["TAD_ARIMA_FILTER"] = flt
        want = orc.run_job("ARIMA", k, t, v, agg_flow="svc")
        for waves in ((None,) if flt == "general" else ("2", "3", "4")):
            if waves: os.environ["TAD_ARIMA_WAVES"] = waves
            res = eng.run("ARIMA", k, t, v, 20, agg_flow="svc")
            os.environ.pop("TAD_ARIMA_WAVES", None)
            assert res.n_rows == want["n_anomalies"], (flt, waves)
            for f in FIELDS:
                a, b = res[f], want[f]
                assert (a.view(np.uint64) == b.view(np.uint64)).all() if a.dtype == np.float64 else (a == b).all(), (flt, waves, f)
            assert res.stats["kalman_steps"] == want["kalman_steps"], (flt, waves)
            print("ok  arima     %-9s filter, waves %-4s rows %5d  kalman steps %d" % (flt, waves, res.n_rows, want["kalman_steps"]))
        got = eng.series_arima(gold["throughput_list"])
        assert (np.asarray(got).view(np.uint64) == np.asarray(ao.calculate_arima_exact(gold["throughput_list"])).view(np.uint64)).all(), flt
        assert eng.series_arima_anomaly(gold["throughput_list"], gold["stddev"]).tolist() == gold["expected_anomaly_list_arima"], flt
    os.environ.pop("TAD_ARIMA_FILTER")
    print("ok  arima     reference golden series, both filters")
    check_fused(eng)
    print("all experiments agree with the oracle on the emulator")


if __name__ == "__main__":
    main()
