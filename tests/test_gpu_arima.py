"""GPU parity tests for the ARIMA detector: HIP path through the C ABI vs the oracle and the reference's golden
verdicts.

Contract: the device code (theia_amd/csrc/tad_arima.hip) and the checker (oracle/arima_exact.c) follow
ONE arithmetic contract — IEEE double +, -, *, /, sqrt in a fixed order, no FMA contraction, transcendental functions from
the shared deterministic source tad_detmath.h — so the L-BFGS-B trajectories coincide and every prediction is compared
at BASELINE.json's 1e-6 relative tolerance ON EVERY POINT, with zero verdict flips; in fact the results are bit-identical
and that is asserted too.  Against the reference itself (statsmodels 0.14.0, absent) parity is pinned by its golden verdict
list and the five leading characters of its golden predictions (anomaly_detection_test.py:261-283, 320-345)."""
TOL = 1e-6   # BASELINE.json north_star: "EWMA/ARIMA scores within 1e-6 relative"
import numpy as np
import pytest

import arima_gap
from oracle import arima_oracle as ao
from oracle import tad_oracle as orc

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.abs(b)


def assert_same(got, want, what):
    """every finite prediction within 1e-6 (in fact bit-identical); a fit whose optimiser walks into a non-finite
    likelihood yields NaN on both sides at the same index (Python: abs(x - nan) > sigma is False -> no anomaly row).
    That happens where the reference algorithm itself is numerically void: Box-Cox with a strongly negative lambda
    compresses the data to a variance of ~1e-18 next to the 1e6 diffuse prior, and P - K F K' cancels to a negative F
    (the scipy-driven restatement oracle/arima_oracle.py:calculate_arima shows the same NaNs at a similar rate)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all(), (what, "NaN predictions at different indices")
    inf = np.isinf(want)                  # inv_boxcox overflow (huge lambda): same infinity on both sides
    assert (got[inf] == want[inf]).all(), (what, "infinite predictions differ")
    fin = ~(nan | inf)
    rel = rel_err(got[fin], want[fin])
    assert (rel <= TOL).all(), (what, float(rel.max()), int((rel > TOL).sum()))
    assert (got[fin].view(np.uint64) == want[fin].view(np.uint64)).all(), (what, "within 1e-6 but not bit-identical", float(rel.max()))


def test_series_arima_golden_series(engine, golden):
    x, sd = golden["throughput_list"], golden["stddev"]
    got = engine.series_arima(x)
    assert_same(got, ao.calculate_arima_exact(x), "golden series")
    # the reference's asserted goldens
    verdict = engine.series_arima_anomaly(x, sd)
    assert verdict.tolist() == golden["expected_anomaly_list_arima"]
    five = [int(str(float(v))[:5]) for v in got]
    hits = sum(a == b for a, b in zip(five, golden["expected_arima_row_list"]))
    assert hits == 81, hits                      # (the reference's own two lists agree at 78 / 90)
    arima_gap.check(got, golden, "GPU")          # the same nine misses with the same bits as the CPU gate (tests/arima_gap.py)
    # reported, not gated: distance to the reference's unasserted full-precision list (:288-318)
    full = np.array(golden["expanded_arima_row_list"])
    rel = rel_err(got, full)
    print("vs expanded_arima_row_list: median %.3g p90 %.3g max %.3g; 5-digit hits %d/90" % (np.median(rel), np.percentile(rel, 90), rel.max(), hits))


def test_series_arima_none_cases(engine):
    assert engine.series_arima([1, 2, 3]) is None
    assert engine.series_arima([5, 5, 5, 5, 5]) is None
    assert engine.series_arima([5, 0, 7, 9, 11]) is None
    assert engine.series_arima_anomaly([1, 2, 3], 1.0).tolist() == [False]
    assert engine.series_arima_anomaly([5, 5, 5, 5, 5], 1.0).tolist() == [False]


def test_series_arima_reference_series(engine, ref_outputs):
    # rand_n250 carries x0.05 / x2.5 / x12 spikes (flat likelihoods after each spike): the case that used to diverge
    for name in ("rand_n90", "rand_n250"):
        e = ref_outputs["series"][name]
        x, sd = e["x"], e["stddev_numpy_ddof1"]
        want = ao.calculate_arima_exact(x)
        assert_same(engine.series_arima(x), want, name)
        verdict = engine.series_arima_anomaly(x, sd)
        assert verdict.tolist() == [abs(float(a) - p) > sd for a, p in zip(x, want)]      # identical verdicts
    assert engine.series_arima(ref_outputs["series"]["ramp"]["x"]) is None                   # contains 0 -> boxcox raises


def test_series_arima_random_shapes(engine):
    # short histories, large / small magnitudes, spikes, near-constant data, lambda far from 0
    rng = np.random.default_rng(11)
    for i in range(40):
        n = int(rng.integers(4, 70))
        x = 10 ** rng.uniform(0.5, 12) * np.exp(rng.normal(0, rng.uniform(1e-4, 0.8), n))
        if i % 3 == 0:
            x[rng.integers(0, n)] *= rng.uniform(2, 15)
        x = (np.floor(x) + 1.0).astype(np.uint64)
        want = ao.calculate_arima_exact(x)
        got = engine.series_arima(x)
        if want is None:
            assert got is None, i
        else:
            assert_same(got, want, "random series %d" % i)


def test_series_arima_every_length(engine):
    """Every history length from 4 to 41: the strided column streams of k_arima_prep (batches of eight, one batch ahead) and the
    five-value window of k_arima_start (four rows at a time, four ahead) have their seams at 8, 16, 17, 24, ... and at
    rows = 4, 5, 8, 9, ...  (Series with missing buckets, which the compaction's batches of eight grid cells skip: the job tests below.)"""
    rng = np.random.default_rng(23)
    base = np.floor(3e6 * np.exp(rng.normal(0, 0.3, 41))).astype(np.uint64) + 1
    for n in range(4, 42):
        x = base[:n].copy()
        x[n // 2] *= 3
        want = ao.calculate_arima_exact(x)
        got = engine.series_arima(x)
        assert want is not None and got is not None, n
        assert_same(got, want, "length %d" % n)


def check_job(engine, k, t, v, K):
    want = orc.run_job("ARIMA", k, t, v, agg_flow="svc")
    allp = engine.run("ARIMA", k, t, v, K, agg_flow="svc", emit_all=True)
    pk, pt, pv = want["points"]
    keep = np.repeat(np.array([r is not None for r in want_results(want)]), np.diff(want["ptr"]))
    assert allp.n_rows == int(keep.sum()) and (allp["key_id"] == pk[keep]).all() and (allp["flow_end_s"] == pt[keep]).all()
    assert (allp["throughput"] == orc.u64_to_f64(pv)[keep]).all()
    assert (allp["stddev"] == np.repeat(want["sigma"], np.diff(want["ptr"]))[keep]).all()
    assert_same(allp["algo_calc"], want["calc_all"][keep], "job predictions")
    assert (allp["anomaly"].astype(bool) == want["anomaly_all"][keep]).all()       # zero verdict flips
    res = engine.run("ARIMA", k, t, v, K, agg_flow="svc")
    assert res.n_rows == want["n_anomalies"] and (res["key_id"] == want["key_id"]).all() and (res["flow_end_s"] == want["flow_end_s"]).all()
    assert_same(res["algo_calc"], want["algo_calc"], "job anomaly rows")
    assert res.stats["keys_no_result"] == want["keys_no_result"]
    return want, res


def want_results(want):
    """per key: None when the oracle's calculate_arima returned None (all-zero calc and n > 0 is not a criterion: use the
    counters the oracle keeps)"""
    return want["arima_results"]


def test_job_arima_matches_oracle(engine):
    k, t, v = orc.synth_rows(0, 4000, 24, 40)
    want, res = check_job(engine, k, t, v, 24)
    assert res.stats["arima_fits"] == want["n_points"] - 3 * want["n_keys"]
    assert res.stats["kalman_steps"] == want["kalman_steps"]          # the flop figure's counter, cross-checked


def test_job_arima_c3_table_sample(engine):
    # >= 200 keys of the C3 table's shape (250 buckets, ~4 rows per point): every prediction and verdict
    k, t, v = orc.synth_rows(0, 220000, 220, 250)
    want, res = check_job(engine, k, t, v, 220)
    assert want["n_keys"] == 220 and want["n_points"] > 50000
    assert np.isnan(want["calc_all"]).mean() < 1e-3            # measured 18 of 53979, all on histories of 4 or 6 points
    assert res.stats["kalman_steps"] == want["kalman_steps"]


def test_job_arima_many_keys_share_the_position_cursor(engine):
    """9 000 keys x 9 buckets: three wavefronts per series position pull their keys from ONE cursor (k_arima_fit's refill, a
    64-bit global atomic); which lane fits which key must not enter the results — every prediction against the oracle."""
    k, t, v = orc.synth_rows(0, 9000 * 9 * 2, 9000, 9)
    want, res = check_job(engine, k, t, v, 9000)
    assert want["n_keys"] > 8900
    assert res.stats["kalman_steps"] == want["kalman_steps"]


def test_job_arima_keys_without_result(engine):
    # key 0: 3 points (n <= 3), key 1: constant, key 2: contains a zero, key 3: a real series
    key = np.repeat(np.arange(4, dtype=np.uint64), [3, 6, 6, 30])
    t = np.concatenate([np.arange(n) for n in (3, 6, 6, 30)]).astype(np.int64) * 60
    rng = np.random.default_rng(4)
    v = np.concatenate([[5, 6, 7], [9] * 6, [4, 0, 5, 6, 7, 8], 1e9 * np.exp(rng.normal(0.0, 0.5, 30))]).astype(np.uint64)
    want, _ = check_job(engine, key, t, v, 4)
    assert want["keys_no_result"] == 3
    allp = engine.run("ARIMA", key, t, v, 4, agg_flow="svc", emit_all=True)
    assert allp.stats["keys_no_result"] == 3
    assert allp.n_rows == 30 and (allp["key_id"] == 3).all()          # the other keys yield no rows at all (:284-287)
