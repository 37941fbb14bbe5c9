"""GPU parity tests for the ARIMA detector: HIP path through the C ABI vs the oracle and the reference's
golden verdicts.

Contract (DESIGN.md "ARIMA parity"): verdicts identical to the reference's golden list and to the oracle;
predictions compared with the oracle at BASELINE.json's 1e-6 relative tolerance, of which the FRACTION
that meets it is asserted together with bounds on the rest.  A hard 1e-6 on every point is not a property
the reference algorithm has: its objective is minimised by L-BFGS-B on forward-difference gradients
(h = 1e-5) and stopped at factr = 1e7, so a 1-ulp difference in one likelihood value (libm log, summation
order) moves the gradient by 1e-11, the next iterate by 1e-10 and the line-search interpolation by 1e-6
(trace: tools/arima_trace.cpp vs tools/arima_trace_scipy.py at golden index 64).  The reference's own two
golden lists for this series (anomaly_detection_test.py:261-273 vs :288-318) differ by up to 2.6e-4."""
TOL = 1e-6   # BASELINE.json north_star: "EWMA/ARIMA scores within 1e-6 relative"
import numpy as np
import pytest

from oracle import arima_oracle as ao
from oracle import tad_oracle as orc

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.abs(b)


def test_series_arima_golden_series(engine, golden):
    x, sd = golden["throughput_list"], golden["stddev"]
    got = engine.series_arima(x)
    want = np.array(ao.calculate_arima(x))
    rel = rel_err(got, want)
    print("GPU vs oracle: median %.3g p90 %.3g max %.3g; within 1e-6: %d/90" % (np.median(rel), np.percentile(rel, 90), rel.max(), int((rel <= 1e-6).sum())))
    assert (rel <= TOL).sum() >= 60 and np.percentile(rel, 90) < 1e-4 and rel.max() < 5e-3   # measured 74/90, 3.8e-6, 2.5e-4
    # the reference's asserted goldens
    verdict = engine.series_arima_anomaly(x, sd)
    assert verdict.tolist() == golden["expected_anomaly_list_arima"]
    five = [int(str(float(v))[:5]) for v in got]
    hits = sum(a == b for a, b in zip(five, golden["expected_arima_row_list"]))
    print("5-digit matches vs the reference's asserted list: %d/90" % hits)
    assert hits >= 75


def test_series_arima_none_cases(engine):
    assert engine.series_arima([1, 2, 3]) is None
    assert engine.series_arima([5, 5, 5, 5, 5]) is None
    assert engine.series_arima([5, 0, 7, 9, 11]) is None
    assert engine.series_arima_anomaly([1, 2, 3], 1.0).tolist() == [False]
    assert engine.series_arima_anomaly([5, 5, 5, 5, 5], 1.0).tolist() == [False]


def test_series_arima_reference_series(engine, ref_outputs):
    # benign Box-Cox lambdas (0.93, -0.05).  The nearly constant seeded series get lambda 5..20, i.e. transformed
    # values of 1e50..1e100 against a 1e6 "diffuse" prior: numerically meaningless for the reference algorithm
    # itself, so they are no parity yardstick.  rand_n250 carries x0.05 / x2.5 / x12 spikes: after each spike the
    # likelihood is flat and the optimiser's end point is reproducible to ~1e-5 only (CPU twin vs oracle: 33 %
    # within 1e-6, median 1.1e-5, max 2e-3) — the same effect as in the reference's own golden lists.
    for name, frac, med in (("rand_n90", 0.9, 1e-7), ("rand_n250", 0.2, 2e-4)):
        e = ref_outputs["series"][name]
        x, sd = e["x"], e["stddev_numpy_ddof1"]
        want = ao.calculate_arima(x)
        got = engine.series_arima(x)
        rel = rel_err(got, want)
        print("%s: within 1e-6 %.3f, median %.3g, max %.3g" % (name, (rel <= TOL).mean(), np.median(rel), rel.max()))
        assert (rel <= TOL).mean() >= frac and np.median(rel) < med and rel.max() < 2e-2
        verdict = engine.series_arima_anomaly(x, sd)
        assert verdict.tolist() == [abs(float(a) - p) > sd for a, p in zip(x, want)]      # identical verdicts
    assert engine.series_arima(ref_outputs["series"]["ramp"]["x"]) is None                   # contains 0 -> boxcox raises


def test_job_arima_matches_oracle(engine):
    k, t, v = orc.synth_rows(0, 4000, 24, 40)
    want = orc.run_job("ARIMA", k, t, v, agg_flow="svc")
    allp = engine.run("ARIMA", k, t, v, 24, agg_flow="svc", emit_all=True)
    pk, pt, pv = want["points"]
    assert allp.n_rows == want["n_points"] and (allp["key_id"] == pk).all() and (allp["flow_end_s"] == pt).all()
    assert (allp["throughput"] == orc.u64_to_f64(pv)).all()
    assert (allp["stddev"] == np.repeat(want["sigma"], np.diff(want["ptr"]))).all()
    rel = rel_err(allp["algo_calc"], want["calc_all"])
    pos = np.concatenate([np.arange(n) for n in np.diff(want["ptr"])])      # history length of each fit
    longh = pos >= 12
    print("job: within 1e-6 %.4f (history >= 12: %.4f), within 1e-3 %.4f, max %.3g (history >= 12: %.3g)"
          % ((rel <= TOL).mean(), (rel[longh] <= TOL).mean(), (rel <= 1e-3).mean(), rel.max(), rel[longh].max()))
    # fits on fewer than a dozen observations have 3 parameters and a multi-modal likelihood: the optimiser's
    # end point there is decided by rounding noise (tests/test_oracle_arima.py documents the same for the
    # reference's own goldens); they are bounded loosely, the rest tightly
    assert (rel <= TOL).mean() >= 0.8 and (rel[longh] <= TOL).mean() >= 0.85
    assert (rel[longh] <= 1e-3).mean() >= 0.98 and (rel <= 1e-3).mean() >= 0.93
    flips = int((allp["anomaly"].astype(bool) != want["anomaly_all"]).sum())
    print("verdict flips vs oracle: %d of %d points" % (flips, rel.size))
    assert flips <= max(1, rel.size // 200)
    res = engine.run("ARIMA", k, t, v, 24, agg_flow="svc")
    assert res.n_rows == int(allp["anomaly"].sum())
    assert res.stats["arima_fits"] == want["n_points"] - 3 * want["n_keys"] and res.stats["kalman_steps"] > 0


def test_job_arima_keys_without_result(engine):
    # key 0: 3 points (n <= 3), key 1: constant, key 2: contains a zero, key 3: a real series
    key = np.repeat(np.arange(4, dtype=np.uint64), [3, 6, 6, 30])
    t = np.concatenate([np.arange(n) for n in (3, 6, 6, 30)]).astype(np.int64) * 60
    rng = np.random.default_rng(4)
    v = np.concatenate([[5, 6, 7], [9] * 6, [4, 0, 5, 6, 7, 8], 1e9 * np.exp(rng.normal(0.0, 0.5, 30))]).astype(np.uint64)
    want = orc.run_job("ARIMA", key, t, v, agg_flow="svc")
    assert want["keys_no_result"] == 3
    allp = engine.run("ARIMA", key, t, v, 4, agg_flow="svc", emit_all=True)
    assert allp.stats["keys_no_result"] == 3
    assert allp.n_rows == 30 and (allp["key_id"] == 3).all()          # the other keys yield no rows at all (:284-287)
    rel = rel_err(allp["algo_calc"], want["calc_all"][-30:])
    assert (rel <= 1e-3).mean() >= 0.8 and rel[:3].max() < 1e-12
