"""CPU tests: the ARIMA oracle (oracle/arima_oracle.py) against the reference's golden vectors.

What the reference pins (anomaly_detection_test.py): the verdict list (:320-345, asserted) and the first
five characters of each prediction (:261-283, asserted).  The full-precision list (:288-318) is never
asserted and disagrees with the asserted one at 12 of 90 indices, so a 1e-6 match against statsmodels
is not pinned by anything ("parity unpinned"); the distances are recorded here."""
import numpy as np
import pytest

from oracle import arima_oracle as ao
import arima_gap


@pytest.fixture(scope="module")
def golden_pred(golden):
    return ao.calculate_arima(golden["throughput_list"])


def test_arima_verdicts_equal_reference_golden(golden, golden_pred):
    x, sd = golden["throughput_list"], golden["stddev"]
    verdict = [abs(float(a) - p) > sd for a, p in zip(x, golden_pred)]
    assert verdict == golden["expected_anomaly_list_arima"]
    assert [i for i, b in enumerate(verdict) if b] == [58, 59, 60, 68]
    assert ao.calculate_arima_anomaly(x, sd) == golden["expected_anomaly_list_arima"]


def test_arima_values_against_both_reference_lists(golden, golden_pred):
    five = [int(str(v)[:5]) for v in golden_pred]
    hits = sum(a == b for a, b in zip(five, golden["expected_arima_row_list"]))
    # the asserted and the unasserted reference lists agree with each other at only 78 of 90 indices
    self_hits = sum(int(str(v)[:5]) == b for v, b in zip(golden["expanded_arima_row_list"], golden["expected_arima_row_list"]))
    assert self_hits == 78
    assert hits >= 80, hits                      # measured: 81 / 90 (77 before round 3's switch to statsmodels' parameter signs)
    full = np.array(golden["expanded_arima_row_list"])
    rel = np.abs(np.array(golden_pred) - full) / full
    assert np.median(rel) < 1e-7 and np.percentile(rel, 90) < 5e-5 and rel.max() < 5e-4   # measured 1e-9 / 1.2e-5 / 2.5e-4
    # the first three predictions are inv_boxcox(boxcox(x)) (:241,255-256)
    assert np.allclose(golden_pred[:3], golden["throughput_list"][:3], rtol=1e-12)


def test_none_cases():
    assert ao.calculate_arima([1, 2, 3]) is None                 # len <= 3 (:232-234)
    assert ao.calculate_arima([5, 5, 5, 5, 5]) is None           # constant -> boxcox raises (:260-264)
    assert ao.calculate_arima([5, 0, 7, 9, 11]) is None          # non-positive -> boxcox raises
    assert ao.calculate_arima_anomaly([1, 2, 3], 1.0) == [False]  # :284-287


def test_boxcox_lambda_matches_scipy(golden):
    from scipy import stats
    x = np.array(golden["throughput_list"], dtype=np.float64)
    assert abs(ao.boxcox_mle_lambda(x) - stats.boxcox(x)[1]) < 1e-6
    rng = np.random.default_rng(1)
    for _ in range(5):
        x = rng.uniform(1e9, 5e9, size=40)
        assert abs(ao.boxcox_mle_lambda(x) - stats.boxcox(x)[1]) < 1e-5 * max(1.0, abs(stats.boxcox(x)[1]))


def test_c_kalman_equals_numpy_kalman():
    rng = np.random.default_rng(0)
    y = np.cumsum(rng.normal(size=80)) + 50
    for p in [(0.3, -0.4, 1.2), (0.0, 0.0, 0.5), (-0.9, 0.8, 1e-4), (0.99, -0.99, 3.0)]:
        a = ao.kalman_arima111(y, *p)
        b = ao.kalman_fast(y, *p)
        assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(a[1])


def test_start_params_small_sample_paths():
    # the ValueError fallbacks of _conditional_sum_squares for 3, 4 and 5 observations
    for n in (3, 4, 5, 6, 7):
        y = np.array([1.0, 1.5, 1.2, 1.9, 1.7, 2.4, 2.2])[:n]
        phi, theta, var = ao.start_params(y)
        assert abs(phi) < 1 and abs(theta) < 1 and var >= 1e-10
        if n <= 4:
            assert phi == 0.0 and theta == 0.0


# ---- the fixed-arithmetic restatement (oracle/arima_exact.c), the checker the GPU path is held to bit for bit ----
@pytest.fixture(scope="module")
def golden_exact(golden):
    return ao.calculate_arima_exact(golden["throughput_list"])


def test_exact_verdicts_equal_reference_golden(golden, golden_exact):
    x, sd = golden["throughput_list"], golden["stddev"]
    verdict = [abs(float(a) - p) > sd for a, p in zip(x, golden_exact)]
    assert verdict == golden["expected_anomaly_list_arima"]                       # anomaly_detection_test.py:320-345
    assert ao.calculate_arima_anomaly_exact(x, sd) == golden["expected_anomaly_list_arima"]


def test_exact_values_against_both_reference_lists(golden, golden_exact):
    five = [int(str(v)[:5]) for v in golden_exact]
    hits = sum(a == b for a, b in zip(five, golden["expected_arima_row_list"]))   # :261-283, first five characters
    assert hits == 81, hits                       # (the reference's own two lists share 78)
    # the gate is the SET of missed indices, the prediction's bits at each of them and its distance to the reference's
    # full-precision value (tests/arima_gap.py, tests/golden/arima_gap.json): a tenth miss, or another nine, fails
    rows = arima_gap.check(golden_exact, golden, "oracle/arima_exact.c")
    assert [r["index"] for r in rows] == [59, 60, 62, 72, 73, 74, 75, 77, 80]
    assert [r["index"] for r in rows if r["reference_lists_agree"]] == [80]      # every other miss: the reference's two lists disagree there
    with open(arima_gap.GAP_TABLE) as f:                                          # the table DESIGN.md section 4 quotes is this one
        table = f.read()
    assert table == arima_gap.render(rows)
    import os
    with open(os.path.join(arima_gap.ROOT, "DESIGN.md")) as f:
        design = f.read()
    assert all(("  " + line) in design for line in table.splitlines()), "DESIGN.md section 4 does not quote tests/golden/arima_gap_table.md"
    full = np.array(golden["expanded_arima_row_list"])                            # :288-318, never asserted by the reference
    rel = np.abs(np.array(golden_exact) - full) / full
    assert np.median(rel) < 1e-7 and np.percentile(rel, 90) < 5e-5 and rel.max() < 5e-4   # measured 9.3e-10 / 4.9e-6 / 2.5e-4
    assert np.allclose(golden_exact[:3], golden["throughput_list"][:3], rtol=1e-12)


def test_exact_agrees_with_scipy_driven_restatement(golden, golden_pred, golden_exact):
    """Same model, same optimiser settings; scipy's L-BFGS-B (compact-matrix subspace step) vs the two-loop recursion and
    glibc vs tad_detmath.h differ in the last bits, which the loosely stopped optimiser amplifies on flat likelihoods."""
    rel = np.abs(np.array(golden_exact) - np.array(golden_pred)) / np.abs(np.array(golden_pred))
    assert np.median(rel) < 1e-8 and (rel <= 1e-6).sum() >= 60 and rel.max() < 5e-3       # measured 9e-11, 72 / 90, 1.6e-4


def test_exact_pieces_against_the_numpy_restatement(golden):
    lib = ao._load_exact()
    x = np.array(golden["throughput_list"], dtype=np.float64)
    import ctypes
    lam = ctypes.c_double()
    assert lib.arima_exact_boxcox_lambda(x.ctypes.data, x.size, ctypes.byref(lam)) == 1
    assert abs(lam.value - ao.boxcox_mle_lambda(x)) < 1e-6
    y = np.ascontiguousarray(ao.boxcox_transform(x, lam.value))
    for n in (4, 5, 6, 7, 30, 90):
        u = np.zeros(3)
        lib.arima_exact_start_params(y.ctypes.data, n, u.ctypes.data)
        want = ao.untransform_params(*ao.start_params(y[:n]))
        assert np.allclose(u, want, rtol=1e-7, atol=1e-9), (n, u, want)
    rng = np.random.default_rng(0)
    ys = np.cumsum(rng.normal(size=80)) + 50
    for p in [(0.3, -0.4, 1.2), (0.0, 0.0, 0.5), (-0.9, 0.8, 1e-4), (0.99, -0.99, 3.0)]:
        a = ao.kalman_arima111(ys, *p)
        b = ao.kalman_exact(ys, *p)
        # (sigma2 = 1e-4 next to the 1e6 diffuse prior: P - K F K' cancels ten digits, whatever the formulation)
        assert abs(a[0] - b[0]) <= 1e-8 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-8 * abs(a[1])


def test_exact_none_cases():
    assert ao.calculate_arima_exact([1, 2, 3]) is None
    assert ao.calculate_arima_exact([5, 5, 5, 5, 5]) is None
    assert ao.calculate_arima_exact([5, 0, 7, 9, 11]) is None
    assert ao.calculate_arima_anomaly_exact([1, 2, 3], 1.0) == [False]


# ---- the contract's likelihood (collapsed form, arima_exact.c:arima_nll4_collapsed) against the textbook three-state filter ----
def test_collapsed_filter_is_the_same_likelihood():
    """H = 0 puts Z' in the null space of the filtered covariance, so from t = 1 on only p11 evolves and the level is known
    exactly: the collapsed recursion is the general three-state filter in exact arithmetic.  In floating point the general
    form computes those zeros as differences of numbers of size 1e6 (the diffuse prior): the two agree to ~1e-10 unless
    sigma^2 is tiny, where it is the GENERAL form that has lost digits (P - K F K' cancels)."""
    import ctypes
    lib = ao._load_exact()
    rng = np.random.default_rng(0)
    ys = np.ascontiguousarray(np.cumsum(rng.normal(size=80)) + 50)
    try:
        for (phi, theta, s2), tol in [((0.3, -0.4, 1.2), 1e-11), ((0.0, 0.0, 0.5), 1e-9), ((-0.9, 0.8, 1e-4), 1e-8), ((0.99, -0.99, 3.0), 1e-10),
                                      ((0.5, -0.95, 1e-6), 1e-5)]:
            u = ao.untransform_params(phi, theta, s2)
            out = []
            for mode in (0, 1):
                lib.arima_exact_set_filter(mode)
                fc = ctypes.c_double()
                out.append((lib.arima_exact_nll(ys.ctypes.data, ys.size, float(u[0]), float(u[1]), float(u[2]), ctypes.byref(fc)), fc.value))
            assert abs(out[0][0] - out[1][0]) <= tol * abs(out[0][0]), (phi, theta, s2, out)
            assert abs(out[0][1] - out[1][1]) <= tol * abs(out[0][1]), (phi, theta, s2, out)
            # and the collapsed form against the plain numpy three-state filter (no structure used at all)
            ll = ao.kalman_arima111(ys, phi, theta, s2)[0]
            assert abs(-out[1][0] * ys.size - ll) <= max(tol, 1e-9) * abs(ll)
    finally:
        lib.arima_exact_set_filter(1)


def test_textbook_filter_fit_agrees_with_the_contract(golden, golden_exact, golden_pred):
    """The whole walk-forward with the textbook three-state likelihood (round 2's contract) instead of the collapsed one: same
    verdicts, the same distance to the reference's lists — the choice of recursion is not what the remaining distance is."""
    lib = ao._load_exact()
    x, sd = golden["throughput_list"], golden["stddev"]
    lib.arima_exact_set_filter(0)
    try:
        c = {}
        pred = ao.calculate_arima_exact(x, counters=c)
    finally:
        lib.arima_exact_set_filter(1)
    verdict = [abs(float(a) - p) > sd for a, p in zip(x, pred)]
    assert verdict == golden["expected_anomaly_list_arima"]
    five = [int(str(v)[:5]) for v in pred]
    hits = sum(a == b for a, b in zip(five, golden["expected_arima_row_list"]))
    assert hits >= 80, hits                       # measured 82 / 90 (the contract: 81)
    rel = np.abs(np.array(pred) - np.array(golden_exact)) / np.abs(np.array(golden_exact))
    assert np.median(rel) < 1e-8 and (rel <= 1e-6).sum() >= 60 and rel.max() < 5e-3
