"""GPU: the drop detector (TAD_ALGO_DROP / tad_series_drop) against the reference UDF's golden, its outputs on seeded
series (tests/golden/drop_outputs.json) and the oracle on synthetic tables.  Bit-exact: the kernel sums in numpy's
pairwise order."""
import json
import os

import numpy as np
import pytest

from oracle import drop_oracle as dro
from oracle import tad_oracle as orc
from theia_amd import anomaly_detection as ad
from theia_amd import drop_detection as dd

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drop_outputs.json")


@pytest.fixture(scope="module")
def drop_golden():
    with open(GOLD) as f:
        return json.load(f)


def test_end_partition_like_the_reference_unit_test(engine, drop_golden):
    # drop_detection_udf_test.py:141-171, same calls, same exact-equality assertions
    ad.set_engine(engine)
    try:
        det = dd.DropDetection()
        for i, x in enumerate(drop_golden["series"]["reference_test"]["x"]):
            next(det.process(job_type="initial", detection_id=drop_golden["detection_id"], endpoint="antrea-test/Pod-A",
                             direction="ingress", date="2022-01-%02d" % (i + 1), drop_number=x))
        results = list(det.end_partition())
        assert len(results) == len(drop_golden["expected_result"]) == 1
        _, detection_id, _, endpoint, direction, avg_drop, stdev_drop, date, number = results[0]
        assert detection_id == drop_golden["detection_id"]
        assert [endpoint, direction, avg_drop, stdev_drop, date, number] == drop_golden["expected_result"][0]
    finally:
        ad.set_engine(None)


def test_series_equal_reference_udf_outputs_bit_for_bit(engine, drop_golden):
    for name, e in drop_golden["series"].items():
        out = engine.series_drop(e["x"])
        if len(e["x"]) < 3:
            assert out is None and e["rows"] == [], name
            continue
        mean, std, verdict = out
        assert np.flatnonzero(verdict).tolist() == [int(r[2].split("-")[1]) for r in e["rows"]], name
        for r in e["rows"]:
            assert r[0] == mean and r[1] == std, (name, r[:2], mean, std)
        omean, ostd, overdict = dro.drop_detection_series(e["x"])
        assert (mean, std) == (omean, ostd) and (verdict == overdict).all(), name


@pytest.mark.parametrize("n_rows,K,T", [(50000, 300, 40), (400000, 2000, 365), (30000, 4000, 30)])
def test_job_matches_oracle(engine, n_rows, K, T):
    rng = np.random.default_rng(n_rows)
    key = rng.integers(0, K, size=n_rows).astype(np.uint64)
    day = 19000 + rng.integers(0, T, size=n_rows).astype(np.int64)          # one bucket per day
    drops = rng.poisson(3.0, size=n_rows).astype(np.uint64)
    spike = rng.random(n_rows) < 0.002
    drops = np.where(spike, drops * np.uint64(200) + np.uint64(500), drops)
    want = dro.run_job(key, day, drops)
    res = engine.run("DROP", key, day, drops, K, agg_flow="svc")
    assert res.stats["keys_no_result"] == want["keys_no_result"] and res.stats["n_points"] == want["n_points"]
    assert res.n_rows == want["n_anomalies"] > 0
    for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
        assert (res[f] == want[f]).all(), f
    # emit_all: every point of every key with >= 3 samples, verdict column consistent
    allp = engine.run("DROP", key, day, drops, K, agg_flow="svc", emit_all=True)
    assert int(allp["anomaly"].sum()) == want["n_anomalies"]
    upper = allp["algo_calc"] + 3.0 * allp["stddev"]
    lower = allp["algo_calc"] - 3.0 * allp["stddev"]
    assert (((allp["throughput"] > upper) | (allp["throughput"] < lower)) == allp["anomaly"].astype(bool)).all()


def test_table_function(engine):
    ad.set_engine(engine)
    try:
        ep = ["ns/a"] * 12 + ["ns/b"] * 12 + ["10.0.0.9"] * 2
        di = ["ingress"] * 12 + ["egress"] * 12 + ["ingress"] * 2
        date = ["2022-03-%02d" % (i + 1) for i in range(12)] * 2 + ["2022-03-01", "2022-03-02"]
        num = [3, 2, 4, 3, 2, 90, 3, 4, 2, 3, 4, 2] + [5] * 12 + [1, 100]
        rows = dd.drop_detection_table(ep, di, date, num, detection_id="d1")
        assert len(rows) == 1 and rows[0][3:5] == ("ns/a", "ingress") and rows[0][7:] == ("2022-03-06", 90)
        mean, std, _ = dro.drop_detection_series(num[:12])
        assert rows[0][5] == mean and rows[0][6] == std
    finally:
        ad.set_engine(None)
