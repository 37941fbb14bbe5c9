"""The drop-detection oracle (oracle/drop_oracle.py) pinned against the REFERENCE UDF: its own golden test
(snowflake/udfs/udfs/drop_detection/drop_detection_udf_test.py:130-139) and outputs of the reference code run on
seeded series in the build container (tests/golden/drop_outputs.json, made by oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import drop_oracle as dro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drop_outputs.json")


@pytest.fixture(scope="module")
def drop_golden():
    with open(GOLD) as f:
        return json.load(f)


def test_reference_unit_test_golden(drop_golden):
    x = drop_golden["series"]["reference_test"]["x"]
    mean, std, verdict = dro.drop_detection_series(x)
    # drop_detection_udf_test.py asserts exact equality on these
    assert [["antrea-test/Pod-A", "ingress", mean, std, "2022-01-05", 100]] == drop_golden["expected_result"]
    assert np.flatnonzero(verdict).tolist() == [4]


def test_oracle_equals_reference_udf_outputs_bit_for_bit(drop_golden):
    for name, e in drop_golden["series"].items():
        r = dro.drop_detection_series(e["x"])
        if len(e["x"]) < 3:
            assert r is None and e["rows"] == [], name
            continue
        mean, std, verdict = r
        want_idx = [int(row[2].split("-")[1]) for row in e["rows"]]
        assert np.flatnonzero(verdict).tolist() == want_idx, name
        for row in e["rows"]:
            assert row[0] == mean and row[1] == std, (name, row[:2], mean, std)     # exact: same summation order as numpy
            assert row[3] == e["x"][int(row[2].split("-")[1])]


def test_pairwise_sum_is_numpy_sum():
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 127, 128, 129, 1000, 4099):
        a = rng.normal(0, 1e9, size=n)
        assert dro.pairwise_sum(a) == float(np.add.reduce(a))


@pytest.mark.skipif(not dro.reference_available(), reason="needs /root/reference (build container)")
def test_against_the_live_reference_udf():
    udf = dro.load_reference_udf()
    rng = np.random.default_rng(9)
    for n in (3, 50, 200, 600):
        x = rng.poisson(20.0, size=n).astype(np.int64)
        x[rng.integers(0, n)] *= 40
        d = udf.DropDetection()
        for i, v in enumerate(x):
            next(d.process("initial", "id", "e", "egress", i, int(v)))
        rows = list(d.end_partition())
        mean, std, verdict = dro.drop_detection_series(x)
        assert [r[7] for r in rows] == np.flatnonzero(verdict).tolist()
        assert all(r[5] == mean and r[6] == std for r in rows)
