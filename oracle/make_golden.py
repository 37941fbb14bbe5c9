"""Generate tests/golden/*.json from the REFERENCE (run in the build container only).

  python -m oracle.make_golden

1. reference_golden.json — the golden vectors of the reference's own unit tests, lifted from
   /root/reference/plugins/anomaly-detection/anomaly_detection_test.py (:199-217 throughput_list,
   :219-249 EWMA, :261-273 ARIMA 5-digit, :286 stddev, :288-318 expanded ARIMA, :320-391 verdicts)
   plus the generate_tad_sql_query cases (:46-195) evaluated by importing the reference test module.
2. reference_outputs.json — outputs of the reference's pure functions (calculate_ewma,
   calculate_ewma_anomaly, calculate_dbscan_anomaly, remove_meaningless_labels) run from
   /root/reference on seeded series (random, adversarial), so the oracle and the HIP path can be
   checked against the reference itself on more than one series.
Nothing here is copied from the reference: the functions are executed where they lie.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT_DIR = os.path.join(ROOT, "tests", "golden")

from oracle import ref_loader  # noqa: E402


def seeded_series():
    """name -> list[int] (uint64 values).  Deterministic."""
    rng = np.random.default_rng(20220811)
    out = {}
    base = 4_005_000_000
    for i, n in enumerate([1, 2, 3, 4, 5, 7, 16, 63, 64, 65, 90, 128, 250, 1000]):
        v = base + rng.integers(-2_000_000, 2_000_000, size=n)
        if n >= 16:
            for j in rng.choice(n, size=max(1, n // 30), replace=False):
                v[j] = int(v[j] * rng.choice([0.05, 0.25, 2.5, 12.0]))
        out["rand_n%d" % n] = [int(x) for x in v]
    out["constant_8"] = [4_000_000_000] * 8
    out["zeros_6"] = [0] * 6
    out["two_clusters"] = [int(x) for x in np.concatenate([1_000_000_000 + rng.integers(0, 1000, 20),
                                                          9_000_000_000 + rng.integers(0, 1000, 3)])]
    # exact-eps chain: consecutive gaps of exactly 250000000 (inclusive <= eps)
    out["eps_chain"] = [1_000_000_000 + 250_000_000 * i for i in range(10)]
    out["eps_plus1_chain"] = [1_000_000_000 + 250_000_001 * i for i in range(10)]
    # values at and above 2^53 / 2^63: u64 -> double rounding
    out["huge"] = [2**53 + 1, 2**53 + 3, 2**63 + 12345, 2**64 - 1, 2**63, 2**53, 2**53 + 2, 2**62]
    out["ramp"] = [1000 * i * i for i in range(40)]
    return out


def drop_golden():
    """tests/golden/drop_outputs.json: the reference drop-detection UDF (snowflake/udfs/udfs/drop_detection/
    drop_detection_udf.py) executed from where it lies on its own test data (drop_detection_udf_test.py:7-128) and on
    seeded series."""
    import ast
    from oracle import drop_oracle as dro
    udf = dro.load_reference_udf()
    test_src = open(os.path.join(os.path.dirname(dro.UDF_FILE), "drop_detection_udf_test.py")).read()
    tree = ast.parse(test_src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef)][0]
    consts = {t.targets[0].id: ast.literal_eval(t.value) for t in cls.body if isinstance(t, ast.Assign)}
    rng = np.random.default_rng(20230117)
    series = {"reference_test": [int(r[3]) for r in consts["aggregated_flows"]]}
    for n in (2, 3, 4, 7, 8, 9, 20, 31, 127, 128, 129, 300, 1000):
        v = rng.poisson(6.0, size=n).astype(np.int64)
        if n >= 7:
            v[rng.integers(0, n)] += int(rng.integers(50, 400))
        series["poisson_n%d" % n] = [int(x) for x in v]
    big = rng.integers(2**33, 2**34, size=200)
    big[77] = 2**44 + 12345                       # sums of squares far beyond 2^53: the float arithmetic order matters
    series["big_counts"] = [int(x) for x in big]
    series["constant"] = [5] * 10
    out = {"source": dro.UDF_FILE, "expected_result": consts["expected_result"], "detection_id": consts["detection_id"], "series": {}}
    for name, xs in series.items():
        d = udf.DropDetection()
        for i, x in enumerate(xs):
            next(d.process("initial", "det-1", "ns/pod", "ingress", "day-%04d" % i, x))
        rows = list(d.end_partition())
        out["series"][name] = {"x": xs, "rows": [[r[5], r[6], r[7], int(r[8])] for r in rows]}   # avg, std, date, number
    with open(os.path.join(OUT_DIR, "drop_outputs.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote drop_outputs.json:", {k: len(v["rows"]) for k, v in out["series"].items()})


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    drop_golden()
    ad = ref_loader.load_reference_job()
    t = ref_loader.load_reference_tests()

    golden = {
        "source": "/root/reference/plugins/anomaly-detection/anomaly_detection_test.py",
        "throughput_list": t.throughput_list,
        "expected_ewma_row_list": t.expected_ewma_row_list,
        "expected_arima_row_list": t.expected_arima_row_list,
        "expanded_arima_row_list": t.expanded_arima_row_list,
        "stddev": t.stddev,
        "expected_anomaly_list_arima": t.expected_anomaly_list_arima,
        "expected_anomaly_list_ewma": t.expected_anomaly_list_ewma,
        "expected_dbscan_anomaly_list": t.expected_dbscan_anomaly_list,
    }
    # sanity: the reference's own functions reproduce its goldens in this container
    assert ad.calculate_ewma(t.throughput_list) == t.expected_ewma_row_list
    assert ad.calculate_ewma_anomaly(t.throughput_list, t.stddev) == t.expected_anomaly_list_ewma
    assert ad.calculate_dbscan_anomaly(t.throughput_list, t.stddev) == t.expected_dbscan_anomaly_list
    # SQL goldens: the parametrize table of test_generate_sql_query (:46-195)
    cases = t.test_generate_sql_query.pytestmark[0].args[1]
    golden["sql_cases"] = [{"args": list(a), "sql": s} for a, s in cases]
    for c in golden["sql_cases"]:
        assert ad.generate_tad_sql_query(*c["args"]) == c["sql"]
    with open(os.path.join(OUT_DIR, "reference_golden.json"), "w") as f:
        json.dump(golden, f, indent=1)

    outputs = {"source": "reference functions executed from /root/reference by oracle/make_golden.py", "series": {}}
    for name, x in seeded_series().items():
        sd = float(np.std(np.array(x, dtype=np.float64), ddof=1)) if len(x) > 1 else None
        entry = {"x": [str(v) for v in x], "stddev_numpy_ddof1": sd}
        entry["ewma"] = ad.calculate_ewma(x)
        entry["ewma_anomaly"] = ad.calculate_ewma_anomaly(x, sd)
        entry["ewma_anomaly_half_sigma"] = ad.calculate_ewma_anomaly(x, sd / 2 if sd is not None else None)
        entry["dbscan_anomaly"] = [bool(b) for b in ad.calculate_dbscan_anomaly(x, sd)]
        outputs["series"][name] = entry
    labels = ['{"app":"a","pod-template-hash":"x"}', '{"b":"2","a":"1","controller-revision-hash":"h"}',
              'not json', '{}', '{"pod-template-generation":"3"}']
    outputs["remove_meaningless_labels"] = [[s, ad.remove_meaningless_labels(s)] for s in labels]
    with open(os.path.join(OUT_DIR, "reference_outputs.json"), "w") as f:
        json.dump(outputs, f, indent=1)
    print("wrote", os.listdir(OUT_DIR))


if __name__ == "__main__":
    main()
