"""The ARIMA parity gap against the reference's ASSERTED golden list, index by index (VERDICT r4 "next" #1b).

The reference asserts the first five characters of all 90 predictions (anomaly_detection_test.py:261-283); the contract
this repository computes (oracle/arima_exact.c == the GPU, bit for bit) hits 81.  `hits >= 80` let a regression from 81
to 80 pass; the gate is now the SET of missed indices and, per missed index, the prediction's bits and its relative
distance to the reference's unasserted full-precision list (:288-318).  tests/golden/arima_gap.json is written by
`python tests/arima_gap.py --write` from the exact oracle; DESIGN.md section 4 quotes tests/golden/arima_gap_table.md,
which the same command renders — the table is generated, not typed."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GAP_JSON = os.path.join(HERE, "golden", "arima_gap.json")
GAP_TABLE = os.path.join(HERE, "golden", "arima_gap_table.md")


def five(v):
    """the reference's comparison: the first five characters of str(prediction) (anomaly_detection_test.py:277-283)"""
    return int(str(float(v))[:5])


def gap_rows(pred, golden):
    """one row per index the contract misses in the asserted list"""
    asserted, full = golden["expected_arima_row_list"], golden["expanded_arima_row_list"]
    rows = []
    for i, (p, a, f) in enumerate(zip(pred, asserted, full)):
        if five(p) != a:
            rows.append({"index": i, "asserted": a, "unasserted": five(f), "contract": five(p), "prediction": float(p).hex(),
                         "rel_to_unasserted": abs(float(p) - f) / f, "reference_lists_agree": five(f) == a})
    return rows


def render(rows):
    out = ["| index | asserted (:261-273) | unasserted list (:288-318) | contract (oracle == GPU) | contract vs unasserted (rel.) | the reference's two lists agree |",
           "|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %d | %d | %d | %d | %.2e | %s |" % (r["index"], r["asserted"], r["unasserted"], r["contract"], r["rel_to_unasserted"],
                                                          "yes" if r["reference_lists_agree"] else "no"))
    return "\n".join(out) + "\n"


def load():
    with open(GAP_JSON) as f:
        return json.load(f)


def check(pred, golden, what):
    """pred (90 predictions of the golden series) must miss exactly the recorded indices, with the recorded bits"""
    want = load()
    got = gap_rows(pred, golden)
    assert [r["index"] for r in got] == [r["index"] for r in want["missed"]], (what, "the set of missed indices changed",
                                                                              [r["index"] for r in got])
    assert 90 - len(got) == want["hits"] == 81, what
    for g, w in zip(got, want["missed"]):
        assert g["prediction"] == w["prediction"], (what, g["index"], "prediction bits changed", g["prediction"], w["prediction"])
        assert g["contract"] == w["contract"] and g["asserted"] == w["asserted"], (what, g["index"])
        # the distance to the reference's own full-precision value, gated at its measured size (+ 1 %)
        assert g["rel_to_unasserted"] <= w["rel_to_unasserted"] * 1.01, (what, g["index"], g["rel_to_unasserted"])
    return got


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    from oracle import arima_oracle as ao
    with open(os.path.join(HERE, "golden", "reference_golden.json")) as f:
        golden = json.load(f)
    rows = gap_rows(ao.calculate_arima_exact(golden["throughput_list"]), golden)
    if "--write" in sys.argv:
        with open(GAP_JSON, "w") as f:
            json.dump({"source": "oracle/arima_exact.c on tests/golden/reference_golden.json:throughput_list (python tests/arima_gap.py --write)",
                       "hits": 90 - len(rows), "missed": rows}, f, indent=1)
        with open(GAP_TABLE, "w") as f:
            f.write(render(rows))
    print(render(rows))
