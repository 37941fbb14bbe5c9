"""Host half of the job (theia_amd/anomaly_detection.py: WHERE predicates, dictionary encoding, result-row
expansion, sentinel row, CLI) checked on CPU against the string-column oracle (oracle/job_oracle.py).
The numeric middle is stood in by the key-id oracle here; tests/test_gpu_job.py runs the same cases through
the HIP engine."""
import io
import json

import numpy as np
import pytest

from oracle import job_oracle as jo
from oracle import tad_oracle as orc
from theia_amd import anomaly_detection as ad

CASES = [
    dict(agg_flow=""),
    dict(agg_flow="", start_time="2022-08-11 07:30:00", end_time="2022-08-11 08:00:00"),
    dict(agg_flow="", ns_ignore_list=["kube-system"]),
    dict(agg_flow="svc"),
    dict(agg_flow="svc", svc_port_name="svc-1:http", end_time="2022-08-11 08:00:00"),
    dict(agg_flow="external"),
    dict(agg_flow="external", external_ip="52.1.1.2"),
    dict(agg_flow="pod"),
    dict(agg_flow="pod", pod_label="APP1"),                       # ilike: case-insensitive
    dict(agg_flow="pod", pod_label="app_", pod_namespace="default"),   # '_' is a LIKE wildcard
    dict(agg_flow="pod", pod_name="pod-2"),
    dict(agg_flow="pod", pod_name="pod-2", pod_namespace="flow-visibility", ns_ignore_list=["kube-system"]),
    dict(agg_flow="pod", pod_name="pod-1", start_time="2022-08-11 07:50:00"),  # pod SQL ignores the time window
]


class FakeResult:
    """Shaped like theia_amd.engine.TadResult, filled from the key-id oracle."""

    def __init__(self, want):
        self.n_rows = want["n_anomalies"]
        self._h = {k: want[k] for k in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
        self.stats = {}

    def to_host(self):
        return self._h


def canon(rows):
    return sorted(json.dumps(r, sort_keys=True) for r in rows)


def oracle_middle(prep, algo, agg_flow):
    return orc.run_job(algo, prep.key_id, prep.flow_end_s, prep.value, agg_flow=agg_flow, key_id2=prep.key_id2,
                       flow_start_s=prep.flow_start_s, start_time=prep.start_time, end_time=prep.end_time)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
@pytest.mark.parametrize("algo", ["EWMA", "DBSCAN"])
def test_host_half_matches_string_oracle(case, algo):
    flows = jo.synth_flows(6000)
    kw = dict(start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="",
              svc_port_name="", pod_name="", pod_namespace="")
    kw.update(case)
    prep = ad.prepare_columns(flows, **kw)
    want_mid = oracle_middle(prep, algo, kw["agg_flow"])
    got = ad.result_rows(prep, FakeResult(want_mid), algo, kw["agg_flow"], "id-1")
    want = jo.run(flows, algo, tad_id="id-1", **kw)
    assert len(got) == len(want) and len(got) > 0
    for r in want:   # a 1-point key has a null stddev_samp in Spark; tadetector's Float64 column is not Nullable
        sd = r["throughputStandardDeviation"]       # (create_table.sh:376) -> the engine writes the column default 0.0
        r["throughputStandardDeviation"] = 0.0 if sd is None else float(sd)
    if got[0]["anomaly"] != "NO ANOMALY DETECTED":
        assert canon(got) == canon(want)
    else:
        g, w = dict(got[0]), dict(want[0])
        g.pop("flowStartSeconds")
        assert g == w


def test_generate_tad_sql_query_equals_the_reference_goldens(golden):
    # the reference's own parametrised test (anomaly_detection_test.py:46-195): 12 argument tuples -> exact SQL strings
    assert len(golden["sql_cases"]) == 12
    for case in golden["sql_cases"]:
        start, end, ns, agg, label, ip, svc, name, namespace = case["args"]
        assert ad.generate_tad_sql_query(start, end, ns, agg, label, ip, svc, name, namespace) == case["sql"], case["args"]


def test_sentinel_row_when_nothing_is_anomalous():
    flows = jo.synth_flows(300)
    flows["throughput"] = np.full(300, 1000, dtype=np.uint64)     # constant: sigma 0, |x - e| > 0 only in EWMA warm-up
    prep = ad.prepare_columns(flows, agg_flow="svc")
    want_mid = oracle_middle(prep, "DBSCAN", "svc")
    # every key of this table has >= 4 identical points -> all core -> no noise
    if want_mid["n_anomalies"] == 0:
        rows = ad.result_rows(prep, FakeResult(want_mid), "DBSCAN", "svc", "x")
        assert len(rows) == 1 and rows[0]["anomaly"] == "NO ANOMALY DETECTED" and rows[0]["aggType"] == "svc"
        assert rows[0]["flowEndSeconds"] == 0 and rows[0]["algoCalc"] == 0.0 and rows[0]["sourceIP"] == "None"
    rows = ad.result_rows(prep, FakeResult({"n_anomalies": 0, "key_id": np.zeros(0, np.uint64), "flow_end_s": np.zeros(0, np.int64),
                                            "throughput": np.zeros(0), "algo_calc": np.zeros(0), "stddev": np.zeros(0)}), "EWMA", "", "y")
    assert rows[0]["aggType"] == "None" and rows[0]["algoType"] == "EWMA" and rows[0]["id"] == "y"


def test_remove_meaningless_labels():
    # same cases as the reference's test (anomaly_detection_test.py: labels with generated keys, bad json)
    assert ad.remove_meaningless_labels('{"app":"a","pod-template-hash":"1","controller-revision-hash":"2",'
                                        '"pod-template-generation":"3","z":"y"}') == '{"app": "a", "z": "y"}'
    assert ad.remove_meaningless_labels("not json") == ""
    assert ad.remove_meaningless_labels('{"b":"1","a":"2"}') == '{"a": "2", "b": "1"}'


def test_prepare_columns_key_ids_are_dense_and_rejected_rows_are_skipped():
    flows = jo.synth_flows(2000)
    prep = ad.prepare_columns(flows, agg_flow="external")
    live = prep.key_id != np.uint64(ad.capi.TAD_KEY_SKIP)
    assert (live == (flows["flowType"] == 3)).all()
    assert set(np.unique(prep.key_id[live]).tolist()) == set(range(prep.num_keys))
    assert (prep.key_table["destinationIP"][prep.key_id[live].astype(int)] == flows["destinationIP"][live]).all()
    pod = ad.prepare_columns(flows, agg_flow="pod", pod_name="pod-3")
    assert pod.key_id2 is not None and pod.flow_start_s is None and pod.start_time == 0
    assert ((pod.key_id != np.uint64(ad.capi.TAD_KEY_SKIP)) == (flows["destinationPodName"] == "pod-3")).all()
    assert ((pod.key_id2 != np.uint64(ad.capi.TAD_KEY_SKIP)) == (flows["sourcePodName"] == "pod-3")).all()


def test_cli_rejects_bad_arguments_with_exit_code_2():
    for argv in (["--algo", "LSTM"], ["--start_time", "yesterday"], ["--end_time", "2022-13-01 00:00:00"],
                 ["--ns_ignore_list", '{"a": 1}'], ["--bogus"], ["--flows", "x.npz"], ["--db_jdbc_url", "ftp://x"]):   # no --algo; bad url
        with pytest.raises(SystemExit) as exc:
            ad.main(argv)
        assert exc.value.code == 2


def test_write_result_json_lines():
    buf = io.StringIO()
    rows = [{"destinationServicePortName": "s", "flowEndSeconds": 5, "throughputStandardDeviation": 1.5, "aggType": "svc",
             "algoType": "EWMA", "algoCalc": 2.0, "throughput": 9.0, "anomaly": "true", "id": "abc"}]
    assert ad.write_anomaly_detection_result(rows, buf, tad_id_input="abc") == "abc"
    assert json.loads(buf.getvalue())["destinationServicePortName"] == "s"
    assert len(ad.write_anomaly_detection_result(rows, io.StringIO())) == 36     # generated uuid (ref:715-718)


def test_job_arima_exact_vs_scipy_driven():
    """The GPU path is held bit for bit to oracle/arima_exact.c, which shares tad_detmath.h and the arithmetic contract with the
    kernel by design: a shared mistake in start parameters, Box-Cox or the optimiser would pass those tests.  This one
    holds the exact oracle's JOB against the scipy-driven restatement (real scipy Brent + L-BFGS-B, numpy filter, glibc) on a
    synthetic table: same keys emitted, verdict sets equal up to the flips two optimisers show on flat likelihoods, algoCalc close."""
    from oracle import arima_oracle as ao
    flows = jo.synth_flows(2500, n_buckets=30)
    a = jo.run(flows, "ARIMA", agg_flow="svc", tad_id="x")                                   # calculate_arima_exact (the contract)
    b = jo.run(flows, "ARIMA", agg_flow="svc", tad_id="x", arima_fn=ao.calculate_arima)      # scipy-driven
    ka = {(r["destinationServicePortName"], r["flowEndSeconds"]): r for r in a if r["anomaly"] == "true"}
    kb = {(r["destinationServicePortName"], r["flowEndSeconds"]): r for r in b if r["anomaly"] == "true"}
    assert len(ka) > 20
    both = set(ka) & set(kb)
    # gated at what it measures (39 vs 39 rows, 38 in common; median 3.6e-7, 37 of 38 within 1.1e-4, one flat-likelihood outlier)
    assert len(ka) == len(kb) and len(both) >= len(ka) - 1, (len(ka), len(kb), len(both))     # at most one verdict flip
    rel = np.array([abs(ka[k]["algoCalc"] - kb[k]["algoCalc"]) / abs(kb[k]["algoCalc"]) for k in both])
    assert np.median(rel) <= 1e-6 and (rel < 2e-4).sum() >= len(both) - 1 and (rel < 1e-6).mean() >= 0.6, (np.median(rel), np.sort(rel)[-3:])
    assert all(ka[k]["throughput"] == kb[k]["throughput"] and ka[k]["throughputStandardDeviation"] == kb[k]["throughputStandardDeviation"] for k in both)
