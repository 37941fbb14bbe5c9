"""Whole-job oracle on STRING columns — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A pandas restatement of the reference job from the `flows` table to `tadetector` rows
(/root/reference/plugins/anomaly-detection/anomaly_detection.py): the SQL of generate_tad_sql_query
(:507-614) evaluated with pandas group-bys on the raw string columns, series assembly + stddev_samp
(:664-684), the detectors (:146-349, from oracle/tad_oracle.py + oracle/arima_oracle.py), explode /
filter / sentinel row (:352-421), label canonicalisation after grouping (:686-695).  It shares no code
with theia_amd/anomaly_detection.py (the product's host half), so the two check each other.

PARITY STATUS: the SQL strings are pinned by the reference's goldens (anomaly_detection_test.py:46-195,
checked in tests/test_oracle.py through oracle/ref_loader.py); what the SQL *computes* inside ClickHouse
and Spark's arrays_zip/explode are pinned by no reference unit test ("parity unpinned" at this level,
SURVEY.md §8c) — this file defines them: ascending flowEndSeconds per key, ddof = 1, n = 1 -> null sigma,
ARIMA None -> no rows.
"""
import json
import re
from datetime import datetime, timezone

import numpy as np
import pandas as pd

from . import tad_oracle as orc

MEANINGLESS = {"pod-template-hash", "controller-revision-hash", "pod-template-generation"}  # :139-143


def _epoch(s):
    return int(datetime.strptime(s, "%Y-%m-%d %H:%M:%S").replace(tzinfo=timezone.utc).timestamp())


def _ilike(series, label):
    """ClickHouse ilike(col, '%label%') (:518-521): case-insensitive, % and _ are wildcards."""
    rx = "".join(".*" if c == "%" else "." if c == "_" else re.escape(c) for c in label)
    return series.str.contains(rx, case=False, regex=True)


def canonical_labels(s):
    """remove_meaningless_labels (:631-644)."""
    try:
        d = json.loads(s)
        return json.dumps({k: v for k, v in d.items() if k not in MEANINGLESS}, sort_keys=True)
    except Exception:
        return ""


def stage0_sql(df, start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="",
               svc_port_name="", pod_name="", pod_namespace=""):
    """What the SQL of generate_tad_sql_query returns: one row per (key columns, flowEndSeconds) with the
    aggregated throughput in column `v` (uint64, sum wraps, max unsigned)."""
    ns_ok = pd.Series(True, index=df.index)
    if ns_ignore_list:
        ns_ok = ~df.sourcePodNamespace.isin(list(ns_ignore_list)) & ~df.destinationPodNamespace.isin(list(ns_ignore_list))

    def agg(frame, keys, op):
        if frame.empty:
            return pd.DataFrame(columns=keys + ["flowEndSeconds", "v"])
        g = frame.groupby(keys + ["flowEndSeconds"], sort=True)["throughput"]
        if op == "max":
            out = g.max()
        else:  # wrapping uint64 sum
            out = g.apply(lambda s: np.add.reduce(s.to_numpy(dtype=np.uint64), dtype=np.uint64))
        return out.rename("v").reset_index()

    if agg_flow == "pod":
        parts = []
        ident = "PodLabels" if (pod_label or not pod_name) else "PodName"
        out_ident = "podLabels" if ident == "PodLabels" else "podName"
        for side, direction in (("destination", "inbound"), ("source", "outbound")):
            col, ns = df[side + ident], df[side + "PodNamespace"]
            if pod_label:
                cond = _ilike(col, pod_label)
                if pod_namespace:
                    cond &= ns == pod_namespace
            elif pod_name:
                cond = col == pod_name
                if pod_namespace:
                    cond &= ns == pod_namespace
            else:
                cond = col != ""
            sub = df[cond & ns_ok]
            frame = pd.DataFrame({"podNamespace": sub[side + "PodNamespace"], out_ident: sub[side + ident],
                                  "direction": direction, "flowEndSeconds": sub.flowEndSeconds, "throughput": sub.throughput})
            parts.append(agg(frame, ["podNamespace", out_ident, "direction"], "sum"))
        return pd.concat(parts, ignore_index=True), ["podNamespace", out_ident, "direction"]

    cond = ns_ok.copy()
    if start_time:
        cond &= df.flowStartSeconds >= _epoch(start_time)
    if end_time:
        cond &= df.flowEndSeconds < _epoch(end_time)
    if agg_flow == "external":
        cond &= df.flowType == 3
        if external_ip:
            cond &= df.destinationIP == external_ip
        keys, op = ["destinationIP"], "sum"
    elif agg_flow == "svc":
        cond &= (df.destinationServicePortName == svc_port_name) if svc_port_name else (df.destinationServicePortName != "")
        keys, op = ["destinationServicePortName"], "sum"
    else:
        keys, op = ["sourceIP", "sourceTransportPort", "destinationIP", "destinationTransportPort", "protocolIdentifier",
                    "flowStartSeconds"], "max"
    return agg(df[cond], keys, op), keys


def run(flows, algo, start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="",
        svc_port_name="", pod_name="", pod_namespace="", tad_id="", arima_fn=None):
    """-> list of tadetector rows (dicts), in (key columns, flowEndSeconds) order of the pandas group-by."""
    df = pd.DataFrame({k: np.asarray(v) for k, v in flows.items()})
    pts, keys = stage0_sql(df, start_time, end_time, ns_ignore_list, agg_flow, pod_label, external_ip, svc_port_name,
                           pod_name, pod_namespace)
    rows = []
    agg_type = agg_flow if agg_flow else "None"
    if arima_fn is None and algo == "ARIMA":
        from .arima_oracle import calculate_arima_exact as arima_fn   # the fixed-arithmetic restatement
    for key_vals, grp in (pts.groupby(keys, sort=True) if len(pts) else []):
        grp = grp.sort_values("flowEndSeconds")
        if not isinstance(key_vals, tuple):
            key_vals = (key_vals,)
        x = grp.v.to_numpy(dtype=np.uint64)
        xf = orc.u64_to_f64(x)
        sd = orc.stddev_samp_series(xf)
        if algo == "EWMA":
            calc = orc.calculate_ewma([int(v) for v in x])
            verdict = orc.calculate_ewma_anomaly([int(v) for v in x], sd)
        elif algo == "DBSCAN":
            calc = [0.0] * len(x)
            verdict = orc.dbscan_noise_1d(xf).tolist()
        else:
            calc = arima_fn(x)
            if calc is None:
                continue                       # [False] zipped with a null algoCalc array: no rows (:284-287)
            verdict = [False] * len(x) if sd is None else [abs(float(a) - p) > sd for a, p in zip(xf, calc)]
        for t, xv, c, a in zip(grp.flowEndSeconds.tolist(), xf.tolist(), calc, verdict):
            if not a:
                continue
            row = dict(zip(keys, key_vals))
            if "podLabels" in row:
                row["podLabels"] = canonical_labels(row["podLabels"])
            for k, v in list(row.items()):
                if isinstance(v, (np.integer,)):
                    row[k] = int(v)
            row.update({"flowEndSeconds": int(t), "throughputStandardDeviation": sd, "aggType": agg_type,
                        "algoType": algo, "algoCalc": float(c), "throughput": float(xv), "anomaly": "true", "id": str(tad_id)})
            rows.append(row)
    if not rows:
        rows.append({"sourceIP": "None", "sourceTransportPort": 0, "destinationIP": "None", "destinationTransportPort": 0,
                     "protocolIdentifier": 0, "podNamespace": "None", "podLabels": "None", "podName": "None",
                     "destinationServicePortName": "None", "direction": "None", "flowEndSeconds": 0,
                     "throughputStandardDeviation": 0, "aggType": agg_type, "algoType": algo, "algoCalc": 0.0,
                     "throughput": 0.0, "anomaly": "NO ANOMALY DETECTED", "id": str(tad_id)})
    return rows


def synth_flows(n_rows, seed=7, n_pods=6, n_svc=4, n_buckets=40):
    """A small deterministic `flows` table with every column the job touches (create_table.sh:31-85)."""
    rng = np.random.default_rng(seed)
    ns = np.array(["default", "kube-system", "flow-visibility", "prod"])
    pods = np.array(["pod-%d" % i for i in range(n_pods)])
    pod_ns = ns[np.arange(n_pods) % 3 if n_pods else 0]
    labels = np.array([json.dumps({"app": "app%d" % (i % 3), "pod-template-hash": "h%d" % i, "tier": "T%d" % (i % 2)})
                       for i in range(n_pods)] + [""])
    src = rng.integers(0, n_pods, n_rows)
    dst = rng.integers(0, n_pods + 1, n_rows)          # index n_pods = traffic to an external address
    ext = dst == n_pods
    dsti = np.minimum(dst, n_pods - 1)
    bucket = rng.integers(0, n_buckets, n_rows)
    t_end = 1660202814 + 60 * bucket
    svc_names = np.array([""] + ["svc-%d:http" % i for i in range(n_svc)])
    base = 1_000_000_000 + 250_000_000 * (src + 1)
    val = (base + rng.integers(-1_000_000, 1_000_000, n_rows)).astype(np.uint64)
    spike = rng.random(n_rows) < 0.01
    val = np.where(spike, val * np.uint64(9), val)
    return {
        "flowStartSeconds": (t_end - rng.integers(1, 4, n_rows) * 30).astype(np.int64),
        "flowEndSeconds": t_end.astype(np.int64),
        "sourceIP": np.char.add("10.0.0.", (src + 1).astype(str)),
        "destinationIP": np.where(ext, np.char.add("52.1.1.", (bucket % 3 + 1).astype(str)), np.char.add("10.0.0.", (dsti + 1).astype(str))),
        "sourceTransportPort": (40000 + src * 7 + rng.integers(0, 2, n_rows)).astype(np.int64),
        "destinationTransportPort": np.where(ext, 443, 8080 + dsti % 2).astype(np.int64),
        "protocolIdentifier": np.full(n_rows, 6, dtype=np.int64),
        "sourcePodName": pods[src], "sourcePodNamespace": pod_ns[src],
        "destinationPodName": np.where(ext, "", pods[dsti]), "destinationPodNamespace": np.where(ext, "", pod_ns[dsti]),
        "destinationServicePortName": np.where(ext, "", svc_names[(dsti % (n_svc + 1))]),
        "flowType": np.where(ext, 3, 1).astype(np.int64),
        "sourcePodLabels": labels[src], "destinationPodLabels": np.where(ext, "", labels[dsti]),
        "throughput": val,
    }
