"""Host-side mirror of the reference's abnormal-traffic-drop UDF, computed on the MI355X engine.

Reference: /root/reference/snowflake/udfs/udfs/drop_detection/drop_detection_udf.py (cited as `ref:`) — a Snowflake UDTF
partitioned by (endpoint, direction): `process` collects (date, drop_number) pairs (ref:25-40), `end_partition` yields one
row per anomalous day (ref:42-56).  Same class and method names here; `end_partition` calls tad_series_drop through the C
ABI, `drop_detection_table` runs every partition of an aggregated table in ONE tad_run (algo DROP).  No CPU fallback.
Only the UDTF's call protocol is mirrored (a thin adaptor); the reference's Result helper class is not reproduced.
"""
import datetime
import uuid

import numpy as np

from . import anomaly_detection as _ad


RESULT_COLUMNS = ("job_type", "detection_id", "time_created", "endpoint", "direction", "avg_drop", "stdev_drop",
                  "anomaly_drop_date", "anomaly_drop_number")      # the UDTF's output row (ref:6-19, 55-56)


class DropDetection:
    """Thin adaptor with the UDTF's call protocol (ref:21-56: one instance per (endpoint, direction) partition,
    `process` once per row, `end_partition` yields the anomalous days as RESULT_COLUMNS tuples).  The partition's
    statistics and verdicts come from the engine (tad_series_drop); nothing is computed here."""

    def __init__(self, engine=None):
        self._engine = engine
        self._partition = None            # (job_type, detection_id, endpoint, direction), constant within a partition
        self._dates, self._drops = [], []

    def process(self, job_type, detection_id, endpoint, direction, date, drop_number):
        if job_type != "initial":         # ref:33
            raise AssertionError("drop detection supports job_type 'initial' only")
        self._partition = (job_type, detection_id, endpoint, direction)
        self._dates.append(date)
        self._drops.append(int(drop_number))
        yield None

    def end_partition(self):
        if len(self._drops) < 3:          # ref:44-45
            return
        out = (self._engine or _ad.get_engine()).series_drop(self._drops)
        if out is None:
            return
        mean, std, verdict = out
        job_type, detection_id, endpoint, direction = self._partition
        for i in np.flatnonzero(verdict):   # a fresh id PER ROW when none was given, as Result.__init__ does (ref:8-11)
            yield (job_type, detection_id or str(uuid.uuid4()), datetime.datetime.now(), endpoint, direction, mean, std,
                   self._dates[i], self._drops[i])


def drop_detection_table(endpoint, direction, date, drop_number, detection_id=None, job_type="initial", engine=None):
    """All partitions at once: columns of the `aggregated_flows` CTE (snowflake/cmd/dropDetection.go:151-162) ->
    list of result tuples in (endpoint, direction, date) order.  `date` may be strings (YYYY-MM-DD) or day numbers."""
    import pandas as pd
    endpoint, direction = np.asarray(endpoint).astype(str), np.asarray(direction).astype(str)
    codes, uniq = pd.MultiIndex.from_arrays([endpoint, direction]).factorize()
    d = np.asarray(date)
    if d.dtype.kind in "USO":
        day = np.asarray(pd.to_datetime(d).values.astype("datetime64[D]").astype(np.int64))
    else:
        day = d.astype(np.int64)
    eng = engine or _ad.get_engine()
    # one lattice bucket per day; Stage 0 sums the drop numbers of equal (key, day)
    res = eng.run("DROP", codes.astype(np.uint64), day, np.asarray(drop_number, dtype=np.uint64), max(len(uniq), 1),
                  agg_flow="svc", value_op="sum")
    now = datetime.datetime.now()
    host = res.to_host()
    rows = []
    for k, t, x, mean, std in zip(host["key_id"].tolist(), host["flow_end_s"].tolist(), host["throughput"].tolist(),
                                  host["algo_calc"].tolist(), host["stddev"].tolist()):
        ep, di = uniq[int(k)]
        dd = str(np.datetime64(int(t), "D")) if d.dtype.kind in "USO" else int(t)
        rows.append((job_type, detection_id or str(uuid.uuid4()), now, ep, di, mean, std, dd, int(x)))   # (ref:8-11: id per row)
    return rows
