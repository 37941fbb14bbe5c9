"""Host-side mirror of the reference's abnormal-traffic-drop UDF, computed on the MI355X engine.

Reference: /root/reference/snowflake/udfs/udfs/drop_detection/drop_detection_udf.py (cited as `ref:`) — a Snowflake UDTF
partitioned by (endpoint, direction): `process` collects (date, drop_number) pairs (ref:25-40), `end_partition` yields one
row per anomalous day (ref:42-56).  Same class and method names here; `end_partition` calls tad_series_drop through the C
ABI, `drop_detection_table` runs every partition of an aggregated table in ONE tad_run (algo DROP).  No CPU fallback.
"""
import datetime
import uuid

import numpy as np

from . import anomaly_detection as _ad


class Result:   # ref:6-19
    def __init__(self, job_type, detection_id, endpoint, direction, avg_drop, stdev_drop, anomaly_drop_date, anomaly_drop_number):
        self.job_type = job_type
        self.detection_id = detection_id if detection_id else str(uuid.uuid4())
        self.time_created = datetime.datetime.now()
        self.endpoint = endpoint
        self.direction = direction
        self.avg_drop = avg_drop
        self.stdev_drop = stdev_drop
        self.anomaly_drop_date = anomaly_drop_date
        self.anomaly_drop_number = anomaly_drop_number


class DropDetection:
    def __init__(self, engine=None):
        self._date_dropnumber_pairs = []
        self._engine = engine

    def process(self, job_type, detection_id, endpoint, direction, date, drop_number):   # ref:25-40
        assert job_type == "initial"
        self._job_type = job_type
        self._detection_id = detection_id
        self._endpoint = endpoint
        self._direction = direction
        self._date_dropnumber_pairs.append((date, drop_number))
        yield None

    def end_partition(self):   # ref:42-56
        pairs = self._date_dropnumber_pairs
        if len(pairs) < 3:
            return
        eng = self._engine or _ad.get_engine()
        out = eng.series_drop([int(n) for _, n in pairs])
        if out is None:
            return
        mean, std, verdict = out
        for (date, drop_number), bad in zip(pairs, verdict.tolist()):
            if bad:
                row = Result(self._job_type, self._detection_id, self._endpoint, self._direction, mean, std, date, drop_number)
                yield (row.job_type, row.detection_id, row.time_created, row.endpoint, row.direction, row.avg_drop,
                       row.stdev_drop, row.anomaly_drop_date, row.anomaly_drop_number)


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
    det = detection_id if detection_id else str(uuid.uuid4())
    now = datetime.datetime.now()
    host = res.to_host()
    rows = []
    for k, t, x, mean, std in zip(host["key_id"].tolist(), host["flow_end_s"].tolist(), host["throughput"].tolist(),
                                  host["algo_calc"].tolist(), host["stddev"].tolist()):
        ep, di = uniq[int(k)]
        dd = str(np.datetime64(int(t), "D")) if d.dtype.kind in "USO" else int(t)
        rows.append((job_type, det, now, ep, di, mean, std, dd, int(x)))
    return rows
