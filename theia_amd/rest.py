"""The read-back side of the job: `tadetector` rows -> ThroughputAnomalyDetectorStats (SURVEY.md §3.3, §8f rank 2's anchor rest.go:249-315).

What `theia throughput-anomaly-detection retrieve` shows is produced by the API server's REST handler
(pkg/apiserver/registry/intelligence/throughputanomalydetector/rest.go): `Get` / `List` copy the custom resource into the API type and,
for a COMPLETED job, `getTADetectorResult` picks one of five SELECTs by `AggregatedFlow` / `PodName` (rest.go:59-123, 249-263) and scans
every row of the job into a `ThroughputAnomalyDetectorStats` whose fields are ALL strings (pkg/apis/intelligence/v1alpha1/types.go:108-126);
the CLI prints them as one table per aggregation type or the "No Anomaly found" line (pkg/theia/commands/anomaly_detection_retrieve.go:94-137).
The engine writes the rows (theia_amd/anomaly_detection.py:store_result_columns); this module is the other end, so that a job can be checked
in the consumer's own format: the strings Go's database/sql produces when it scans a ClickHouse column into a `string` (`convertAssign`:
integers in base 10, float64 through strconv.FormatFloat(v, 'g', -1, 64) — 4005703059 reads "4.005703059e+09", which is where the five-character
prefixes of the reference's e2e result map come from, test/e2e/throughputanomalydetection_test.go:191-221 — and DateTime as RFC 3339).

Scope (round 5): ONLY `get_tad_result` and the Go string forms it needs — they pin the throughput prefixes of the reference's e2e result
map on rows the engine wrote.  The REST verbs (Create / Delete / List), the API type and the CLI's `retrieve` table / `--file` JSON that round 4
also restated are SURVEY.md section 2 #6 / #7, OUT OF SCOPE, and were removed again.
"""
import math
from dataclasses import dataclass
from datetime import datetime, timezone

import numpy as np

@dataclass
class ThroughputAnomalyDetectorStats:
    """pkg/apis/intelligence/v1alpha1/types.go:108-126: every field a string, "" = not selected for this aggregation type."""
    id: str = ""
    sourceIP: str = ""
    sourceTransportPort: str = ""
    destinationIP: str = ""
    destinationTransportPort: str = ""
    flowStartSeconds: str = ""
    podNamespace: str = ""
    podLabels: str = ""
    podName: str = ""
    direction: str = ""
    destinationServicePortName: str = ""
    flowEndSeconds: str = ""
    throughput: str = ""
    aggType: str = ""
    algoType: str = ""
    algoCalc: str = ""
    anomaly: str = ""


# The five SELECTs of rest.go:59-123 (queryMap), column for column.  The reference binds the id with database/sql's `?`; the HTTP
# interface binds a named parameter, `{id:String}` + `param_id` in the URL — the value never enters the SQL text.
_COLUMNS = {
    "tad": ["id", "sourceIP", "sourceTransportPort", "destinationIP", "destinationTransportPort", "flowStartSeconds", "flowEndSeconds",
            "throughput", "aggType", "algoType", "algoCalc", "anomaly"],
    "external": ["id", "destinationIP", "flowEndSeconds", "throughput", "aggType", "algoType", "algoCalc", "anomaly"],
    "podLabel": ["id", "podNamespace", "podLabels", "direction", "flowEndSeconds", "throughput", "aggType", "algoType", "algoCalc", "anomaly"],
    "podName": ["id", "podNamespace", "podName", "direction", "flowEndSeconds", "throughput", "aggType", "algoType", "algoCalc", "anomaly"],
    "svc": ["id", "destinationServicePortName", "flowEndSeconds", "throughput", "aggType", "algoType", "algoCalc", "anomaly"],
}


def query_kind(agg_flow, pod_name):
    """getTADetectorResult's switch (rest.go:249-263)."""
    if agg_flow == "external":
        return "external"
    if agg_flow == "pod":
        return "podName" if pod_name != "" else "podLabel"
    if agg_flow == "svc":
        return "svc"
    return "tad"


def result_columns(agg_flow, pod_name=""):
    return list(_COLUMNS[query_kind(agg_flow, pod_name)])


def reference_result_query(agg_flow, pod_name=""):
    """queryMap's statement with the reference's positional placeholder (what rest_test.go:104-131 matches with QueryMatcherEqual,
    up to its line breaks)."""
    return "SELECT " + ", ".join(result_columns(agg_flow, pod_name)) + " FROM tadetector WHERE id = (?);"


def result_query(agg_flow, pod_name=""):
    return "SELECT " + ", ".join(result_columns(agg_flow, pod_name)) + " FROM tadetector WHERE id = ({id:String})"


def go_format_float(v):
    """strconv.FormatFloat(v, 'g', -1, 64) — what database/sql's convertAssign applies to a float64 scanned into a string: the shortest
    digits that round-trip, in %e form (one digit, '.', the rest, exponent sign and at least two exponent digits) when the decimal exponent
    is < -4 or >= 6 (ftoa.go: `if shortest { eprec = 6 }`), in plain %f form with exactly the digits needed otherwise."""
    from decimal import Decimal
    v = float(v)
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "+Inf" if v > 0 else "-Inf"
    if v == 0.0:
        return "-0" if math.copysign(1.0, v) < 0 else "0"
    _, dig, exp = Decimal(repr(abs(v))).as_tuple()        # Python's repr is the shortest round-trip representation as well
    digits = "".join(map(str, dig)).lstrip("0")
    stripped = digits.rstrip("0")
    exp += len(digits) - len(stripped)
    digits = stripped
    nd, dp = len(digits), len(digits) + exp                 # value = 0.d1d2...dnd x 10^dp
    x = dp - 1
    if x < -4 or x >= 6:
        s = digits[0] + ("." + digits[1:] if nd > 1 else "") + "e" + ("-" if x < 0 else "+") + ("%02d" % abs(x))
    elif dp <= 0:
        s = "0." + "0" * (-dp) + digits
    elif nd <= dp:
        s = digits + "0" * (dp - nd)
    else:
        s = digits[:dp] + "." + digits[dp:]
    return "-" + s if v < 0 else s


def go_string(v):
    """database/sql convertAssign into a *string: string as is, []byte decoded, integers base 10, float64 'g' / -1, bool "true" / "false",
    time.Time in RFC 3339 with nanoseconds trimmed (ClickHouse DateTime has none)."""
    if v is None:
        return ""
    if isinstance(v, (bytes, bytearray)):
        return bytes(v).decode()
    if isinstance(v, str):
        return v
    if isinstance(v, (bool, np.bool_)):
        return "true" if v else "false"
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    if isinstance(v, (float, np.floating)):
        return go_format_float(v)
    if isinstance(v, np.datetime64):
        v = datetime.fromtimestamp(int(v.astype("datetime64[s]").astype(np.int64)), timezone.utc)
    if isinstance(v, datetime):
        if v.tzinfo is None:
            v = v.replace(tzinfo=timezone.utc)
        u = v.astimezone(timezone.utc)
        frac = ("%06d" % u.microsecond).rstrip("0")
        return u.strftime("%Y-%m-%dT%H:%M:%S") + ("." + frac if frac else "") + "Z"
    return str(v)


_TIME_COLUMNS = ("flowStartSeconds", "flowEndSeconds")


def get_tad_result(client, job_id, agg_flow="", pod_name=""):
    """getTADetectorResult (rest.go:249-315): the rows of job `job_id` as ThroughputAnomalyDetectorStats, in the order the server returns
    them.  `client` = theia_amd.clickhouse.ClickHouseHTTP (query_columns with a bound `id`).  Errors carry the reference's wording."""
    cols = result_columns(agg_flow, pod_name)
    try:
        got = client.query_columns(result_query(agg_flow, pod_name), params={"id": job_id})
    except Exception as exc:
        raise RuntimeError("failed to get Throughput Anomaly Detector results with id %s: %s" % (job_id, exc))
    if not got:
        return []
    what = {"tad": "", "external": " External IP Aggregate", "podLabel": " Pod Aggregate", "podName": " Pod Aggregate", "svc": " Service Aggregate"}
    missing = [c for c in cols if c not in got]
    if missing:
        raise RuntimeError("failed to scan Throughput Anomaly Detector%s results: missing columns %s" % (what[query_kind(agg_flow, pod_name)], missing))
    n = len(got[cols[0]])
    stats = []
    for i in range(n):
        s = ThroughputAnomalyDetectorStats()
        for c in cols:
            v = got[c][i]
            if c in _TIME_COLUMNS and isinstance(v, (int, np.integer)):       # query_columns hands DateTime over as epoch seconds
                v = datetime.fromtimestamp(int(v), timezone.utc)
            setattr(s, c, go_string(v))
        stats.append(s)
    return stats
