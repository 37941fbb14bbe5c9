"""Host-side mirror of the reference job's interface for the Throughput Anomaly Detection path.

Reference: /root/reference/plugins/anomaly-detection/anomaly_detection.py (cited below as `ref:`).
Same function names, argument meaning and error behaviour as the reference module, so that the
parity tests read like the reference's own (anomaly_detection_test.py) — but every number comes
from the HIP kernels in libtad_mi355x.so through the C ABI (include/tad.h).  There is no CPU
fallback here and nothing under oracle/ is imported: without the library or a GPU the compute
entry points raise TadError / OSError.

What stays on the host (as SURVEY.md §8b assigns it): the string work of the SQL the reference
pushes into ClickHouse (ref:507-614) — evaluating the WHERE predicates on the string columns and
dictionary-encoding the mode's key columns into dense uint64 ids — and the expansion of the
engine's (key_id, flowEndSeconds, ...) result rows back into `tadetector` rows (ref:352-421,
create_table.sh:363-384).  The engine sees integers only.

Flow tables are dicts of equally long numpy arrays named like the ClickHouse columns
(create_table.sh:31-85); DateTime columns are epoch seconds (int64).  `load_flows` reads such a
table from .npz / .parquet / .csv.
"""
import getopt
import json
import logging
import re
import sys
import time
import uuid
from datetime import datetime, timezone

import numpy as np

from . import _capi as capi
from .engine import TadEngine, TadError

logger = logging.getLogger("anomaly_detection")

table_name = "default.flows"                      # ref:48
RESULT_TABLE_NAME = "default.tadetector"          # ref:732
VALID_ALGOS = ("EWMA", "ARIMA", "DBSCAN")         # ref:816
VALID_AGG_FLOWS = ("", "pod", "external", "svc")  # controller.go:560-620
MEANINGLESS_LABELS = ("pod-template-hash", "controller-revision-hash", "pod-template-generation")  # ref:139-143
TIME_FORMAT = "%Y-%m-%d %H:%M:%S"                 # ref:833, pkg/controller/util.go:45

# key columns per mode, in result-row order (ref:109-137)
KEY_COLUMNS = {
    "": ("sourceIP", "sourceTransportPort", "destinationIP", "destinationTransportPort", "protocolIdentifier",
         "flowStartSeconds"),
    "external": ("destinationIP",),           # flowType is part of the GROUP BY (ref:130-133) but constant 3 and dropped (ref:621)
    "svc": ("destinationServicePortName",),
    "pod": ("podNamespace", "podLabels", "direction"),
    "podname": ("podNamespace", "podName", "direction"),
}

_engine = None


def get_engine():
    """The process-wide engine on HIP device 0 (created on first use)."""
    global _engine
    if _engine is None:
        _engine = TadEngine(device=0)
    return _engine


def set_engine(engine):
    global _engine
    _engine = engine


# ------------------------------------------------------------------------------------------------
# the reference's pure per-series functions (ref:146-349), computed by the GPU kernels
# ------------------------------------------------------------------------------------------------
def _as_u64_list(values):
    return np.asarray([int(v) for v in values], dtype=np.uint64)


def calculate_ewma(throughput_list):
    """ref:146-165 -> list of float, e_t = 0.5 e_{t-1} + 0.5 float(x_t), e_{-1} = 0."""
    return get_engine().series_ewma(_as_u64_list(throughput_list)).tolist()


def calculate_ewma_anomaly(throughput_row, stddev):
    """ref:168-212 -> list of bool; stddev None -> all False."""
    return get_engine().series_ewma_anomaly(_as_u64_list(throughput_row), stddev).tolist()


def calculate_arima(throughputs):
    """ref:215-264 -> list of float, or None (len <= 3, non-positive or constant data: Box-Cox raises)."""
    out = get_engine().series_arima(_as_u64_list(throughputs))
    return None if out is None else out.tolist()


def calculate_arima_anomaly(throughput_row, stddev):
    """ref:267-309 -> list of bool; [False] when calculate_arima returns None (ref:284-287)."""
    return get_engine().series_arima_anomaly(_as_u64_list(throughput_row), stddev).tolist()


def calculate_dbscan(throughput_list):
    """ref:312-322: the algoCalc placeholder, 0.0 per point (no device work to do)."""
    return [0.0] * len(throughput_list)


def calculate_dbscan_anomaly(throughput_row, stddev=None):
    """ref:325-349 -> list of bool (label == -1); stddev is ignored, as in the reference."""
    return get_engine().series_dbscan_anomaly(_as_u64_list(throughput_row)).tolist()


def remove_meaningless_labels(podLabels):
    """ref:631-644: drop the controller-generated labels, re-serialise with sorted keys; bad JSON -> ""."""
    try:
        labels = json.loads(podLabels)
        kept = {k: v for k, v in labels.items() if k not in MEANINGLESS_LABELS}
    except Exception as exc:  # same catch-all as the reference
        logger.error("Error %s: labels %s are not in json format", exc, podLabels)
        return ""
    return json.dumps(kept, sort_keys=True)


# ------------------------------------------------------------------------------------------------
# The SQL of the reference job (ref:507-614), for a host that lets ClickHouse evaluate the WHERE clause and stream
# the raw columns (or, as the reference does, the aggregated points).  String-identical to the reference's
# generate_tad_sql_query — pinned by its 12 golden queries (anomaly_detection_test.py:46-195) in tests/test_host_job.py.
# prepare_columns below evaluates the same predicates on a column dict.
# ------------------------------------------------------------------------------------------------
_SELECT = {   # mode -> (select list, group-by columns)
    "": ("sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, protocolIdentifier, flowStartSeconds, "
         "flowEndSeconds, max(throughput)",
         "sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, protocolIdentifier, flowStartSeconds"),
    "external": ("destinationIP, flowType, flowEndSeconds, sum(throughput)", "destinationIP, flowType"),
    "svc": ("destinationServicePortName, flowEndSeconds, sum(throughput)", "destinationServicePortName"),
}


def _quoted_list(names):
    return ", ".join("'{}'".format(x) for x in names)


def generate_tad_sql_query(start_time, end_time, ns_ignore_list, agg_flow=None, pod_label=None, external_ip=None,
                           svc_port_name=None, pod_name=None, pod_namespace=None):
    ns_clause = ""
    if ns_ignore_list:
        ns_clause = "sourcePodNamespace NOT IN ({0}) AND destinationPodNamespace NOT IN ({0})".format(_quoted_list(ns_ignore_list))
    if agg_flow == "pod":
        ident, out = ("PodName", "podName") if (pod_name and not pod_label) else ("PodLabels", "podLabels")
        halves = []
        for side, direction in (("destination", "inbound"), ("source", "outbound")):
            if pod_label:
                cond = "ilike({}PodLabels, '%{}%')".format(side, pod_label) + (" " if side == "destination" else "")
            elif pod_name:
                cond = "{}PodName = '{}'".format(side, pod_name)
            else:
                cond = "{}PodLabels <> ''".format(side) + (" " if side == "destination" else "")
            if (pod_label or pod_name) and pod_namespace:
                cond += " AND {}PodNamespace = '{}'".format(side, pod_namespace)
            halves.append("(SELECT {side}PodNamespace AS podNamespace, {side}{ident} AS {out}, '{direction}' AS direction, "
                          "flowEndSeconds, sum(throughput) FROM {table} WHERE {cond} {ext} GROUP BY podNamespace, {out}, "
                          "direction, flowEndSeconds)".format(side=side, ident=ident, out=out, direction=direction,
                                                              table=table_name, cond=cond,
                                                              ext=("AND " + ns_clause) if ns_clause else ""))
        return "SELECT * FROM " + halves[0] + " UNION ALL " + halves[1] + " "
    mode = agg_flow if agg_flow in ("external", "svc") else ""
    select, group = _SELECT[mode]
    where = [ns_clause] if ns_clause else []
    if start_time:
        where.append("flowStartSeconds >= '{}'".format(start_time))
    if end_time:
        where.append("flowEndSeconds < '{}'".format(end_time))
    if mode == "external":
        where.append("flowType = 3")
        if external_ip:
            where.append("destinationIP = '{}'".format(external_ip))
    elif mode == "svc":
        where.append("destinationServicePortName = '{}'".format(svc_port_name) if svc_port_name
                     else "destinationServicePortName <> ''")
    sql = "SELECT {} FROM {} ".format(select, table_name)
    if where:
        sql += "WHERE " + " AND ".join(where) + " "
    return sql + "GROUP BY {}, flowEndSeconds ".format(group)


# ------------------------------------------------------------------------------------------------
# Stage 0, host half: WHERE predicates + dictionary encoding (ref:507-614)
# ------------------------------------------------------------------------------------------------
def _epoch(ts):
    """'YYYY-MM-DD hh:mm:ss' (UTC) -> epoch seconds; '' / None -> 0."""
    if not ts:
        return 0
    return int(datetime.strptime(ts, TIME_FORMAT).replace(tzinfo=timezone.utc).timestamp())


def _like_regex(pattern):
    """ClickHouse LIKE pattern -> compiled case-insensitive regex (ilike, ref:518-521)."""
    out, i = [], 0
    while i < len(pattern):
        c = pattern[i]
        if c == "\\" and i + 1 < len(pattern):
            out.append(re.escape(pattern[i + 1]))
            i += 2
            continue
        out.append(".*" if c == "%" else "." if c == "_" else re.escape(c))
        i += 1
    return re.compile("^" + "".join(out) + "$", re.IGNORECASE | re.DOTALL)


class DictColumn:
    """A string column as (codes[N], values[D]): row i holds values[codes[i]].  This is what Arrow dictionary arrays and
    ClickHouse's LowCardinality columns are, and what theia_amd.clickhouse.query_columns(dict_strings=True) delivers: the host
    then evaluates the SQL's string predicates on the D distinct values and touches the N rows with integer operations only
    (prepare_columns on plain string arrays runs at 1-3e6 rows/s — DESIGN.md section 5)."""

    def __init__(self, codes, values):
        self.codes = np.asarray(codes)
        self.values = np.asarray(values).astype(str)

    def __len__(self):
        return self.codes.size

    def materialise(self):
        return self.values[self.codes] if self.values.size else np.zeros(self.codes.size, dtype=str)

    def take(self, sel):
        return DictColumn(self.codes[sel], self.values)

    def per_row(self, value_mask):
        """bool[D] over the distinct values -> bool[N] over the rows"""
        return np.asarray(value_mask, dtype=bool)[self.codes] if self.values.size else np.zeros(self.codes.size, dtype=bool)

    @staticmethod
    def concatenate(cols):
        """rows of several DictColumns over ONE unified dictionary (distinct strings merged)"""
        values, inv = np.unique(np.concatenate([c.values for c in cols]), return_inverse=True)
        out, at = [], 0
        for c in cols:
            out.append(inv[at:at + c.values.size][c.codes] if c.values.size else np.zeros(0, dtype=np.int64))
            at += c.values.size
        return DictColumn(np.concatenate(out), values)


def _str_col(flows, name):
    c = flows[name]
    return c if isinstance(c, DictColumn) else np.asarray(c).astype(str)


def _eq(col, s):
    return col.per_row(col.values == s) if isinstance(col, DictColumn) else col == s


def _isin(col, lst):
    return col.per_row(np.isin(col.values, lst)) if isinstance(col, DictColumn) else np.isin(col, lst)


def _take(col, sel):
    return col.take(sel) if isinstance(col, DictColumn) else col[sel]


def _concat(cols):
    return DictColumn.concatenate(cols) if any(isinstance(c, DictColumn) for c in cols) else np.concatenate(cols)


def _ilike_contains(col, needle):
    rx = _like_regex("%" + needle + "%")
    if isinstance(col, DictColumn):
        return col.per_row(np.fromiter((rx.match(u) is not None for u in col.values), dtype=bool, count=col.values.size))
    uniq, inv = np.unique(col, return_inverse=True)
    hit = np.fromiter((rx.match(u) is not None for u in uniq), dtype=bool, count=uniq.size)
    return hit[inv]


class PreparedColumns:
    """What tad_run needs for one job: the encoded columns + the dictionary to decode results."""

    def __init__(self, mode, key_id, key_id2, flow_end_s, flow_start_s, value, key_table, start_time, end_time, key_hist=None):
        self.mode = mode                  # key of KEY_COLUMNS
        self.key_id = key_id              # u64[N], TAD_KEY_SKIP where the predicates reject the row
        self.key_id2 = key_id2            # pod mode: the row's outbound key (ref:556-565), else None
        self.flow_end_s = flow_end_s      # i64[N]
        self.flow_start_s = flow_start_s  # i64[N] or None (pod mode: the SQL has no time filter)
        self.value = value                # u64[N] throughput
        self.key_table = key_table        # dict column name -> array[num_keys] (device ingest: DeviceKeyColumn, decoded on demand)
        self.start_time = start_time      # epoch seconds handed to the engine (0 = unset)
        self.end_time = end_time
        self.key_hist = key_hist          # device ingest: tad_factorize_hist's by-product for these very columns (engine.run(key_hist=...))

    @property
    def num_keys(self):
        return len(next(iter(self.key_table.values()))) if self.key_table else 0


def _factorize(columns):
    """Rows of `columns` (list of equally long arrays or DictColumns) -> (codes int64[N], list of unique-value arrays), ids in
    order of first appearance.  DictColumns take part with their integer codes; their strings are only looked up for the
    distinct keys at the end."""
    import pandas as pd
    n = len(columns[0])
    if n == 0:
        return np.zeros(0, dtype=np.int64), [(c.values[:0] if isinstance(c, DictColumn) else np.asarray(c)[:0]) for c in columns]
    raw = [c.codes if isinstance(c, DictColumn) else np.asarray(c) for c in columns]
    codes, uniques = pd.MultiIndex.from_arrays(raw).factorize() if len(raw) > 1 else pd.factorize(raw[0])
    if len(raw) > 1:
        uniq_cols = [np.asarray(uniques.get_level_values(i)) for i in range(len(raw))]
    else:
        uniq_cols = [np.asarray(uniques)]
    uniq_cols = [(c.values[u] if isinstance(c, DictColumn) else u) for c, u in zip(columns, uniq_cols)]
    return np.asarray(codes, dtype=np.int64), uniq_cols


def _int_codes(col):
    """a key column as int64 per row: DictColumn codes, integers as they are, plain strings through their sorted distinct values"""
    if isinstance(col, DictColumn):
        return np.ascontiguousarray(col.codes, dtype=np.int64), col.values
    a = np.asarray(col)
    if a.dtype.kind in "iub":
        return np.ascontiguousarray(a, dtype=np.int64), None
    values, inv = np.unique(a.astype(str), return_inverse=True)
    return np.ascontiguousarray(inv, dtype=np.int64), values


def _factorize_gpu(engine, cols_a, keep_a, cols_b=None, keep_b=None):
    """_factorize on the GPU (tad_factorize, include/tad.h): the rows' key tuples as int64 columns -> ids in order of first
    appearance over [kept rows of side a ++ kept rows of side b], TAD_KEY_SKIP elsewhere; the key table is read back from the rows
    where the keys first appear.  Same ids and tables as the pandas path (tests/test_gpu_factorize.py)."""
    enc_a = [_int_codes(c) for c in cols_a]
    enc_b = [_int_codes(c) for c in cols_b] if cols_b is not None else None
    n = enc_a[0][0].size
    key1, key2, first = engine.factorize([e[0] for e in enc_a], keep_a, [e[0] for e in enc_b] if enc_b else None, keep_b)
    side_b = first >= np.uint64(n)
    row = (first - np.where(side_b, np.uint64(n), np.uint64(0))).astype(np.int64)
    uniq = []
    for c in range(len(enc_a)):
        codes = enc_a[c][0][row]
        if enc_b is not None and side_b.any():
            codes = np.where(side_b, enc_b[c][0][row], codes)
        if enc_a[c][1] is None:
            uniq.append(codes)
        elif enc_b is None or not side_b.any():
            uniq.append(enc_a[c][1][codes] if enc_a[c][1].size else np.zeros(0, dtype=str))
        else:   # every side decodes with its own dictionary
            out = np.empty(codes.size, dtype=object)
            out[~side_b] = enc_a[c][1][codes[~side_b]]
            out[side_b] = enc_b[c][1][codes[side_b]]
            uniq.append(out.astype(str))
    return key1, key2, uniq, side_b


def prepare_columns(flows, start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="",
                    svc_port_name="", pod_name="", pod_namespace="", engine=None):
    """The host half of generate_tad_sql_query (ref:507-614): same predicates, same GROUP BY keys.  engine (a TadEngine): the
    key tuples are factorised on the GPU (tad_factorize) instead of with pandas — same ids, same key tables."""
    n = len(flows["flowEndSeconds"])
    flow_end = np.asarray(flows["flowEndSeconds"], dtype=np.int64)
    value = np.asarray(flows["throughput"], dtype=np.uint64)
    skip = np.uint64(capi.TAD_KEY_SKIP)
    keep = np.ones(n, dtype=bool)
    if ns_ignore_list:  # ref:549-553, 576-580
        ign = np.asarray(list(ns_ignore_list), dtype=str)
        keep &= ~_isin(_str_col(flows, "sourcePodNamespace"), ign) & ~_isin(_str_col(flows, "destinationPodNamespace"), ign)

    if agg_flow == "pod":
        by_name = bool(pod_name) and not pod_label
        ident = "PodName" if by_name else "PodLabels"
        sides = []
        for side, direction in (("destination", "inbound"), ("source", "outbound")):
            col = _str_col(flows, side + ident)
            ns = _str_col(flows, side + "PodNamespace")
            if pod_label:                                      # ref:516-527
                ok = _ilike_contains(col, pod_label)
                if pod_namespace:
                    ok &= _eq(ns, pod_namespace)
            elif pod_name:                                     # ref:528-543
                ok = _eq(col, pod_name)
                if pod_namespace:
                    ok &= _eq(ns, pod_namespace)
            else:                                              # ref:544-548
                ok = ~_eq(col, "")
            sides.append((ok & keep, ns, col, direction))
        mode = "podname" if by_name else "pod"
        if engine is not None and n:
            key_id, key_id2, uniq, side_b = _factorize_gpu(engine, [sides[0][1], sides[0][2]], sides[0][0], [sides[1][1], sides[1][2]], sides[1][0])
            uniq.append(np.where(side_b, sides[1][3], sides[0][3]))
            return PreparedColumns(mode, key_id, key_id2, flow_end, None, value, dict(zip(KEY_COLUMNS[mode], uniq)), 0, 0)
        sel = [np.flatnonzero(s[0]) for s in sides]
        codes, uniq = _factorize([_concat([_take(sides[i][1], sel[i]) for i in range(2)]),
                                  _concat([_take(sides[i][2], sel[i]) for i in range(2)]),
                                  np.concatenate([np.full(sel[i].size, sides[i][3]) for i in range(2)])])
        key_id = np.full(n, skip, dtype=np.uint64)
        key_id2 = np.full(n, skip, dtype=np.uint64)
        key_id[sel[0]] = codes[:sel[0].size].astype(np.uint64)
        key_id2[sel[1]] = codes[sel[0].size:].astype(np.uint64)
        table = dict(zip(KEY_COLUMNS[mode], uniq))
        # the pod SQL carries no flowStartSeconds / flowEndSeconds predicate (ref:556-565)
        return PreparedColumns(mode, key_id, key_id2, flow_end, None, value, table, 0, 0)

    if agg_flow == "external":
        keep &= np.asarray(flows["flowType"]).astype(np.int64) == 3      # ref:590
        if external_ip:
            keep &= _eq(_str_col(flows, "destinationIP"), external_ip)   # ref:591-593
        cols = [_str_col(flows, "destinationIP")]
    elif agg_flow == "svc":
        svc = _str_col(flows, "destinationServicePortName")
        keep &= _eq(svc, svc_port_name) if svc_port_name else ~_eq(svc, "")  # ref:594-601
        cols = [svc]
    elif not agg_flow:
        cols = [_str_col(flows, "sourceIP"), np.asarray(flows["sourceTransportPort"]).astype(np.int64),
                _str_col(flows, "destinationIP"), np.asarray(flows["destinationTransportPort"]).astype(np.int64),
                np.asarray(flows["protocolIdentifier"]).astype(np.int64), np.asarray(flows["flowStartSeconds"], dtype=np.int64)]
    else:
        raise ValueError("aggregated flow type should be 'pod' or 'external' or 'svc'")
    mode = agg_flow or ""
    if engine is not None and n:
        key_id, _, uniq, _ = _factorize_gpu(engine, cols, keep)
    else:
        sel = np.flatnonzero(keep)
        codes, uniq = _factorize([_take(c, sel) for c in cols])
        key_id = np.full(n, skip, dtype=np.uint64)
        key_id[sel] = codes.astype(np.uint64)
    table = dict(zip(KEY_COLUMNS[mode], uniq))
    flow_start = np.asarray(flows["flowStartSeconds"], dtype=np.int64) if start_time else None
    return PreparedColumns(mode, key_id, None, flow_end, flow_start, value, table, _epoch(start_time), _epoch(end_time))


class DeviceKeyColumn:
    """One column of the key table of a device-ingested job, decoded ON DEMAND: `col[kid]` gathers the column's values at the rows where the
    requested keys first appear (tad_widen_column as a gather) and looks the strings up in the host dictionary.  A table in mode None has a
    key per connection — tens of millions — of which only the keys with anomalous points are ever shown: reading the whole key table back
    was 0.84 s of a 1.0 s ingest at 5e7 keys (profiles/r6_i2_ingest_e2e_default_c8.log).  `sides`: [(column, constant)] per side of the
    key tuple (pod mode: inbound, outbound); a column is a DeviceDictColumn, a DeviceArray of integers, or None when `constant` is the value
    (pod mode's `direction`).  first: DeviceArray u64[num_keys], the virtual row (side * n + row) where every key first appears."""

    def __init__(self, engine, n_rows, first, sides):
        self.engine, self.n_rows, self.first, self.sides = engine, int(n_rows), first, sides

    def __len__(self):
        return self.first.n

    def __getitem__(self, kid):
        from .engine import DeviceArray
        kid = np.ascontiguousarray(np.asarray(kid).astype(np.uint64))
        if kid.size == 0:
            return np.zeros(0, dtype=str)
        uniq, inv = np.unique(kid, return_inverse=True)      # every key once (a key has many result rows)
        dk = DeviceArray.from_host(self.engine, uniq)
        vrow = self.engine.gather(self.first, dk)
        dk.free()
        side = (vrow >= np.uint64(self.n_rows)).astype(np.int64) if len(self.sides) > 1 else np.zeros(uniq.size, dtype=np.int64)
        rows = DeviceArray.from_host(self.engine, (vrow - side.astype(np.uint64) * np.uint64(self.n_rows)).astype(np.uint64))
        out = None
        for sidx, (col, const) in enumerate(self.sides):
            if col is None:
                vals = np.full(uniq.size, const)
            elif hasattr(col, "values"):
                codes = self.engine.gather(col.codes, rows).astype(np.int64)
                vals = col.values[codes] if col.values.size else np.zeros(codes.size, dtype=str)
            else:
                vals = self.engine.gather(col, rows).astype(np.int64)
            out = vals if out is None else np.where(side == sidx, vals, out)
        rows.free()
        return np.asarray(out)[inv]

    def __array__(self, dtype=None, copy=None):
        a = self[np.arange(len(self), dtype=np.uint64)]
        return a if dtype is None else a.astype(dtype)


def prepare_columns_device(flows, start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="",
                           svc_port_name="", pod_name="", pod_namespace="", engine=None):
    """prepare_columns for a table that is already in HBM (theia_amd.clickhouse.fetch_flows_device: 8-byte integer columns as DeviceArray,
    string columns as DeviceDictColumn = device codes + host dictionary).  The same predicates and GROUP BY keys (ref:507-614): every string
    predicate is evaluated on the column's DISTINCT values here and applied to the rows on the GPU (tad_mask_rows); the key tuples are
    factorised on the GPU (tad_factorize); the key table stays on the device and is decoded on demand — for the keys that have result rows —
    at the rows where the keys first appear (DeviceKeyColumn).  No per-row work on the host; nothing but the masks over distinct values and
    the result's keys crosses PCIe."""
    eng = engine or get_engine()
    flow_end, value = flows["flowEndSeconds"], flows["throughput"]
    n = flow_end.n
    common = []          # (codes, bool mask over the distinct values): terms every side's keep mask carries
    if ns_ignore_list:   # ref:549-553, 576-580
        ign = np.asarray(list(ns_ignore_list), dtype=str)
        for name in ("sourcePodNamespace", "destinationPodNamespace"):
            col = flows[name]
            common.append((col.codes, ~np.isin(col.values, ign)))

    if agg_flow == "pod":
        by_name = bool(pod_name) and not pod_label
        ident = "PodName" if by_name else "PodLabels"
        mode = "podname" if by_name else "pod"
        sides = []
        for side, direction in (("destination", "inbound"), ("source", "outbound")):
            col, ns = flows[side + ident], flows[side + "PodNamespace"]
            if pod_label:                                      # ref:516-527
                rx = _like_regex("%" + pod_label + "%")
                ok = np.fromiter((rx.match(u) is not None for u in col.values), dtype=bool, count=col.values.size)
            elif pod_name:                                     # ref:528-543
                ok = col.values == pod_name
            else:                                              # ref:544-548
                ok = col.values != ""
            terms = [(col.codes, ok)] + common
            if (pod_label or pod_name) and pod_namespace:
                terms.append((ns.codes, ns.values == pod_namespace))
            sides.append((eng.mask_rows(n, terms), ns, col, direction))
        if n == 0:
            return PreparedColumns(mode, flow_end, flow_end, flow_end, None, value, {k: np.zeros(0, dtype=str) for k in KEY_COLUMNS[mode]}, 0, 0)
        key_id, key_id2, first, hist = eng.factorize([sides[0][1].codes, sides[0][2].codes], sides[0][0], [sides[1][1].codes, sides[1][2].codes], sides[1][0],
                                                     with_hist=True)
        table = {KEY_COLUMNS[mode][0]: DeviceKeyColumn(eng, n, first, [(sides[0][1], None), (sides[1][1], None)]),
                 KEY_COLUMNS[mode][1]: DeviceKeyColumn(eng, n, first, [(sides[0][2], None), (sides[1][2], None)]),
                 KEY_COLUMNS[mode][2]: DeviceKeyColumn(eng, n, first, [(None, sides[0][3]), (None, sides[1][3])])}
        for k in (sides[0][0], sides[1][0]):
            k.free()
        # the pod SQL carries no flowStartSeconds / flowEndSeconds predicate (ref:556-565)
        return PreparedColumns(mode, key_id, key_id2, flow_end, None, value, table, 0, 0, key_hist=hist)

    terms = list(common)
    if agg_flow == "external":
        ft = np.zeros(65536, dtype=bool)
        ft[3] = True                                           # ref:590 `flowType = 3`: the integer column is its own code
        terms.append((flows["flowType"], ft))
        ip = flows["destinationIP"]
        if external_ip:
            terms.append((ip.codes, ip.values == external_ip))  # ref:591-593
        cols = [ip]
    elif agg_flow == "svc":
        svc = flows["destinationServicePortName"]
        terms.append((svc.codes, (svc.values == svc_port_name) if svc_port_name else (svc.values != "")))   # ref:594-601
        cols = [svc]
    elif not agg_flow:
        cols = [flows["sourceIP"], flows["sourceTransportPort"], flows["destinationIP"], flows["destinationTransportPort"],
                flows["protocolIdentifier"], flows["flowStartSeconds"]]
    else:
        raise ValueError("aggregated flow type should be 'pod' or 'external' or 'svc'")
    mode = agg_flow or ""
    flow_start = flows["flowStartSeconds"] if start_time else None
    if n == 0:
        return PreparedColumns(mode, flow_end, None, flow_end, flow_start, value, {k: np.zeros(0, dtype=str) for k in KEY_COLUMNS[mode]},
                               _epoch(start_time), _epoch(end_time))
    keep = eng.mask_rows(n, terms) if terms else None
    key_id, _, first, hist = eng.factorize([c.codes if hasattr(c, "values") else c for c in cols], keep, with_hist=True)
    table = {name: DeviceKeyColumn(eng, n, first, [(c, None)]) for name, c in zip(KEY_COLUMNS[mode], cols)}
    if keep is not None:
        keep.free()
    return PreparedColumns(mode, key_id, None, flow_end, flow_start, value, table, _epoch(start_time), _epoch(end_time), key_hist=hist)


# ------------------------------------------------------------------------------------------------
# the job (ref:647-710) and the result rows (ref:352-421, 500-503)
# ------------------------------------------------------------------------------------------------
def _sentinel_row(algo_type, agg_flow, tad_id):
    """ref:395-420: the single row written when no point is anomalous."""
    return {
        "sourceIP": "None", "sourceTransportPort": 0, "destinationIP": "None", "destinationTransportPort": 0,
        "protocolIdentifier": 0, "flowStartSeconds": datetime.now().strftime(TIME_FORMAT),
        "podNamespace": "None", "podLabels": "None", "podName": "None", "destinationServicePortName": "None",
        "direction": "None", "flowEndSeconds": 0, "throughputStandardDeviation": 0,
        "aggType": agg_flow if agg_flow else "None", "algoType": algo_type, "algoCalc": 0.0, "throughput": 0.0,
        "anomaly": "NO ANOMALY DETECTED", "id": str(tad_id),
    }


def result_rows(prep, res, algo_type, agg_flow, tad_id):
    """Engine result -> list of `tadetector` rows (only the mode's columns are set; ClickHouse defaults the rest)."""
    if res.n_rows == 0:
        return [_sentinel_row(algo_type, agg_flow, tad_id)]
    host = res.to_host()
    kid = host["key_id"].astype(np.int64)
    cols = {}
    for name in KEY_COLUMNS[prep.mode]:
        vals = prep.key_table[name][kid]
        if name == "podLabels":                       # ref:686-695: canonicalised AFTER the grouping
            canon = {u: remove_meaningless_labels(u) for u in np.unique(vals)}
            vals = np.asarray([canon[v] for v in vals], dtype=object)
        cols[name] = vals.tolist()
    agg_type = agg_flow if agg_flow else "None"       # ref:617-628
    rows = []
    for i in range(res.n_rows):
        row = {name: cols[name][i] for name in cols}
        row.update({
            "flowEndSeconds": int(host["flow_end_s"][i]),
            "throughputStandardDeviation": float(host["stddev"][i]),
            "aggType": agg_type, "algoType": algo_type,
            "algoCalc": float(host["algo_calc"][i]), "throughput": float(host["throughput"][i]),
            "anomaly": "true", "id": str(tad_id),     # ref:500-503 (cast boolean to string, add id)
        })
        rows.append(row)
    return rows


def result_columns(prep, res, algo_type, agg_flow, tad_id):
    """Engine result -> the `tadetector` rows as COLUMNS (dict name -> numpy array), vectorised: what the Arrow insert of
    theia_amd.clickhouse takes for large outputs.  Same content as result_rows (which builds one dict per row)."""
    if res.n_rows == 0:
        row = _sentinel_row(algo_type, agg_flow, tad_id)
        return {k: np.asarray([v]) for k, v in row.items()}
    host = res.to_host()
    kid = host["key_id"].astype(np.int64)
    n = res.n_rows
    cols = {}
    for name in KEY_COLUMNS[prep.mode]:
        col = prep.key_table[name]
        vals = col[kid] if isinstance(col, DeviceKeyColumn) else np.asarray(col)[kid]
        if name == "podLabels":
            uniq, inv = np.unique(vals.astype(str), return_inverse=True)
            vals = np.asarray([remove_meaningless_labels(u) for u in uniq], dtype=object)[inv]
        cols[name] = vals
    cols["flowEndSeconds"] = host["flow_end_s"]
    cols["throughputStandardDeviation"] = host["stddev"]
    cols["aggType"] = np.full(n, agg_flow if agg_flow else "None", dtype=object)
    cols["algoType"] = np.full(n, algo_type, dtype=object)
    cols["algoCalc"] = host["algo_calc"]
    cols["throughput"] = host["throughput"]
    cols["anomaly"] = np.full(n, "true", dtype=object)
    cols["id"] = np.full(n, str(tad_id), dtype=object)
    return cols


def anomaly_detection(algo_type, flows, start_time, end_time, tad_id_input, ns_ignore_list, agg_flow=None,
                      pod_label=None, external_ip=None, svc_port_name=None, pod_name=None, pod_namespace=None,
                      engine=None, pushdown=False, columnar=False, connections=0):
    """ref:647-710.  `flows` stands where the reference has the JDBC address: a column dict, a path for load_flows,
    or a theia_amd.clickhouse.ClickHouseHTTP client (then the rows come over ClickHouse's HTTP interface as Arrow
    batches; pushdown=True lets ClickHouse run the reference's GROUP BY and ships aggregated points instead).
    connections > 0 (with a ClickHouse client): the device ingest — that many parallel dictionary-encoded reads straight into HBM
    (theia_amd.clickhouse.fetch_flows_device + prepare_columns_device); falls back to the single-connection read when the table
    changes between the count query and the reads.
    Returns (stats dict, list of result rows) — or, with columnar=True, (stats dict, dict of result columns)."""
    if algo_type not in VALID_ALGOS:
        raise ValueError("Algorithm should be in {}".format(" or ".join(VALID_ALGOS)))
    agg_flow = agg_flow or ""
    if agg_flow not in VALID_AGG_FLOWS:
        raise ValueError("aggregated flow type should be 'pod' or 'external' or 'svc'")
    if isinstance(flows, str):
        flows = load_flows(flows)
    elif hasattr(flows, "query_columns"):   # ClickHouse over HTTP (ref:651-662 reads through JDBC)
        from . import clickhouse as ch
        args = (start_time or "", end_time or "", list(ns_ignore_list or ()), agg_flow, pod_label or "", external_ip or "",
                svc_port_name or "", pod_name or "", pod_namespace or "")
        if pushdown:
            flows = ch.fetch_points(flows, generate_tad_sql_query(*args), agg_flow, pod_name or "")
            start_time, end_time, ns_ignore_list = "", "", ()   # ClickHouse has applied them already
        elif connections:
            client, eng = flows, engine or get_engine()
            try:
                dev = ch.fetch_flows_device(client, eng, *args, connections=connections)
            except RuntimeError as exc:
                logger.warning("device ingest abandoned (%s): single-connection read", exc)
                flows = ch.fetch_flows(client, *args, engine=eng)
            else:
                prep = prepare_columns_device(dev, *args, engine=eng)
                res = eng.run(algo_type, prep.key_id, prep.flow_end_s, prep.value, max(prep.num_keys, 1), agg_flow=agg_flow,
                              key_id2=prep.key_id2, flow_start_s=prep.flow_start_s, start_time=prep.start_time,
                              end_time=prep.end_time, job_id=str(tad_id_input or ""), key_hist=prep.key_hist)
                if columnar:
                    return res.stats, result_columns(prep, res, algo_type, agg_flow, tad_id_input)
                return res.stats, result_rows(prep, res, algo_type, agg_flow, tad_id_input)
        else:
            flows = ch.fetch_flows(flows, *args, engine=engine or get_engine())   # string columns dictionary-encoded on the GPU
    eng = engine or get_engine()
    prep = prepare_columns(flows, start_time or "", end_time or "", ns_ignore_list or (), agg_flow, pod_label or "",
                           external_ip or "", svc_port_name or "", pod_name or "", pod_namespace or "", engine=eng)   # key tuples factorised on the GPU
    res = eng.run(algo_type, prep.key_id, prep.flow_end_s, prep.value, max(prep.num_keys, 1), agg_flow=agg_flow,
                  key_id2=prep.key_id2, flow_start_s=prep.flow_start_s, start_time=prep.start_time,
                  end_time=prep.end_time, job_id=str(tad_id_input or ""))
    if columnar:
        return res.stats, result_columns(prep, res, algo_type, agg_flow, tad_id_input)
    return res.stats, result_rows(prep, res, algo_type, agg_flow, tad_id_input)


RESULT_COLUMNS = ("sourceIP", "sourceTransportPort", "destinationIP", "destinationTransportPort", "protocolIdentifier",
                  "flowStartSeconds", "podNamespace", "podLabels", "podName", "destinationServicePortName", "direction",
                  "flowEndSeconds", "throughputStandardDeviation", "aggType", "algoType", "algoCalc", "throughput",
                  "anomaly", "id")   # create_table.sh:363-384


def write_anomaly_detection_result(result_rows_, destination, result_table_name=RESULT_TABLE_NAME, tad_id_input=None):
    """ref:713-726.  Appends the rows as JSON lines (ClickHouse `FORMAT JSONEachRow` input) to `destination`
    — a path or a file object; the ClickHouse transport itself is SURVEY.md §8f rank 1.  Returns the job id."""
    tad_id = tad_id_input if tad_id_input else str(uuid.uuid4())
    own = isinstance(destination, str)
    fh = open(destination, "a") if own else destination
    try:
        for row in result_rows_:
            out = {k: row[k] for k in RESULT_COLUMNS if k in row}
            out["id"] = row.get("id") if row.get("id") not in (None, "None", "") else tad_id
            fh.write(json.dumps(out) + "\n")
    finally:
        if own:
            fh.close()
    return tad_id


def store_result_columns(client, cols, table=RESULT_TABLE_NAME):
    """ref:713-726: append the result columns to default.tadetector — columnar (Arrow) insert, no per-row work on the host.
    Returns the number of rows written (1 for the 'NO ANOMALY DETECTED' sentinel row, ref:395-420)."""
    if len(cols["anomaly"]) == 1 and cols["anomaly"][0] != "true":   # the sentinel row: its DateTime fields are text
        client.insert_rows([_db_row({k: (v[0].item() if hasattr(v[0], "item") else v[0]) for k, v in cols.items()})], table)
        return 1
    client.insert_columns(cols, table)
    return len(cols["anomaly"])


def _db_row(row):
    """A result row as ClickHouse's JSONEachRow wants it: DateTime columns as 'YYYY-MM-DD hh:mm:ss' (UTC)."""
    out = {k: row[k] for k in RESULT_COLUMNS if k in row}
    for col in ("flowEndSeconds", "flowStartSeconds"):
        v = out.get(col)
        if isinstance(v, (int, np.integer)):
            out[col] = datetime.fromtimestamp(int(v), tz=timezone.utc).strftime(TIME_FORMAT)
    return out


def load_flows(path):
    """.npz (numpy), .parquet / .csv (pyarrow) -> column dict; DateTime columns as epoch seconds."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=True) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".parquet"):
        import pyarrow.parquet as pq
        tbl = pq.read_table(path)
    elif path.endswith(".csv"):
        import pyarrow.csv as pcsv
        tbl = pcsv.read_csv(path)
    else:
        raise ValueError("flows file must be .npz, .parquet or .csv")
    out = {}
    for name in tbl.column_names:
        col = tbl.column(name)
        if str(col.type).startswith("timestamp"):
            out[name] = col.cast("timestamp[s]").cast("int64").to_numpy()
        else:
            out[name] = col.to_numpy(zero_copy_only=False)
    return out


HELP_MESSAGE = """
    Start the Throughput Anomaly Detection job on the MI355X engine.
        Options:
        -h, --help: Show help message.
        -a, --algo=EWMA: EWMA, ARIMA or DBSCAN.
        -d, --db_jdbc_url=URL: ClickHouse (jdbc:clickhouse://host:8123 or http://host:8123); flow records are read
            and results appended over its HTTP interface with CH_USERNAME / CH_PASSWORD.  Default
            jdbc:clickhouse://clickhouse-clickhouse.flow-visibility.svc:8123 when --flows is not given.
        -G, --pushdown-groupby: let ClickHouse run the GROUP BY (the reference's SQL) and ship aggregated points.
        -C, --connections (optional): read the raw rows over this many parallel connections straight into GPU memory,
            string columns as Arrow dictionaries (LowCardinality); default 0 = one connection, host decode.
        -F, --flows=PATH: read the flow table from a file instead (.npz / .parquet / .csv, columns of default.flows).
        -o, --out=PATH: append the tadetector rows as JSON lines to PATH instead of inserting them ('-' = stdout).
        -s, --start_time=None / -e, --end_time=None: 'YYYY-MM-DD hh:mm:ss' UTC.
        -i, --id=None: job id (uuid); generated when missing.
        -n, --ns_ignore_list=[]: JSON list of namespaces to ignore.
        -f, --agg-flow=None: pod | external | svc.
        -l, --pod-label, -N, --pod-name, -P, --pod-namespace, -x, --external-ip, -p, --svc-port-name.
    """


def main(argv=None):
    """ref:729-900: same options, same exit codes (2 on a bad argument)."""
    argv = sys.argv[1:] if argv is None else argv
    try:
        opts, _ = getopt.getopt(argv, "ha:d:s:e:i:n:f:l:x:p:N:P:F:o:GC:",
                                ["help", "algo=", "db_jdbc_url=", "start_time=", "end_time=", "id=", "ns_ignore_list=",
                                 "ns-ignore-list=", "agg-flow=", "pod-label=", "external-ip=", "svc-port-name=", "pod-name=",
                                 "pod-namespace=", "flows=", "out=", "pushdown-groupby", "connections="])
    except getopt.GetoptError as exc:
        logger.error("ERROR of getopt.getopt: %s", exc)
        logger.info(HELP_MESSAGE)
        sys.exit(2)
    a = {"algo": "", "start": "", "end": "", "id": None, "ns": [], "agg": "", "label": "", "ip": "", "svc": "",
         "name": "", "namespace": "", "flows": "", "out": "", "db": "", "pushdown": False, "connections": 0}

    def bad(msg):
        logger.error(msg)
        logger.info(HELP_MESSAGE)
        sys.exit(2)

    for opt, arg in opts:
        if opt in ("-h", "--help"):
            logger.info(HELP_MESSAGE)
            sys.exit()
        elif opt in ("-a", "--algo"):
            if arg not in VALID_ALGOS:
                bad("Algorithm should be in {}".format(" or ".join(VALID_ALGOS)))
            a["algo"] = arg
        elif opt in ("-d", "--db_jdbc_url"):
            if not (arg.startswith("jdbc:") or arg.startswith("http://") or arg.startswith("https://")):
                bad("Please provide a valid JDBC url for ClickHouse database")
            a["db"] = arg
        elif opt in ("-G", "--pushdown-groupby"):
            a["pushdown"] = True
        elif opt in ("-C", "--connections"):
            if not arg.isdigit() or not 0 <= int(arg) <= 64:
                bad("connections should be an integer between 0 and 64.")
            a["connections"] = int(arg)
        elif opt in ("-s", "--start_time", "-e", "--end_time"):
            which = "start" if opt in ("-s", "--start_time") else "end"
            try:
                datetime.strptime(arg, TIME_FORMAT)
            except ValueError:
                bad("{}_time should be in 'YYYY-MM-DD hh:mm:ss' format.".format(which))
            a[which] = arg
        elif opt in ("-n", "--ns_ignore_list", "--ns-ignore-list"):   # the controller passes --ns-ignore-list (controller.go:548)
            lst = json.loads(arg)
            if not isinstance(lst, list):
                bad("ns_ignore_list should be a list.")
            a["ns"] = lst
        elif opt in ("-i", "--id"):
            a["id"] = arg
        elif opt in ("-f", "--agg-flow"):
            a["agg"] = arg
        elif opt in ("-l", "--pod-label"):
            a["label"] = arg
        elif opt in ("-N", "--pod-name"):
            a["name"] = arg
        elif opt in ("-P", "--pod-namespace"):
            a["namespace"] = arg
        elif opt in ("-x", "--external-ip"):
            a["ip"] = arg
        elif opt in ("-p", "--svc-port-name"):
            a["svc"] = arg
        elif opt in ("-F", "--flows"):
            a["flows"] = arg
        elif opt in ("-o", "--out"):
            a["out"] = arg
    if not a["algo"]:
        bad("Algorithm should be in {}".format(" or ".join(VALID_ALGOS)))
    tad_id = a["id"] or str(uuid.uuid4())
    client = None
    if not a["flows"] or (a["db"] and not a["out"]):
        from . import clickhouse as ch
        client = ch.ClickHouseHTTP(a["db"] or ch.DEFAULT_JDBC_URL)
    t0 = time.time()
    logger.info("Script started at %s", datetime.now().strftime("%a, %d %B %Y %H:%M:%S"))
    to_db = client is not None and not a["out"]
    try:
        _, rows = anomaly_detection(a["algo"], a["flows"] or client, a["start"], a["end"], tad_id, a["ns"], a["agg"],
                                    a["label"], a["ip"], a["svc"], a["name"], a["namespace"], pushdown=a["pushdown"], columnar=to_db,
                                    connections=a["connections"] if client is not None and not a["flows"] else 0)
    except (TadError, ValueError, OSError) as exc:
        logger.error("Anomaly Detection failed: %s", exc)
        sys.exit(1)
    t1 = time.time()
    if a["out"] or client is None:
        write_anomaly_detection_result(rows, sys.stdout if a["out"] in ("", "-") else a["out"], RESULT_TABLE_NAME, tad_id)
    else:
        store_result_columns(client, rows)
    logger.info("Anomaly Detection completed, id: %s, in %s seconds ", tad_id, t1 - t0)
    return tad_id


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    main()
