"""The ClickHouse HTTP transport (theia_amd/clickhouse.py) against an in-process server that speaks the two formats it
uses: `SELECT ... FORMAT ArrowStream` out, `INSERT ... FORMAT JSONEachRow` in.  CPU tests cover the transport, the
SQL it sends and the raw-rows / pushdown equivalence of the host half; the GPU test runs the CLI end to end."""
import base64
import io
import json
import threading
import urllib.parse
from http.server import BaseHTTPRequestHandler, HTTPServer

import numpy as np
import pandas as pd
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

from oracle import job_oracle as jo
from oracle import tad_oracle as orc
from theia_amd import anomaly_detection as ad
from theia_amd import clickhouse as ch

from test_host_job import FakeResult, canon, oracle_middle


class FakeClickHouse:
    """Answers a SELECT by looking the SQL text up in `responses` (exact match), records INSERTed rows."""

    def __init__(self):
        self.responses = {}
        self.queries, self.inserted, self.auth, self.commands = [], [], [], []
        owner = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def do_POST(self):
                body = self.rfile.read(int(self.headers.get("Content-Length", 0)))
                params = urllib.parse.parse_qs(urllib.parse.urlparse(self.path).query)
                owner.auth.append(self.headers.get("Authorization"))
                if "query" in params and params["query"][0].startswith("INSERT INTO"):
                    if params["query"][0].endswith("FORMAT ArrowStream"):
                        rows = ipc.open_stream(io.BytesIO(body)).read_all().to_pylist()
                    else:
                        rows = [json.loads(l) for l in body.decode().splitlines() if l]
                    owner.inserted.append((params["query"][0], rows))
                    self.send_response(200); self.end_headers(); return
                sql = body.decode()
                if sql.startswith("ALTER TABLE"):                      # a statement without a result set
                    owner.commands.append(sql)
                    self.send_response(200); self.end_headers(); return
                owner.queries.append(sql)
                assert sql.endswith(" FORMAT ArrowStream")
                key = sql[: -len(" FORMAT ArrowStream")]
                if " FROM tadetector" in key:                          # the result table: answered from what has been inserted
                    table = owner.select_tadetector(key, {k[6:]: v[0] for k, v in params.items() if k.startswith("param_")})
                    sink = io.BytesIO()
                    with ipc.new_stream(sink, table.schema) as w:
                        w.write_table(table)
                    self.send_response(200); self.end_headers(); self.wfile.write(sink.getvalue()); return
                known = {k.rstrip(): v for k, v in owner.responses.items()}   # the client strips the SQL's trailing blank
                if key not in known:
                    self.send_response(404); self.end_headers(); self.wfile.write(b"unknown query"); return
                table = known[key]
                sink = io.BytesIO()
                with ipc.new_stream(sink, table.schema) as w:
                    w.write_table(table)
                self.send_response(200); self.end_headers(); self.wfile.write(sink.getvalue())

        self.httpd = HTTPServer(("127.0.0.1", 0), Handler)
        self.url = "http://127.0.0.1:%d" % self.httpd.server_address[1]
        self.thread = threading.Thread(target=self.httpd.serve_forever, daemon=True)
        self.thread.start()

    def select_tadetector(self, sql, params):
        """`SELECT <columns> FROM tadetector WHERE id = ({id:String})` / `SELECT DISTINCT id FROM tadetector` over the inserted rows, with
        the column types of create_table.sh:363-384 (a column an INSERT did not name holds the type's default; rows a
        `ALTER TABLE ... DELETE WHERE id = ('<id>')` named are gone)."""
        deleted = {c.split("('")[1].split("')")[0] for c in self.commands if "DELETE WHERE id = ('" in c}
        rows = [r for _, rs in self.inserted for r in rs if r.get("id") not in deleted]
        if sql.startswith("SELECT DISTINCT id FROM tadetector"):
            return pa.table({"id": pa.array(sorted({r["id"] for r in rows}), pa.string())})
        cols = sql[len("SELECT "):sql.index(" FROM ")].split(", ")
        assert sql.endswith("WHERE id = ({id:String})") and "id" in params, (sql, params)
        rows = [r for r in rows if r.get("id") == params["id"]]
        arrays = {}
        for c in cols:
            kind = ch.TADETECTOR_COLUMNS[c]
            if kind == "datetime":
                arrays[c] = pa.array(np.asarray([r.get(c) or 0 for r in rows], dtype="int64").astype("datetime64[s]"), pa.timestamp("s"))
            elif kind == "f64":
                arrays[c] = pa.array([float(r.get(c) or 0.0) for r in rows], pa.float64())
            elif kind in ("u16", "u8"):
                arrays[c] = pa.array([int(r.get(c) or 0) for r in rows], pa.uint16())
            else:
                arrays[c] = pa.array([str(r.get(c) or "") for r in rows], pa.string())
        return pa.table(arrays)

    def close(self):
        self.httpd.shutdown()


def arrow_table(cols):
    """column dict -> Arrow table typed the way ClickHouse types default.flows (create_table.sh:31-85)."""
    arrays = {}
    for name, v in cols.items():
        v = np.asarray(v)
        if name.endswith("Seconds"):
            arrays[name] = pa.array(v.astype("datetime64[s]"), pa.timestamp("s"))      # DateTime
        elif name in ("throughput", "max(throughput)", "sum(throughput)"):
            arrays[name] = pa.array(v.astype(np.uint64), pa.uint64())
        elif v.dtype.kind in "iu":
            arrays[name] = pa.array(v.astype(np.uint16), pa.uint16())
        else:
            arrays[name] = pa.array(v.astype(str).tolist(), pa.string())
    return pa.table(arrays)


@pytest.fixture()
def server():
    s = FakeClickHouse()
    yield s
    s.close()


KW = dict(start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="", svc_port_name="",
          pod_name="", pod_namespace="")
CASES = [dict(agg_flow="svc"), dict(agg_flow="external", external_ip="52.1.1.2"), dict(agg_flow="", end_time="2022-08-11 08:00:00"),
         dict(agg_flow="pod", pod_label="app1"), dict(agg_flow="pod", pod_name="pod-2", pod_namespace="flow-visibility"),
         dict(agg_flow="pod", ns_ignore_list=["kube-system"])]


def test_jdbc_url_and_basic_auth(server, monkeypatch):
    assert ch.jdbc_to_http("jdbc:clickhouse://clickhouse-clickhouse.flow-visibility.svc:8123") == \
        "http://clickhouse-clickhouse.flow-visibility.svc:8123"
    with pytest.raises(ValueError):
        ch.jdbc_to_http("mysql://x")
    monkeypatch.setenv("CH_USERNAME", "u1")
    monkeypatch.setenv("CH_PASSWORD", "p1")          # controller.go:649-658: credentials arrive in the environment
    server.responses["SELECT 1"] = pa.table({"1": pa.array([1], pa.uint8())})
    out = ch.ClickHouseHTTP(server.url).query_columns("SELECT 1")
    assert out["1"].tolist() == [1]
    assert server.auth[-1] == "Basic " + base64.b64encode(b"u1:p1").decode()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_raw_rows_and_pushdown_reads_give_the_same_job(server, case):
    flows = jo.synth_flows(5000)
    kw = dict(KW); kw.update(case)
    args = (kw["start_time"], kw["end_time"], list(kw["ns_ignore_list"]), kw["agg_flow"], kw["pod_label"], kw["external_ip"],
            kw["svc_port_name"], kw["pod_name"], kw["pod_namespace"])
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    # raw rows: the server ignores the WHERE clause (it returns every row of the requested columns); the host applies
    # the same predicates again, so the result cannot depend on how much ClickHouse filtered
    sql_rows = ch.rows_query(*args)
    want_cols = sql_rows[len("SELECT "):sql_rows.index(" FROM ")].split(", ")
    server.responses[sql_rows] = arrow_table({c: flows[c] for c in want_cols})
    got_flows = ch.fetch_flows(client, *args)
    prep = ad.prepare_columns(got_flows, **kw)
    rows_raw = ad.result_rows(prep, FakeResult(oracle_middle(prep, "EWMA", kw["agg_flow"])), "EWMA", kw["agg_flow"], "j")
    # pushdown: the reference SQL, answered with what that SQL computes (pandas restatement in the oracle)
    sql_push = ad.generate_tad_sql_query(*args)
    pts, keys = jo.stage0_sql(pd.DataFrame({k: np.asarray(v) for k, v in flows.items()}), *args[:2], args[2], *args[3:])
    agg_name = "max(throughput)" if kw["agg_flow"] == "" else "sum(throughput)"
    cols = {k: pts[k].to_numpy() for k in keys}
    cols["flowEndSeconds"] = pts["flowEndSeconds"].to_numpy()
    cols[agg_name] = pts["v"].to_numpy()
    if kw["agg_flow"] == "external":
        cols["flowType"] = np.full(len(pts), 3)
    server.responses[sql_push] = arrow_table(cols)
    pushed = ch.fetch_points(client, sql_push, kw["agg_flow"], kw["pod_name"])
    kw2 = dict(kw, start_time="", end_time="", ns_ignore_list=())
    prep2 = ad.prepare_columns(pushed, **kw2)
    rows_push = ad.result_rows(prep2, FakeResult(oracle_middle(prep2, "EWMA", kw["agg_flow"])), "EWMA", kw["agg_flow"], "j")
    want = jo.run(flows, "EWMA", tad_id="j", **kw)
    for r in want:
        sd = r["throughputStandardDeviation"]
        r["throughputStandardDeviation"] = 0.0 if sd is None else float(sd)
    assert canon(rows_raw) == canon(want)
    assert canon(rows_push) == canon(want)
    assert "WHERE" in sql_rows or not any(args[:3]) and kw["agg_flow"] == ""


def test_rows_query_where_clause():
    q = ch.rows_query("2022-01-01 00:00:00", "2022-01-02 00:00:00", ["a", "b"], "svc", "", "", "http", "", "")
    assert q == ("SELECT destinationServicePortName, flowStartSeconds, flowEndSeconds, throughput, sourcePodNamespace, "
                 "destinationPodNamespace FROM default.flows WHERE sourcePodNamespace NOT IN ('a', 'b') AND "
                 "destinationPodNamespace NOT IN ('a', 'b') AND flowStartSeconds >= '2022-01-01 00:00:00' AND "
                 "flowEndSeconds < '2022-01-02 00:00:00' AND destinationServicePortName = 'http'")
    q = ch.rows_query("2022-01-01 00:00:00", "", [], "pod", "", "", "", "p1", "ns1")
    assert "flowStartSeconds >=" not in q        # the pod SQL carries no time window (anomaly_detection.py:556-565)
    assert "(destinationPodName = 'p1' AND destinationPodNamespace = 'ns1') OR (sourcePodName = 'p1' AND sourcePodNamespace = 'ns1')" in q


def test_rows_query_dictionary_shard_and_count_variants():
    """The statements of the device ingest (fetch_flows_device): only the columns the mode's job reads, string columns as
    toLowCardinality(col) AS col (so that ClickHouse sends Arrow dictionaries), a shard predicate per parallel read, and the count query that
    gives every read its place in the device columns.  The WHERE clause is the reference's (anomaly_detection.py:507-614) in every variant."""
    plain = ch.rows_query("", "", ["kube-system"], "pod", "", "", "", "", "")
    where = plain[plain.index(" WHERE "):]
    q = ch.rows_query("", "", ["kube-system"], "pod", "", "", "", "", "", dictionary=True, shard=(2, 8))
    assert q == ("SELECT toLowCardinality(sourcePodNamespace) AS sourcePodNamespace, toLowCardinality(destinationPodNamespace) AS destinationPodNamespace, "
                 "toLowCardinality(sourcePodLabels) AS sourcePodLabels, toLowCardinality(destinationPodLabels) AS destinationPodLabels, flowEndSeconds, throughput "
                 "FROM default.flows" + where + " AND cityHash64(sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, flowStartSeconds, "
                 "flowEndSeconds) % 8 = 2")
    by_name = ch.rows_query("", "", [], "pod", "", "", "", "p1", "ns1", dictionary=True, shard=(0, 2))
    assert "sourcePodName" in by_name and "PodLabels" not in by_name[:by_name.index(" FROM ")]
    c = ch.rows_query("", "", ["kube-system"], "pod", "", "", "", "", "", dictionary=True, shard=(None, 8), count_only=True)
    assert c == ("SELECT cityHash64(sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, flowStartSeconds, flowEndSeconds) % 8 AS shard, "
                 "count() AS rows FROM default.flows" + where + " GROUP BY shard")
    # svc without a time window or ns-ignore list: three columns; with them: flowStartSeconds and the namespaces come along
    q = ch.rows_query("", "", [], "svc", "", "", "", "", "", dictionary=True, shard=(0, 1))
    assert q[len("SELECT "):q.index(" FROM ")] == "toLowCardinality(destinationServicePortName) AS destinationServicePortName, flowEndSeconds, throughput"
    q = ch.rows_query("2022-01-01 00:00:00", "", ["a"], "svc", "", "", "", "", "", dictionary=True, shard=(0, 1))
    cols = q[len("SELECT "):q.index(" FROM ")]
    assert "flowStartSeconds" in cols and "toLowCardinality(sourcePodNamespace) AS sourcePodNamespace" in cols
    # mode None: flowStartSeconds is part of the key; ports and protocol stay integers
    q = ch.rows_query("", "", [], "", "", "", "", "", "", dictionary=True, shard=(1, 4))
    cols = q[len("SELECT "):q.index(" FROM ")].split(", ")
    assert cols == ["toLowCardinality(sourceIP) AS sourceIP", "sourceTransportPort", "toLowCardinality(destinationIP) AS destinationIP",
                    "destinationTransportPort", "protocolIdentifier", "flowStartSeconds", "flowEndSeconds", "throughput"]
    assert q.endswith(" WHERE cityHash64(sourceIP, sourceTransportPort, destinationIP, destinationTransportPort, flowStartSeconds, flowEndSeconds) % 4 = 1")
    assert ch.rows_query("", "", [], "svc", "", "", "", "", "") == ch.rows_query("", "", [], "svc", "", "", "", "", "", dictionary=False, shard=None)


def test_vocabulary_maps_batch_dictionaries_into_one_in_order_of_appearance():
    v = ch._Vocabulary()
    assert v.remap(pa.array(["b", "a", None])).tolist() == [0, 1, 2] and v.values == ["b", "a", ""]
    assert v.remap(pa.array(["a", "c", "b", ""])).tolist() == [1, 3, 0, 2] and v.values == ["b", "a", "", "c"]
    assert v.remap(pa.array([], pa.string())).tolist() == []
    assert v.remap(pa.array(["d", "d", "a"])).tolist() == [4, 4, 1] and v.values == ["b", "a", "", "c", "d"]      # a dictionary with a repeated value
    import threading
    w = ch._Vocabulary()
    outs = {}

    def work(i):
        outs[i] = [w.remap(pa.array(["v%d" % ((i * 7 + j) % 50) for j in range(30)])) for _ in range(20)]
    ths = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    vals = w.values
    assert sorted(vals) == sorted(set(vals)) and len(vals) == 50           # every value once, whatever the interleaving
    for i in range(6):
        want = ["v%d" % ((i * 7 + j) % 50) for j in range(30)]
        assert all([vals[c] for c in r.tolist()] == want for r in outs[i])


def test_cli_connections_option_is_validated():
    for bad in ("x", "-1", "65"):
        with pytest.raises(SystemExit) as ei:
            ad.main(["--algo", "EWMA", "--connections", bad])
        assert ei.value.code == 2


def test_insert_columns_arrow(server):
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    flows = jo.synth_flows(3000)
    prep = ad.prepare_columns(flows, agg_flow="", **{k: v for k, v in KW.items() if k != "agg_flow"})
    mid = oracle_middle(prep, "EWMA", "")
    cols = ad.result_columns(prep, FakeResult(mid), "EWMA", "", "col-1")
    rows = ad.result_rows(prep, FakeResult(mid), "EWMA", "", "col-1")
    assert client.insert_columns(cols) == len(rows) > 0
    q, got = server.inserted[-1]
    assert q.startswith("INSERT INTO default.tadetector (sourceIP, sourceTransportPort, destinationIP") and q.endswith("FORMAT ArrowStream")
    assert canon([{k: (int(v) if k in ("flowEndSeconds", "flowStartSeconds", "sourceTransportPort", "destinationTransportPort",
                                          "protocolIdentifier") else v) for k, v in r.items()} for r in got]) == canon(rows)


def test_insert_rows_json_each_row(server):
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    rows = [ad._db_row({"destinationServicePortName": "s", "flowEndSeconds": 1660202814, "throughputStandardDeviation": 1.5,
                        "aggType": "svc", "algoType": "EWMA", "algoCalc": 2.0, "throughput": 9.0, "anomaly": "true", "id": "x"})]
    assert client.insert_rows(rows) == 1
    q, got = server.inserted[-1]
    assert q == "INSERT INTO default.tadetector FORMAT JSONEachRow"
    assert got[0]["flowEndSeconds"] == "2022-08-11 07:26:54" and got[0]["id"] == "x"


@pytest.mark.gpu
@pytest.mark.parametrize("pushdown", [False, True])
def test_cli_against_clickhouse_http(engine, server, pushdown):
    ad.set_engine(engine)
    try:
        flows = jo.synth_flows(4000)
        args = ("", "", [], "svc", "", "", "", "", "")
        if pushdown:
            sql = ad.generate_tad_sql_query(*args)
            pts, keys = jo.stage0_sql(pd.DataFrame({k: np.asarray(v) for k, v in flows.items()}), "", "", [], "svc")
            server.responses[sql] = arrow_table({"destinationServicePortName": pts["destinationServicePortName"].to_numpy(),
                                                 "flowEndSeconds": pts["flowEndSeconds"].to_numpy(), "sum(throughput)": pts["v"].to_numpy()})
        else:
            sql = ch.rows_query(*args)
            cols = sql[len("SELECT "):sql.index(" FROM ")].split(", ")
            server.responses[sql] = arrow_table({c: flows[c] for c in cols})
        argv = ["--algo", "EWMA", "--agg-flow", "svc", "--id", "c-1", "--db_jdbc_url", server.url] + (["--pushdown-groupby"] if pushdown else [])
        assert ad.main(argv) == "c-1"
        want = jo.run(flows, "EWMA", tad_id="c-1", agg_flow="svc")
        q, got = server.inserted[-1]
        assert q == ("INSERT INTO default.tadetector (destinationServicePortName, flowEndSeconds, throughputStandardDeviation, "
                     "aggType, algoType, algoCalc, throughput, anomaly, id) FORMAT ArrowStream") and len(got) == len(want)
        key = lambda r: (r["destinationServicePortName"], int(r["flowEndSeconds"]))
        w = {key(r): r for r in want}
        for r in got:
            assert w[key(r)]["algoCalc"] == r["algoCalc"] and w[key(r)]["throughput"] == r["throughput"]
            assert r["anomaly"] == "true" and r["id"] == "c-1" and r["aggType"] == "svc"
    finally:
        ad.set_engine(None)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_dictionary_encoded_columns_prepare_like_plain_strings(case):
    """DictColumn (codes + distinct values, what query_columns(dict_strings=True) returns) through prepare_columns gives the
    same key ids, key table and kept rows as the plain string arrays: the predicates run on the distinct values only."""
    flows = jo.synth_flows(4000)
    kw = dict(KW); kw.update(case)
    enc = {}
    for name, col in flows.items():
        a = np.asarray(col)
        if a.dtype.kind in "US":
            values, codes = np.unique(a.astype(str), return_inverse=True)
            perm = np.random.default_rng(3).permutation(values.size)          # an arbitrary dictionary order
            inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
            enc[name] = ad.DictColumn(inv[codes], values[perm])
        else:
            enc[name] = a
    a, b = ad.prepare_columns(flows, **kw), ad.prepare_columns(enc, **kw)
    assert a.mode == b.mode and a.num_keys == b.num_keys and (a.start_time, a.end_time) == (b.start_time, b.end_time)
    assert (a.key_id == b.key_id).all() and (a.flow_end_s == b.flow_end_s).all() and (a.value == b.value).all()
    assert (a.key_id2 is None) == (b.key_id2 is None) and (a.key_id2 is None or (a.key_id2 == b.key_id2).all())
    for name in a.key_table:
        assert (np.asarray(a.key_table[name]).astype(str) == np.asarray(b.key_table[name]).astype(str)).all(), name


def test_query_columns_unifies_the_dictionaries_of_the_record_batches(server):
    t1 = pa.table({"s": pa.array(["b", "a", "b", ""]), "n": pa.array([1, 2, 3, 4], pa.uint16())})
    t2 = pa.table({"s": pa.array(["c", "a", "c"]), "n": pa.array([5, 6, 7], pa.uint16())})

    class TwoBatches:
        schema = t1.schema
    sink = io.BytesIO()
    with ipc.new_stream(sink, t1.schema) as w:
        w.write_table(t1); w.write_table(t2)
    server.responses["SELECT s, n FROM x"] = pa.concat_tables([t1, t2])   # (the fake re-serialises; batches are per input table)
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    plain = client.query_columns("SELECT s, n FROM x")
    enc = client.query_columns("SELECT s, n FROM x", dict_strings=True)
    assert isinstance(enc["s"], ad.DictColumn) and sorted(enc["s"].values.tolist()) == ["", "a", "b", "c"]
    assert enc["s"].materialise().tolist() == plain["s"].tolist() == ["b", "a", "b", "", "c", "a", "c"]
    assert enc["n"].tolist() == [1, 2, 3, 4, 5, 6, 7]


def test_query_columns_accepts_server_side_dictionaries_bools_dates_and_decimals(server):
    """LowCardinality columns arrive as Arrow dictionaries when output_format_arrow_low_cardinality_as_dictionary is on: the
    indices are remapped like a batch-local dictionary (no re-encode); non-string columns pass through and get no vocabulary."""
    import datetime
    import decimal
    s1 = pa.DictionaryArray.from_arrays(pa.array([1, 0, 1, 2], pa.int8()), pa.array(["a", "b", ""]))
    s2 = pa.DictionaryArray.from_arrays(pa.array([0, 0, None], pa.int8()), pa.array(["c", "a"]))
    def tab(s, n0):
        k = len(s)
        return pa.table({"s": s, "flag": pa.array([True, False, True, True][:k]), "d": pa.array([datetime.date(2022, 8, 11)] * k, pa.date32()),
                         "x": pa.array([decimal.Decimal("1.5")] * k, pa.decimal128(10, 2)), "n": pa.array(list(range(n0, n0 + k)), pa.uint16())})
    server.responses["SELECT q"] = pa.concat_tables([tab(s1, 0), tab(s2, 4)])
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    plain = client.query_columns("SELECT q")
    enc = client.query_columns("SELECT q", dict_strings=True)
    assert plain["s"].tolist() == ["b", "a", "b", "", "c", "c", ""]
    assert isinstance(enc["s"], ad.DictColumn) and enc["s"].materialise().tolist() == plain["s"].tolist()
    for name in ("flag", "d", "x", "n"):
        assert not isinstance(enc[name], ad.DictColumn), name
        assert np.array_equal(np.asarray(enc[name]), np.asarray(plain[name])), name
    assert enc["flag"].tolist() == [True, False, True, True, True, False, True] and enc["x"].tolist() == [1.5] * 7
    assert enc["d"].tolist() == [1660176000] * 7 and enc["n"].tolist() == list(range(7))


@pytest.mark.gpu
def test_query_columns_encodes_string_columns_on_the_gpu_like_arrow_does(engine, server, monkeypatch):
    """query_columns(dict_strings=True, engine=...): plain string columns go through tad_encode_strings chunk by chunk; codes, dictionaries
    and their ORDER are those of the host path (Arrow's dictionary_encode per record batch, unified in order of appearance)."""
    rng = np.random.default_rng(12)
    tabs = []
    for b in range(5):
        n = 3000 + 700 * b
        pods = np.char.add("pod-", rng.integers(0, 40 + 25 * b, n).astype(str))
        ns = np.where(rng.random(n) < 0.3, "", np.char.add("ns", rng.integers(0, 5, n).astype(str)))
        labels = pa.array([None if i % 97 == 0 else '{"app":"a%d"}' % (i % 13) for i in range(n)], pa.string())
        tabs.append(pa.table({"p": pa.array(pods.tolist(), pa.string()), "ns": pa.array(ns.tolist(), pa.string()), "l": labels,
                              "b": pa.array([("ip%d" % (i % 7)).encode() for i in range(n)], pa.binary()),
                              "n": pa.array(rng.integers(0, 60000, n), pa.uint16())}))
    server.responses["SELECT p, ns, l, b, n FROM x"] = pa.concat_tables(tabs)
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    host = client.query_columns("SELECT p, ns, l, b, n FROM x", dict_strings=True)
    for chunk in (ch.STRING_CHUNK_BYTES, 20000, 1):          # one chunk, several record batches per chunk, every batch its own chunk
        monkeypatch.setattr(ch, "STRING_CHUNK_BYTES", chunk)
        gpu = client.query_columns("SELECT p, ns, l, b, n FROM x", dict_strings=True, engine=engine)
        assert set(gpu) == set(host)
        for name in ("p", "ns", "l", "b"):
            assert isinstance(gpu[name], ad.DictColumn)
            assert (gpu[name].codes == host[name].codes).all(), name
            assert gpu[name].values.tolist() == host[name].values.tolist(), name
        assert (gpu["n"] == host["n"]).all()


class _EncodeStandIn:
    """Test double for TadEngine.encode_strings (the GPU entry point is covered by the `-m gpu` test above): same contract — codes in order of
    first appearance, the first rows — computed with Arrow, so that the CHUNKING of query_columns(engine=...) is covered on the CPU tier."""

    def __init__(self):
        self.calls = []

    def encode_strings(self, arr):
        d = pc.dictionary_encode(arr.fill_null(""))
        codes = d.indices.to_numpy(zero_copy_only=False).astype(np.int64)
        first = np.full(len(d.dictionary), -1, np.int64)
        for i in range(codes.size - 1, -1, -1):
            first[codes[i]] = i
        self.calls.append(len(arr))
        return codes, first.astype(np.uint64)


import pyarrow.compute as pc  # noqa: E402


@pytest.mark.parametrize("chunk", [1 << 30, 4000, 1])
def test_query_columns_chunks_the_string_columns_for_the_engine(server, monkeypatch, chunk):
    """Record batches wait until a chunk's worth of column bytes is there, a chunk is one encode call, its codes are mapped into the column's
    unified dictionary (no gather for a column's first chunk), and the result equals the host path's — codes, values, order."""
    rng = np.random.default_rng(5)
    tabs = []
    for b in range(4):
        n = 300 + 50 * b
        tabs.append(pa.table({"p": pa.array(np.char.add("pod-", rng.integers(0, 12 + 9 * b, n).astype(str)).tolist(), pa.string()),
                              "l": pa.array([None if i % 31 == 0 else "lab%d" % (i % 5) for i in range(n)], pa.string()),
                              "n": pa.array(rng.integers(0, 60000, n), pa.uint16())}))
    server.responses["SELECT p, l, n FROM x"] = pa.concat_tables(tabs)
    client = ch.ClickHouseHTTP(server.url, user="", password="")
    host = client.query_columns("SELECT p, l, n FROM x", dict_strings=True)
    monkeypatch.setattr(ch, "STRING_CHUNK_BYTES", chunk)
    eng = _EncodeStandIn()
    got = client.query_columns("SELECT p, l, n FROM x", dict_strings=True, engine=eng)
    for name in ("p", "l"):
        assert (got[name].codes == host[name].codes).all() and got[name].values.tolist() == host[name].values.tolist(), name
    assert (got["n"] == host["n"]).all()
    rows = sum(len(t) for t in tabs)
    assert sum(eng.calls) == 2 * rows                                     # every row of both string columns went through the engine once
    assert len(eng.calls) == (2 if chunk == 1 << 30 else 2 * len(tabs) if chunk == 1 else len(eng.calls))
    # without dict_strings the engine is not consulted
    eng2 = _EncodeStandIn()
    plain = client.query_columns("SELECT p, l, n FROM x", engine=eng2)
    assert eng2.calls == [] and plain["p"].tolist() == host["p"].materialise().tolist()
