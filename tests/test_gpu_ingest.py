"""GPU: the columnar device ingest — ClickHouse's ArrowStream over several connections, string columns as Arrow
DICTIONARY arrays (a dictionary per record batch), straight into HBM (theia_amd.clickhouse.fetch_flows_device), predicates applied as device
gathers over per-value verdicts and key tuples factorised on the GPU (theia_amd.anomaly_detection.prepare_columns_device).  The job's rows
must be those of the host path (fetch_flows + prepare_columns) on the 13 mode / filter cases, through dictionary batches and through plain
string batches (a server that ignores the dictionary settings): 26 cases.  The kernels underneath (tad_widen_column, tad_mask_rows) are
checked against numpy."""
import io
import threading
import urllib.parse
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import numpy as np
import pyarrow as pa
import pyarrow.ipc as ipc
import pytest

from oracle import job_oracle as jo
from theia_amd import TadError
from theia_amd import anomaly_detection as ad
from theia_amd import clickhouse as ch
from theia_amd.engine import DeviceArray, HostBuffer
from test_host_job import CASES, canon

pytestmark = pytest.mark.gpu
KW = dict(start_time="", end_time="", ns_ignore_list=(), agg_flow="", pod_label="", external_ip="", svc_port_name="", pod_name="", pod_namespace="")


def test_widen_column_every_width_tables_and_gathers(engine):
    rng = np.random.default_rng(1)
    n = 100_003
    dst = DeviceArray(engine, n + 10, np.int64)
    for dt in (np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64):
        info = np.iinfo(dt)
        src = rng.integers(max(info.min, -2**62), min(info.max, 2**62), size=n, dtype=np.int64).astype(dt)
        engine.widen_into(dst, 5, src.ctypes.data, 8 * src.itemsize, dt().dtype.kind == "i", n)
        assert (dst.to_host()[5:5 + n] == src.astype(np.int64)).all(), dt
    # a dictionary batch: indices -> job-wide codes through the batch's remap table
    table = rng.permutation(5000).astype(np.int64) * 977 - 3
    dtab = DeviceArray.from_host(engine, table)
    for dt in (np.int8, np.uint16, np.int32, np.int64):
        idx = rng.integers(0, min(5000, np.iinfo(dt).max), size=n).astype(dt)
        engine.widen_into(dst, 0, idx.ctypes.data, 8 * idx.itemsize, dt().dtype.kind == "i", n, table=dtab)
        assert (dst.to_host()[:n] == table[idx.astype(np.int64)]).all(), dt
    # an index outside the table (and a negative one) is an error, not a wild read
    bad = np.array([0, 1, 5000], dtype=np.int32)
    with pytest.raises(TadError):
        engine.widen_into(dst, 0, bad.ctypes.data, 32, True, 3, table=dtab)
    neg = np.array([0, -1], dtype=np.int32)
    with pytest.raises(TadError):
        engine.widen_into(dst, 0, neg.ctypes.data, 32, True, 2, table=dtab)
    with pytest.raises(TadError):
        engine.widen_into(dst, n, bad.ctypes.data, 32, True, 11)        # does not fit the column
    # gather of a device column at device row numbers (how the key table is read back)
    col = DeviceArray.from_host(engine, table)
    rows = DeviceArray.from_host(engine, np.array([4999, 0, 17, 17], dtype=np.uint64))
    assert engine.gather(col, rows).tolist() == table[[4999, 0, 17, 17]].tolist()
    # page-locked source
    hb = HostBuffer(engine, 8 * 1000)
    a = np.frombuffer(hb.view, dtype=np.uint64)
    a[:] = np.arange(1000, dtype=np.uint64) * np.uint64(3)
    engine.widen_into(dst, 0, hb.ptr, 64, False, 1000)
    assert (dst.to_host()[:1000] == np.arange(1000) * 3).all()
    del a
    hb.free()


def test_mask_rows_is_the_and_of_per_value_verdicts(engine):
    rng = np.random.default_rng(2)
    n = 77_777
    codes = [rng.integers(0, d, size=n).astype(np.int64) for d in (7, 300, 65536)]
    masks = [rng.random(d) < p for d, p in ((7, 0.7), (300, 0.5), (65536, 0.9))]
    dcodes = [DeviceArray.from_host(engine, c) for c in codes]
    keep = engine.mask_rows(n, list(zip(dcodes, masks)))
    got = np.frombuffer(keep.to_host().tobytes(), dtype=np.uint8)[:n]
    want = masks[0][codes[0]] & masks[1][codes[1]] & masks[2][codes[2]]
    assert (got.astype(bool) == want).all()
    keep = engine.mask_rows(n, [(dcodes[0], ~masks[0])], keep=keep)          # combined into an existing mask
    got = np.frombuffer(keep.to_host().tobytes(), dtype=np.uint8)[:n]
    assert not got.any()
    allrows = engine.mask_rows(n, [])
    assert np.frombuffer(allrows.to_host().tobytes(), dtype=np.uint8)[:n].all()
    with pytest.raises(TadError):
        engine.mask_rows(n, [(dcodes[1], masks[0])])                         # a code outside its mask


class ShardServer:
    """ClickHouse's HTTP interface as far as the device ingest uses it: the count query, the per-shard row queries (answered from `table`
    split by row number mod G — which rows a shard holds does not matter to the client, only that the counts match), string columns as
    Arrow dictionary arrays with a dictionary PER RECORD BATCH (what output_format_arrow_low_cardinality_as_dictionary produces) or as plain
    strings; `block` rows per record batch."""

    def __init__(self, table, dictionaries=True, block=700, lie_about_counts=False):
        self.table, self.dictionaries, self.block, self.lie = table, dictionaries, block, lie_about_counts
        self.queries = []
        owner = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def do_POST(self):
                sql = self.rfile.read(int(self.headers.get("Content-Length", 0))).decode()
                params = urllib.parse.parse_qs(urllib.parse.urlparse(self.path).query)
                owner.queries.append((sql, params))
                body = owner.answer(sql)
                self.send_response(200)
                if owner.block % 2 == 0:          # half the servers announce the length, the others make the client grow its buffer
                    self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

        self.httpd = ThreadingHTTPServer(("127.0.0.1", 0), Handler)
        self.url = "http://127.0.0.1:%d" % self.httpd.server_address[1]
        threading.Thread(target=self.httpd.serve_forever, daemon=True).start()

    def answer(self, sql):
        assert sql.endswith(" FORMAT ArrowStream")
        sql = sql[:-len(" FORMAT ArrowStream")]
        n = self.table.num_rows
        if sql.startswith("SELECT cityHash64(") and " AS shard, count() AS rows " in sql:
            G = int(sql.split(") % ")[1].split(" ")[0])
            rows = np.bincount(np.arange(n) % G, minlength=G)
            if self.lie:
                rows[0] += 1
            t = pa.table({"shard": pa.array(np.arange(G), pa.uint64()), "rows": pa.array(rows, pa.uint64())})
        else:
            tail = sql.rsplit(") % ", 1)[1]
            G, g = int(tail.split(" = ")[0]), int(tail.split(" = ")[1])
            names = [c.split(" AS ")[-1] for c in sql[len("SELECT "):sql.index(" FROM ")].split(", ")]
            t = self.table.select(names).take(pa.array(np.arange(g, n, G)))
        sink = io.BytesIO()
        batches = t.to_batches(max_chunksize=self.block)
        if self.dictionaries:
            enc = []
            for b in batches:
                cols = [c.dictionary_encode() if pa.types.is_string(c.type) else c for c in b.columns]
                enc.append(pa.record_batch(cols, names=b.schema.names))
            batches = enc
        schema = batches[0].schema if batches else t.schema
        with ipc.new_stream(sink, schema) as w:
            for b in batches:
                w.write_batch(b)
        return sink.getvalue()

    def close(self):
        self.httpd.shutdown()


def flows_table(flows):
    """the synthetic flows typed the way ClickHouse types default.flows (create_table.sh:31-85): DateTime -> UInt32, ports UInt16, ..."""
    arrays = {}
    for name, v in flows.items():
        v = np.asarray(v)
        if name.endswith("Seconds"):
            arrays[name] = pa.array(v.astype(np.uint32), pa.uint32())          # ClickHouse sends DateTime as UInt32 in Arrow
        elif name == "throughput":
            arrays[name] = pa.array(v.astype(np.uint64), pa.uint64())
        elif name in ("protocolIdentifier", "flowType"):
            arrays[name] = pa.array(v.astype(np.uint8), pa.uint8())
        elif v.dtype.kind in "iu":
            arrays[name] = pa.array(v.astype(np.uint16), pa.uint16())
        else:
            arrays[name] = pa.array(v.astype(str).tolist(), pa.string())
    return pa.table(arrays)


def job_rows(engine, prep, agg_flow):
    res = engine.run("EWMA", prep.key_id, prep.flow_end_s, prep.value, max(prep.num_keys, 1), agg_flow=agg_flow, key_id2=prep.key_id2,
                     flow_start_s=prep.flow_start_s, start_time=prep.start_time, end_time=prep.end_time)
    return ad.result_rows(prep, res, "EWMA", agg_flow, "j"), res.stats


@pytest.mark.parametrize("dictionaries", [True, False], ids=["dictionary-batches", "plain-strings"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_device_ingest_gives_the_rows_of_the_host_path(engine, case, dictionaries):
    flows = jo.synth_flows(6000)
    kw = dict(KW)
    kw.update(case)
    server = ShardServer(flows_table(flows), dictionaries=dictionaries, block=700 if dictionaries else 901)
    try:
        client = ch.ClickHouseHTTP(server.url, user="", password="")
        timings = {}
        dev = ch.fetch_flows_device(client, engine, connections=3, pinned=dictionaries, timings=timings, **kw)
        assert timings["rows"] == 6000 and timings["connections"] == 3 and timings["bytes"] > 0
        prep_dev = ad.prepare_columns_device(dev, engine=engine, **kw)
        got, st_dev = job_rows(engine, prep_dev, kw["agg_flow"])
        prep_host = ad.prepare_columns(flows, **kw)
        want, st_host = job_rows(engine, prep_host, kw["agg_flow"])
        assert canon(got) == canon(want)
        assert got[0]["anomaly"] == "true" or len(got) == 1
        for f in ("rows_used", "n_keys", "n_points", "n_anomalies"):
            assert st_dev[f] == st_host[f], f
        assert prep_dev.num_keys == prep_host.num_keys
        # every row query asked for dictionary columns, a shard each, and the settings that make ClickHouse send dictionaries
        row_queries = [(q, p) for q, p in server.queries if "count() AS rows" not in q]
        assert len(row_queries) == 3 and {q.rsplit(" = ", 1)[1].split(" ")[0] for q, _ in row_queries} == {"0", "1", "2"}
        assert all("toLowCardinality(" in q and p["output_format_arrow_low_cardinality_as_dictionary"] == ["1"] for q, p in row_queries)
    finally:
        server.close()


def test_device_ingest_notices_a_table_that_changed_between_count_and_read(engine):
    flows = jo.synth_flows(2000)
    server = ShardServer(flows_table(flows), lie_about_counts=True)
    try:
        client = ch.ClickHouseHTTP(server.url, user="", password="")
        with pytest.raises(RuntimeError):
            ch.fetch_flows_device(client, engine, connections=2, agg_flow="svc")
    finally:
        server.close()


def test_device_ingest_of_an_empty_table(engine):
    flows = {k: np.asarray(v)[:0] for k, v in jo.synth_flows(10).items()}
    server = ShardServer(flows_table(flows))
    try:
        client = ch.ClickHouseHTTP(server.url, user="", password="")
        dev = ch.fetch_flows_device(client, engine, connections=2, agg_flow="svc")
        prep = ad.prepare_columns_device(dev, agg_flow="svc", engine=engine)
        rows, _ = job_rows(engine, prep, "svc")
        assert len(rows) == 1 and rows[0]["anomaly"] == "NO ANOMALY DETECTED"
    finally:
        server.close()
