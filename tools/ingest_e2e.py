#!/usr/bin/env python3
"""The job end to end as an operator sees it, ClickHouse's HTTP interface included: an in-process server answers
the count query and the G per-shard row queries with Arrow streams whose string columns are DICTIONARY arrays with a dictionary per record
batch (what ClickHouse sends for `toLowCardinality(col)` under output_format_arrow_low_cardinality_as_dictionary = 1, blocks of
max_block_size rows) -> theia_amd.clickhouse.fetch_flows_device (G parallel reads straight into HBM) -> prepare_columns_device (predicates as
device gathers, key tuples factorised on the GPU) -> tad_run on device columns -> the anomaly rows on the host.  Timed stage by stage; the
first run is cold (page-locked buffers are allocated), the following ones warm.  `--compare-host ROWS` also runs the single-connection host
path (fetch_flows + prepare_columns, plain string batches) on the first ROWS rows and checks that both paths give the same job.
Not the headline metric (bench.py times tad_run on device-resident columns); this is what stands in front of it.

usage: python tools/ingest_e2e.py [--rows 100000000] [--mode pod|svc|default] [--connections 8] [--no-pinned] [--runs 3]"""
import argparse
import io
import os
import sys
import threading
import time
import urllib.parse
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.ipc as ipc  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd import anomaly_detection as ad  # noqa: E402
from theia_amd import clickhouse as ch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100_000_000)
ap.add_argument("--mode", default="pod", choices=["pod", "svc", "default"])
ap.add_argument("--pods", type=int, default=20_000)
ap.add_argument("--connections", type=int, default=8)
ap.add_argument("--block", type=int, default=4_194_304, help="rows per record batch (ClickHouse: max_block_size)")
ap.add_argument("--no-pinned", action="store_true")
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--compare-host", type=int, default=0, help="also run the host path on this many rows and compare the jobs")
args = ap.parse_args()
n, P, G = args.rows, args.pods, args.connections
agg = {"pod": "pod", "svc": "svc", "default": ""}[args.mode]
kw = dict(start_time="", end_time="", ns_ignore_list=["kube-system"], agg_flow=agg, pod_label="", external_ip="", svc_port_name="", pod_name="",
          pod_namespace="")
pos = (kw["start_time"], kw["end_time"], kw["ns_ignore_list"], kw["agg_flow"], kw["pod_label"], kw["external_ip"], kw["svc_port_name"], kw["pod_name"],
       kw["pod_namespace"])
sql0 = ch.rows_query(*pos, dictionary=True, shard=(0, G))
cols = [c.split(" AS ")[-1] for c in sql0[len("SELECT "):sql0.index(" FROM ")].split(", ")]

# the flows of P pods in 40 namespaces talking to each other, 250 one-minute buckets (the C2 table's lattice)
per_pod = {
    "PodName": pa.array(["%s-%010x-%05x" % (("antrea-agent", "coredns", "flow-aggregator", "clickhouse-shard0", "web")[i % 5], i * 2654435761 % (1 << 40), i)
                         for i in range(P)]),
    "PodNamespace": pa.array(["ns-%02d" % (i % 40) for i in range(P)]),
    "PodLabels": pa.array(['{"app":"a%03d","tier":"t%d"}' % (i % 300, i % 4) for i in range(P)]),
    "IP": pa.array(["10.%d.%d.%d" % (i >> 16 & 255, i >> 8 & 255, i & 255) for i in range(P)]),
    "ServicePortName": pa.array(["ns-%02d/svc-%04d:http" % (i % 40, i % 5000) for i in range(P)]),
}


def shard_payload(g):
    """the Arrow stream of shard g: record batches of `block` rows, every string column a dictionary array whose dictionary holds the
    values of THAT batch (unique + inverse of the pod index; the strings are only touched per distinct value)"""
    rng = np.random.default_rng(1000 + g)
    rows = n // G + (1 if g < n % G else 0)
    sink = io.BytesIO()
    writer = None
    for at in range(0, rows, args.block):
        k = min(args.block, rows - at)
        side = {}
        for s in ("source", "destination"):
            u, inv = np.unique(rng.integers(0, P, size=k), return_inverse=True)
            side[s] = (pa.array(u), pa.array(inv.astype(np.int32), pa.int32()))
        arrays = []
        for c in cols:
            s = "source" if c.startswith("source") else "destination"
            tail = c[len(s):]
            if tail in per_pod:
                u, inv = side[s]
                arrays.append(pa.DictionaryArray.from_arrays(inv, per_pod[tail].take(u)))
            elif c == "flowEndSeconds":
                arrays.append(pa.array((1660202814 + 60 * rng.integers(0, 250, size=k)).astype(np.uint32), pa.uint32()))
            elif c == "flowStartSeconds":
                arrays.append(pa.array(np.full(k, 1660199214, dtype=np.uint32), pa.uint32()))
            elif c == "throughput":
                arrays.append(pa.array(rng.integers(1_000_000_000, 4_000_000_000, size=k).astype(np.uint64), pa.uint64()))
            elif c in ("sourceTransportPort", "destinationTransportPort"):
                arrays.append(pa.array(rng.integers(1024, 65535, size=k).astype(np.uint16), pa.uint16()))
            elif c == "protocolIdentifier":
                arrays.append(pa.array(np.full(k, 6, dtype=np.uint8), pa.uint8()))
            elif c == "flowType":
                arrays.append(pa.array(np.full(k, 3, dtype=np.uint8), pa.uint8()))
            else:
                raise SystemExit("column %s not generated" % c)
        batch = pa.record_batch(arrays, names=cols)
        if writer is None:
            writer = ipc.new_stream(sink, batch.schema)
        writer.write_batch(batch)
    if writer is not None:
        writer.close()
    return rows, sink.getvalue()


t0 = time.perf_counter()
shards = [None] * G
ths = [threading.Thread(target=lambda g=g: shards.__setitem__(g, shard_payload(g))) for g in range(G)]
for th in ths:
    th.start()
for th in ths:
    th.join()
total_bytes = sum(len(p) for _, p in shards)
print("%s mode, %d rows, %d pods, %d columns (%s); %d shard streams of %d-row dictionary batches, %.2f GB (%.1f B/row), built in %.1f s"
      % (args.mode, n, P, len(cols), ", ".join(cols), G, args.block, total_bytes / 1e9, total_bytes / n, time.perf_counter() - t0), flush=True)


class Handler(BaseHTTPRequestHandler):
    def log_message(self, *a):
        pass

    def do_POST(self):
        sql = self.rfile.read(int(self.headers.get("Content-Length", 0))).decode()
        if " AS shard, count() AS rows " in sql:
            t = pa.table({"shard": pa.array(np.arange(G), pa.uint64()), "rows": pa.array([r for r, _ in shards], pa.uint64())})
            sink = io.BytesIO()
            with ipc.new_stream(sink, t.schema) as w:
                w.write_table(t)
            body = sink.getvalue()
        else:
            g = int(sql[:-len(" FORMAT ArrowStream")].rsplit(" = ", 1)[1])
            body = shards[g][1]
        self.send_response(200)
        self.send_header("Content-Length", str(len(body)))
        self.end_headers()
        self.wfile.write(body)


httpd = ThreadingHTTPServer(("127.0.0.1", 0), Handler)
threading.Thread(target=httpd.serve_forever, daemon=True).start()
client = ch.ClickHouseHTTP("http://127.0.0.1:%d" % httpd.server_address[1], user="", password="")
eng = TadEngine(device=0)
results = []
for run in range(args.runs):
    tm = {}
    t0 = time.perf_counter()
    dev = ch.fetch_flows_device(client, eng, *pos, connections=G, pinned=not args.no_pinned, timings=tm)
    t1 = time.perf_counter()
    prep = ad.prepare_columns_device(dev, *pos, engine=eng)
    t2 = time.perf_counter()
    res = eng.run("EWMA", prep.key_id, prep.flow_end_s, prep.value, max(prep.num_keys, 1), agg_flow=agg, key_id2=prep.key_id2,
                  flow_start_s=prep.flow_start_s, start_time=prep.start_time, end_time=prep.end_time, out="device", key_hist=prep.key_hist)
    t3 = time.perf_counter()
    out_cols = ad.result_columns(prep, res, "EWMA", agg, "e2e")
    t4 = time.perf_counter()
    results.append((res.n_rows, prep.num_keys, res.stats["n_points"]))
    print("run %d (%s): read + upload %.3f s (count query %.3f, slowest read %.3f, slowest parse + upload %.3f) | prepare_columns_device %.3f s | "
          "tad_run %.4f s | result columns to host %.3f s | total %.3f s = %.3e rows/s; %d keys, %d anomalies, socket %.2f GB/s aggregate"
          % (run, "cold" if run == 0 else "warm", t1 - t0, tm["count_query_s"], tm["slowest_read_s"], tm["slowest_parse_upload_s"], t2 - t1, t3 - t2, t4 - t3,
             t4 - t0, n / (t4 - t0), prep.num_keys, res.n_rows, tm["bytes"] / max(tm["slowest_read_s"], 1e-9) / 1e9), flush=True)
    res.close()
    for c in dev.values():
        (c.codes if hasattr(c, "codes") else c).free()
    del dev, prep, res, out_cols
assert len(set(results)) == 1, results        # every run is the same job

if args.compare_host:
    # the single-connection host path on a prefix of shard 0's rows (plain string batches), against the device path on the same rows
    m = min(args.compare_host, shards[0][0])
    tab = ipc.open_stream(pa.py_buffer(shards[0][1])).read_all().slice(0, m)
    plain = pa.table({c: (tab[c].cast(pa.string()) if pa.types.is_dictionary(tab[c].type) else tab[c]) for c in tab.column_names})
    flows = {}
    for c in plain.column_names:
        col = plain[c].combine_chunks()
        flows[c] = np.asarray(col.to_pylist(), dtype=str) if pa.types.is_string(col.type) else col.to_numpy(zero_copy_only=False).astype(np.int64 if c != "throughput" else np.uint64)
    t0 = time.perf_counter()
    prep_h = ad.prepare_columns(flows, *pos, engine=eng)
    res_h = eng.run("EWMA", prep_h.key_id, prep_h.flow_end_s, prep_h.value, max(prep_h.num_keys, 1), agg_flow=agg, key_id2=prep_h.key_id2)
    rows_h = ad.result_rows(prep_h, res_h, "EWMA", agg, "e2e")
    th_ = time.perf_counter() - t0
    shards = [(m, None)] + [(0, None)] * (G - 1)
    sink = io.BytesIO()
    with ipc.new_stream(sink, tab.schema) as w:
        w.write_table(tab, max_chunksize=1 << 20)
    shards[0] = (m, sink.getvalue())
    empty = io.BytesIO()
    with ipc.new_stream(empty, tab.schema):
        pass
    for g in range(1, G):
        shards[g] = (0, empty.getvalue())
    dev = ch.fetch_flows_device(client, eng, *pos, connections=G, pinned=not args.no_pinned)
    prep_d = ad.prepare_columns_device(dev, *pos, engine=eng)
    res_d = eng.run("EWMA", prep_d.key_id, prep_d.flow_end_s, prep_d.value, max(prep_d.num_keys, 1), agg_flow=agg, key_id2=prep_d.key_id2, key_hist=prep_d.key_hist)
    rows_d = ad.result_rows(prep_d, res_d, "EWMA", agg, "e2e")
    import json
    canon = lambda rows: sorted(json.dumps(r, sort_keys=True) for r in rows)
    assert canon(rows_h) == canon(rows_d), "host path and device path disagree"
    print("host path (prepare_columns on string arrays + tad_run, %d rows, no transport): %.2f s; %d anomaly rows identical to the device path's" % (m, th_, len(rows_h)))
eng.close()
httpd.shutdown()
