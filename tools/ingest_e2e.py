#!/usr/bin/env python3
"""The job end to end as an operator sees it, ClickHouse's HTTP interface included: an in-process server streams the SELECT's result (Arrow record
batches of 65 536 rows, string columns as plain strings) -> theia_amd.clickhouse.fetch_flows -> prepare_columns -> tad_run, timed stage by stage,
(a) with the GPU ingest kernels (string columns through tad_encode_strings, key tuples through tad_factorize) and (b) with the host encode
(Arrow's dictionary_encode per record batch on one core; key tuples still on the GPU).  Not the headline metric (bench.py times tad_run on
device-resident columns); this is what stands in front of it.
usage: python tools/ingest_e2e.py [--rows 20000000] [--mode pod|svc|default]"""
import argparse
import io
import os
import sys
import threading
import time
import urllib.parse
from http.server import BaseHTTPRequestHandler, HTTPServer

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.ipc as ipc  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd import anomaly_detection as ad  # noqa: E402
from theia_amd import clickhouse as ch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=20_000_000)
ap.add_argument("--mode", default="pod", choices=["pod", "svc", "default"])
ap.add_argument("--pods", type=int, default=20_000)
args = ap.parse_args()
n, P = args.rows, args.pods
rng = np.random.default_rng(2)
agg = {"pod": "pod", "svc": "svc", "default": ""}[args.mode]
kw = dict(start_time="", end_time="", ns_ignore_list=["kube-system"], agg_flow=agg, pod_label="", external_ip="", svc_port_name="", pod_name="",
          pod_namespace="")
sql = ch.rows_query(kw["start_time"], kw["end_time"], kw["ns_ignore_list"], kw["agg_flow"], kw["pod_label"], kw["external_ip"], kw["svc_port_name"],
                    kw["pod_name"], kw["pod_namespace"])
cols = sql[len("SELECT "):sql.index(" FROM ")].split(", ")

# the flows of P pods in 40 namespaces talking to each other, 250 one-minute buckets (the C2 table's lattice)
pod_names = pa.array(["%s-%010x-%05x" % (("antrea-agent", "coredns", "flow-aggregator", "clickhouse-shard0", "web")[i % 5], i * 2654435761 % (1 << 40), i)
                      for i in range(P)])
pod_ns = pa.array(["ns-%02d" % (i % 40) for i in range(P)])
pod_labels = pa.array(['{"app":"a%03d","tier":"t%d"}' % (i % 300, i % 4) for i in range(P)])
pod_ip = pa.array(["10.%d.%d.%d" % (i >> 16 & 255, i >> 8 & 255, i & 255) for i in range(P)])
svc = pa.array(["ns-%02d/svc-%04d:http" % (i % 40, i % 5000) for i in range(P)])
src = pa.array(rng.integers(0, P, size=n))
dst = pa.array(rng.integers(0, P, size=n))
table = {}
for c in cols:
    if c == "sourcePodName": table[c] = pod_names.take(src)
    elif c == "destinationPodName": table[c] = pod_names.take(dst)
    elif c == "sourcePodNamespace": table[c] = pod_ns.take(src)
    elif c == "destinationPodNamespace": table[c] = pod_ns.take(dst)
    elif c == "sourcePodLabels": table[c] = pod_labels.take(src)
    elif c == "destinationPodLabels": table[c] = pod_labels.take(dst)
    elif c == "sourceIP": table[c] = pod_ip.take(src)
    elif c == "destinationIP": table[c] = pod_ip.take(dst)
    elif c == "destinationServicePortName": table[c] = svc.take(dst)
    elif c == "flowEndSeconds": table[c] = pa.array((1660202814 + 60 * rng.integers(0, 250, size=n)).astype("datetime64[s]"), pa.timestamp("s"))
    elif c == "flowStartSeconds": table[c] = pa.array(np.full(n, 1660199214).astype("datetime64[s]"), pa.timestamp("s"))
    elif c == "throughput": table[c] = pa.array(rng.integers(1_000_000_000, 4_000_000_000, size=n).astype(np.uint64), pa.uint64())
    elif c in ("sourceTransportPort", "destinationTransportPort"): table[c] = pa.array(rng.integers(1024, 65535, size=n).astype(np.uint16), pa.uint16())
    elif c == "protocolIdentifier": table[c] = pa.array(np.full(n, 6, dtype=np.uint16), pa.uint16())
    elif c == "flowType": table[c] = pa.array(np.full(n, 3, dtype=np.uint16), pa.uint16())
    else: raise SystemExit("column %s not generated" % c)
tab = pa.table(table)
sink = io.BytesIO()
with ipc.new_stream(sink, tab.schema) as w:
    w.write_table(tab, max_chunksize=65536)
payload = sink.getvalue()
del tab, table, sink


class Handler(BaseHTTPRequestHandler):
    def log_message(self, *a):
        pass

    def do_POST(self):
        self.rfile.read(int(self.headers.get("Content-Length", 0)))
        self.send_response(200)
        self.send_header("Content-Length", str(len(payload)))
        self.end_headers()
        self.wfile.write(payload)


httpd = HTTPServer(("127.0.0.1", 0), Handler)
threading.Thread(target=httpd.serve_forever, daemon=True).start()
client = ch.ClickHouseHTTP("http://127.0.0.1:%d" % httpd.server_address[1], user="", password="")
eng = TadEngine(device=0)
args_pos = (kw["start_time"], kw["end_time"], kw["ns_ignore_list"], kw["agg_flow"], kw["pod_label"], kw["external_ip"], kw["svc_port_name"], kw["pod_name"],
            kw["pod_namespace"])
print("%s mode, %d rows, %d pods, %d columns (%s), Arrow stream %.2f GB" % (args.mode, n, P, len(cols), ", ".join(cols), len(payload) / 1e9), flush=True)
results = {}
for label, ingest_engine in (("GPU ingest (tad_encode_strings + tad_factorize)", eng), ("host encode (Arrow dictionary_encode per batch) + tad_factorize", None),
                             ("GPU ingest, second run", eng)):
    t0 = time.perf_counter()
    flows = ch.fetch_flows(client, *args_pos, engine=ingest_engine)
    t1 = time.perf_counter()
    prep = ad.prepare_columns(flows, *args_pos, engine=eng)
    t2 = time.perf_counter()
    res = eng.run("EWMA", prep.key_id, prep.flow_end_s, prep.value, max(prep.num_keys, 1), agg_flow=agg, key_id2=prep.key_id2, flow_start_s=prep.flow_start_s,
                  start_time=prep.start_time, end_time=prep.end_time)
    t3 = time.perf_counter()
    results[label] = (res.n_rows, prep.num_keys)
    print("%-70s read + decode %.2f s | prepare_columns %.2f s | tad_run (host columns) %.3f s | total %.2f s = %.2e rows/s; %d keys, %d anomalies"
          % (label, t1 - t0, t2 - t1, t3 - t2, t3 - t0, n / (t3 - t0), prep.num_keys, res.n_rows), flush=True)
    del flows, prep, res
assert len(set(results.values())) == 1, results        # the three runs are the same job
# the socket alone: how fast this process can pull the stream and parse the record batches without touching the strings
t0 = time.perf_counter()
with client._request({}, (sql + " FORMAT ArrowStream").encode()) as resp:
    rows = sum(b.num_rows for b in ipc.open_stream(resp))
dt = time.perf_counter() - t0
print("socket + Arrow IPC framing alone: %.2f s = %.2e rows/s (%.2f GB/s)" % (dt, rows / dt, len(payload) / dt / 1e9))
eng.close()
httpd.shutdown()
