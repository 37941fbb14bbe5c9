#!/usr/bin/env python3
"""Host-side cost of the job OUTSIDE tad_run: prepare_columns (string predicates + dictionary encoding of the key columns,
theia_amd/anomaly_detection.py) on a svc-mode table with C2's shape.  The numbers go into DESIGN.md section 5 next to the
kernel times: with a real ClickHouse ingest this, not the 1.4 ms of GPU time, is what an operator waits for.
With a GPU (round 4) the same calls are repeated with engine=TadEngine: the key tuples are factorised by tad_factorize instead of pandas (host
columns: the copies to and from the GPU are inside the timing), and the default mode's six key columns are timed both ways.
usage: python tools/host_prepare_timing.py [rows] [keys]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import anomaly_detection as ad  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keys = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
rng = np.random.default_rng(1)
names = np.array(["svc-%06d:http" % i for i in range(keys)])      # 15-character service port names
t0 = time.perf_counter()
flows = {"destinationServicePortName": names[rng.integers(0, keys, size=rows)],
         "flowEndSeconds": 1660202814 + 60 * rng.integers(0, 250, size=rows),
         "flowStartSeconds": np.full(rows, 1660202000, dtype=np.int64),
         "throughput": rng.integers(1_000_000_000, 4_000_000_000, size=rows).astype(np.uint64),
         "sourcePodNamespace": np.full(rows, "default"), "destinationPodNamespace": np.full(rows, "default")}
t_gen = time.perf_counter() - t0
import pyarrow as pa  # noqa: E402
import pyarrow.compute as pc  # noqa: E402

t0 = time.perf_counter()
arr = pa.array(flows["destinationServicePortName"])             # what arrives over the wire: an Arrow string column
t_arrow = time.perf_counter() - t0
t0 = time.perf_counter()
d = pc.dictionary_encode(arr.combine_chunks() if hasattr(arr, 'combine_chunks') else arr)                                   # clickhouse.query_columns(dict_strings=True) does this per record batch
enc = dict(flows)
enc["destinationServicePortName"] = ad.DictColumn(d.indices.to_numpy(zero_copy_only=False), np.asarray(d.dictionary.to_pylist()))
_ns = pa.array(flows["sourcePodNamespace"])
ns = pc.dictionary_encode(_ns.combine_chunks() if hasattr(_ns, 'combine_chunks') else _ns)
enc["sourcePodNamespace"] = ad.DictColumn(ns.indices.to_numpy(zero_copy_only=False), np.asarray(ns.dictionary.to_pylist()))
enc["destinationPodNamespace"] = enc["sourcePodNamespace"]
t_enc = time.perf_counter() - t0
for label, kw in (("svc, no filters", dict(agg_flow="svc")), ("svc, ns-ignore-list + time window", dict(agg_flow="svc", ns_ignore_list=["kube-system"],
                                                                                                       start_time="2022-08-11 00:00:00", end_time="2022-08-12 00:00:00"))):
    t0 = time.perf_counter()
    prep = ad.prepare_columns(enc, **kw)
    dt = time.perf_counter() - t0
    print("dictionary-encoded columns: prepare_columns [%s]: %.2f s = %.2e rows/s (+ Arrow dictionary_encode of the string columns %.2f s = %.2e rows/s), %d keys"
          % (label, dt, rows / dt, t_enc, rows / t_enc, prep.num_keys))
for label, kw in (("svc, no filters", dict(agg_flow="svc")), ("svc, ns-ignore-list + time window", dict(agg_flow="svc", ns_ignore_list=["kube-system"],
                                                                                                       start_time="2022-08-11 00:00:00", end_time="2022-08-12 00:00:00"))):
    t0 = time.perf_counter()
    prep = ad.prepare_columns(flows, **kw)
    dt = time.perf_counter() - t0
    print("prepare_columns [%s]: %d rows / %d keys in %.2f s = %.2e rows/s on one host core (%d distinct keys found); table generation %.1f s"
          % (label, rows, keys, dt, rows / dt, prep.num_keys, t_gen))

# ---- round 4: the same with the GPU factorisation (tad_factorize), if there is a GPU ----
try:
    from theia_amd import TadEngine
    eng = TadEngine(device=0)
except Exception as exc:      # no GPU here: the host figures above are all there is
    print("no engine (%s): GPU factorisation not timed" % exc)
    sys.exit(0)
ad.prepare_columns(enc, agg_flow="svc", engine=eng)       # warm-up: scratch buffers
for label, kw in (("svc, no filters", dict(agg_flow="svc")), ("svc, ns-ignore-list + time window", dict(agg_flow="svc", ns_ignore_list=["kube-system"],
                                                                                                       start_time="2022-08-11 00:00:00", end_time="2022-08-12 00:00:00"))):
    t0 = time.perf_counter()
    prep = ad.prepare_columns(enc, engine=eng, **kw)
    dt = time.perf_counter() - t0
    print("dictionary-encoded columns, GPU factorisation: prepare_columns [%s]: %.2f s = %.2e rows/s (host columns: PCIe copies included), %d keys" % (label, dt, rows / dt, prep.num_keys))
# the default mode (agg_flow None): six key columns per row (anomaly_detection.py:52-61)
conn = rng.integers(0, max(1, rows // 100), size=rows)
ips = np.array(["10.%d.%d.%d" % (i >> 16 & 255, i >> 8 & 255, i & 255) for i in range(4096)])
none_flows = {"sourceIP": ad.DictColumn(conn % 4096, ips), "destinationIP": ad.DictColumn((conn * 7 + 3) % 4096, ips),
              "sourceTransportPort": 1024 + conn % 50000, "destinationTransportPort": 80 + conn % 7, "protocolIdentifier": 6 + (conn % 2) * 11,
              "flowStartSeconds": 1660200000 + conn // 16, "flowEndSeconds": flows["flowEndSeconds"], "throughput": flows["throughput"],
              "sourcePodNamespace": enc["sourcePodNamespace"], "destinationPodNamespace": enc["destinationPodNamespace"]}
for label, e in (("pandas MultiIndex, one core", None), ("GPU factorisation", eng)):
    t0 = time.perf_counter()
    prep = ad.prepare_columns(none_flows, agg_flow="", engine=e)
    dt = time.perf_counter() - t0
    print("default mode, six key columns [%s]: %.2f s = %.2e rows/s, %d keys" % (label, dt, rows / dt, prep.num_keys))
eng.close()
