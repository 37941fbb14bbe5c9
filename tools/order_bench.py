#!/usr/bin/env python3
"""The C2 / C4 job on the SAME rows in the orders a caller can bring them in.  The synthetic table has its rows in arbitrary order; the
`flows` table is `ORDER BY (timeInserted, flowEndSeconds)` (create_table.sh:85), so a read without ORDER BY delivers them roughly by time,
and a pre-aggregated view (`pod_view_table`, create_table.sh:115) or a `GROUP BY` result delivers them by key.  Integer sum / max do not
depend on the order, so every order must give the arbitrary order's rows bit for bit; what changes is which queues of pass B fill.

usage: python tools/order_bench.py [--config c2|c4] [--rows N] [--jobs 10]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from theia_amd import TadEngine  # noqa: E402

CONFIGS = {"c2": dict(algo="EWMA", rows=100_000_000, keys=100_000, buckets=250, agg="svc"),
           "c4": dict(algo="DBSCAN", rows=100_000_000, keys=1_000_000, buckets=100, agg="")}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--jobs", type=int, default=10)
ap.add_argument("--library", default=None, help="a variant of libtad_mi355x.so (tools/build_variants.py)")
args = ap.parse_args()
cfg = CONFIGS[args.config]
n, K, T = args.rows or cfg["rows"], cfg["keys"], cfg["buckets"]
dev = torch.device("cuda", 0)
eng = TadEngine(0, library_path=args.library)
key = torch.empty(n, dtype=torch.int64, device=dev)
tend = torch.empty(n, dtype=torch.int64, device=dev)
val = torch.empty(n, dtype=torch.int64, device=dev)
eng.synth(0, n, K, T, into=(key, tend, val))


def order(name):
    if name == "arbitrary":
        return None
    if name == "ids by first appearance":                      # what any dictionary encoder hands out (tad_factorize, pandas.factorize, a Go map)
        first = torch.full((K,), n, dtype=torch.int64, device=dev).scatter_reduce(0, key, torch.arange(n, device=dev), "amin")
        newid = torch.empty(K, dtype=torch.int64, device=dev)
        newid[torch.sort(first).indices] = torch.arange(K, device=dev)
        return newid
    if name == "by time":
        return torch.sort(tend, stable=True).indices
    if name == "by time, 64 K-row blocks shuffled within":     # rows of a block arrive together, blocks in time order
        o = torch.sort(tend, stable=True).indices
        blk = torch.arange(n, device=dev) >> 16
        return o[torch.sort(blk * (1 << 20) + torch.randint(0, 1 << 20, (n,), device=dev)).indices]
    if name == "by key":
        return torch.sort(key, stable=True).indices
    if name == "by (key, time)":
        return torch.sort(key * 4096 + (tend - tend.min()) // 60, stable=True).indices
    if name == "by (time, key)":
        return torch.sort(((tend - tend.min()) // 60) * (1 << 21) + key, stable=True).indices
    raise ValueError(name)


want = None
print("%s: %d rows, %d keys, %d buckets, %s" % (args.config, n, K, T, cfg["algo"]))
for name in ("arbitrary", "ids by first appearance", "by time", "by time, 64 K-row blocks shuffled within", "by key", "by (key, time)", "by (time, key)"):
    o = order(name)
    relabel = name == "ids by first appearance"
    if relabel:
        k, t, v = o[key].contiguous(), tend, val
        inv = torch.sort(o).indices.cpu().numpy()                  # new id -> old id
    else:
        k, t, v = (key, tend, val) if o is None else (key[o].contiguous(), tend[o].contiguous(), val[o].contiguous())
    del o
    torch.cuda.synchronize()
    res = eng.run(cfg["algo"], k, t, v, K, agg_flow=cfg["agg"])
    got = {f: res[f].copy() for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
    st0 = res.stats
    res.close()
    if want is None:
        want = got
    if relabel:     # the same series under other ids: the rows of each key, in key order of the OLD ids, are the arbitrary order's
        old = inv[got["key_id"].astype("int64")]
        perm = old.argsort(kind="stable")
        got = {f: (old[perm].astype(got[f].dtype) if f == "key_id" else got[f][perm]) for f in got}
    same = all(got[f].shape == want[f].shape and (got[f] == want[f]).all() for f in want)
    ms, pa, pb, s0, det = [], [], [], [], []
    for _ in range(args.jobs):
        t0 = time.perf_counter()
        r = eng.run(cfg["algo"], k, t, v, K, agg_flow=cfg["agg"], out="device")
        st = r.stats
        r.close()
        ms.append((time.perf_counter() - t0) * 1e3)
        pa.append(st["ms_meta"]); pb.append(st["ms_scatter"]); s0.append(st["ms_stage0"]); det.append(st["ms_detect"])
    md = statistics.median
    print("  %-42s %7.3f ms/job (pass A %.3f, stage 0 %.3f [pass B %.3f], detect %.3f) path %d attempts %d/%d hist_sampled %d | %d rows %s"
          % (name, md(ms), md(pa), md(s0), md(pb), md(det), st["stage0_path"], st0["stage0_attempts"], st["stage0_attempts"],
             st["hist_sampled"], got["key_id"].size, "== arbitrary order's" if same else "DIFFER from the arbitrary order's"))
    del k, t, v

# keys that live for a part of the table only (pods / connections come and go): the rows in time order carry, at any moment, the keys alive
# then — every workgroup of pass B sees a narrow range of ids (ids by first appearance ascend with time) instead of all of them
for frac in (0.1, 0.01):
    W = max(int(K * frac), 1)
    i = torch.arange(n, device=dev)
    k_live = ((i.double() * ((K - W) / n)).long() + (key * 2654435761 % W)) % K       # a window of W ids sliding over the table
    t_live = tend.min() + 60 * ((i.double() * (T / n)).long())
    res = {}
    for name, o in (("in time order", None), ("the same rows shuffled", torch.sort(torch.randint(0, 1 << 40, (n,), device=dev)).indices)):
        k, t, v = (k_live, t_live, val) if o is None else (k_live[o].contiguous(), t_live[o].contiguous(), val[o].contiguous())
        del o
        torch.cuda.synchronize()      # (the columns were written on torch's stream, the engine reads them on its own)
        eng.close()
        eng = TadEngine(0, library_path=args.library)      # (a fresh engine: what the FIRST job on such a table does, nothing remembered)
        for _ in range(2):       # (the engine's buffers: two jobs on the hashed table; what it remembers is about that table)
            eng.run(cfg["algo"], key, tend, val, K, agg_flow=cfg["agg"], out="device").close()
        t0 = time.perf_counter()
        r = eng.run(cfg["algo"], k, t, v, K, agg_flow=cfg["agg"], out="device")
        first_ms = (time.perf_counter() - t0) * 1e3
        st0 = r.stats
        r.close()
        r = eng.run(cfg["algo"], k, t, v, K, agg_flow=cfg["agg"])
        res[name] = {f: r[f].copy() for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
        r.close()
        ms, pa, pb, s0 = [], [], [], []
        for _ in range(args.jobs):
            t0 = time.perf_counter()
            r = eng.run(cfg["algo"], k, t, v, K, agg_flow=cfg["agg"], out="device")
            st = r.stats
            r.close()
            ms.append((time.perf_counter() - t0) * 1e3)
            pa.append(st["ms_meta"]); pb.append(st["ms_scatter"]); s0.append(st["ms_stage0"])
        md = statistics.median
        print("  keys alive for %4.0f %% of the table, %-24s %7.3f ms/job (pass A %.3f, stage 0 %.3f [pass B %.3f]) path %d attempts %d/%d hist_sampled %d, "
              "the engine's first job on it %.3f ms | %d points, %d rows"
              % (frac * 100, name, md(ms), md(pa), md(s0), md(pb), st["stage0_path"], st0["stage0_attempts"], st["stage0_attempts"], st["hist_sampled"],
                 first_ms, st["n_points"], res[name]["key_id"].size))
        del k, t, v
    a, b = res["in time order"], res["the same rows shuffled"]
    print("    rows of the two orders %s" % ("identical" if all(a[f].shape == b[f].shape and (a[f] == b[f]).all() for f in a) else "DIFFER"))
    del k_live, t_live, i

# the job with a time window (theia tad run --start-time / --end-time -> anomaly_detection.py:581-586): every row is tested, pass A reads the
# time column in full, the generic forms of pass A / pass B run
eng.close()
eng = TadEngine(0, library_path=args.library)      # (a fresh engine: the contexts of the one above remember the sorted tables of this shape)
tstart = tend - 30
t_lo, t_hi = int(tend.min()), int(tend.max())
for name, kw in (("no window", {}),
                 ("end_time keeps 80 % of the buckets", dict(end_time=t_lo + (t_hi - t_lo) * 4 // 5)),
                 ("start_time + end_time keep the middle 60 %", dict(flow_start_s=tstart, start_time=t_lo + (t_hi - t_lo) // 5, end_time=t_lo + (t_hi - t_lo) * 4 // 5))):
    ms, pa, pb, s0 = [], [], [], []
    first = None
    for i in range(args.jobs + 2):
        t0 = time.perf_counter()
        r = eng.run(cfg["algo"], key, tend, val, K, agg_flow=cfg["agg"], out="device", **kw)
        st = r.stats
        r.close()
        if first is None:
            first = (st["stage0_attempts"], st["hist_sampled"])
        if i >= 2:
            ms.append((time.perf_counter() - t0) * 1e3)
            pa.append(st["ms_meta"]); pb.append(st["ms_scatter"]); s0.append(st["ms_stage0"])
    md = statistics.median
    print("  %-46s %7.3f ms/job (pass A %.3f, stage 0 %.3f [pass B %.3f]) rows used %d of %d, %d anomalies, attempts %d hist_sampled %d (first job: %d, %d)"
          % (name, md(ms), md(pa), md(s0), md(pb), st["rows_used"], n, st["n_anomalies"], st["stage0_attempts"], st["hist_sampled"], first[0], first[1]))
eng.close()
