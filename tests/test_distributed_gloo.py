"""The N>1 path on CPU: world_size 2 and 3 over gloo.  Key-sharded and row-sharded (all-to-all) ingest, the
counter all-reduce / moment all-gather, the sentinel decision.  The per-rank compute is stood in by the oracle
(CPU ranks have no GPU); on the GPU box the same host code drives TadEngine.run (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tad_oracle as orc
from theia_amd import distributed as td


class OracleResult:
    def __init__(self, want):
        pk, pt, pv = want["points"]
        x = orc.u64_to_f64(pv)
        mean = float(x.mean()) if x.size else 0.0
        self.rows = {k: want[k] for k in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
        self.stats = {"n_anomalies": want["n_anomalies"], "n_keys": want["n_keys"], "n_points": want["n_points"],
                      "rows_used": 0, "keys_no_result": want["keys_no_result"], "rows_in": 0,
                      "pts_mean": mean, "pts_m2": float(((x - mean) ** 2).sum()) if x.size else 0.0}


def oracle_run(algo, key_id, flow_end_s, value, num_keys, key_id2=None, flow_start_s=None, **job):
    return OracleResult(orc.run_job(algo, key_id, flow_end_s, value, key_id2=key_id2, flow_start_s=flow_start_s, **job))


def table(pod):
    k, t, v = orc.synth_rows(0, 40000, 37, 50)
    k2 = None
    if pod:
        k2 = orc.mix64(k + np.uint64(99)) % np.uint64(37)
        k2 = np.where(k2 % np.uint64(5) == 0, td.SKIP, k2)          # some rows have no second key
        k = np.where(k % np.uint64(7) == 0, td.SKIP, k)             # some have no first key
    return k, t, v, k2


def worker(rank, world, port, mode, algo, pod, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, t, v, k2 = table(pod)
        if mode == "key":
            cols = td.shard_rows(rank, world, k, t, v, k2)
        else:   # rows arrive in arbitrary slices; one all-to-all(v) brings them to their owners
            sl = slice(rank, None, world)
            cols = td.exchange_rows({"key_id": k[sl], "flow_end_s": t[sl], "value": v[sl], "key_id2": None if k2 is None else k2[sl]},
                                    world, rank)
        red = td.JobReducer()
        res, glob = td.run_sharded(oracle_run, algo, cols, 37, red, agg_flow="pod" if pod else "svc")
        rows = dict(res.rows)
        rows["key_id"] = td.global_key(rows["key_id"], rank, world)
        q.put((rank, rows, glob))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,mode,algo,pod", [(2, "key", "EWMA", False), (2, "rows", "EWMA", True), (3, "key", "DBSCAN", True),
                                                 (2, "rows", "DBSCAN", False), (2, "rows", "ARIMA", False), (2, "key", "ARIMA", False)])
def test_sharded_job_equals_single_process(world, mode, algo, pod):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, mode, algo, pod, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    k, t, v, k2 = table(pod)
    want = orc.run_job(algo, k, t, v, key_id2=k2, agg_flow="pod" if pod else "svc")
    cat = {f: np.concatenate([g[1][f] for g in sorted(got, key=lambda g: g[0])]) for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
    order = np.lexsort((cat["flow_end_s"], cat["key_id"]))
    for f in cat:
        assert np.array_equal(cat[f][order], want[f], equal_nan=True), f   # the union of the shards' rows IS the single-GPU result
    x = orc.u64_to_f64(want["points"][2])
    for _, _, glob in got:
        assert glob["n_anomalies"] == want["n_anomalies"] and glob["n_keys"] == want["n_keys"] and glob["n_points"] == want["n_points"]
        assert abs(glob["global_mean"] - x.mean()) / x.mean() < 1e-12
        assert abs(glob["global_sigma"] - x.std(ddof=1)) / x.std(ddof=1) < 1e-12
        assert glob == got[0][2] or {k_: v_ for k_, v_ in glob.items() if k_ != "write_sentinel"} == {k_: v_ for k_, v_ in got[0][2].items() if k_ != "write_sentinel"}
    assert sum(1 for g in got if g[2]["write_sentinel"]) == (1 if want["n_anomalies"] == 0 else 0)


def torch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, t, v, _ = table(False)
        sl = slice(rank, None, world)                      # this rank's arbitrary slice of the rows
        pk, pt, pv = orc.stage0(k[sl], t[sl], v[sl], "sum")  # pre-aggregated partial points (tad_aggregate on the GPU)
        lk, lt, lv = td.exchange_points_torch(torch.from_numpy(pk.astype(np.int64)), torch.from_numpy(pt),
                                              torch.from_numpy(pv.view(np.int64)), world, rank)
        res = oracle_run("EWMA", lk.numpy().view(np.uint64), lt.numpy(), lv.numpy().view(np.uint64), td.num_local_keys(37, rank, world),
                         agg_flow="svc")
        rows = dict(res.rows)
        rows["key_id"] = td.global_key(rows["key_id"], rank, world)
        q.put((rank, rows))
    finally:
        dist.destroy_process_group()


def test_preaggregated_points_exchange_on_torch_tensors():
    # row-sharded ingest, pre-aggregated: partial sums travel, owners re-aggregate (sum of sums) -> the single-GPU result
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=torch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    k, t, v, _ = table(False)
    want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
    cat = {f: np.concatenate([g[1][f] for g in got]) for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
    order = np.lexsort((cat["flow_end_s"], cat["key_id"]))
    for f in cat:
        assert (cat[f][order] == want[f]).all(), f


def test_shard_helpers():
    k = np.array([0, 1, 2, 3, 4, 5, td.SKIP], dtype=np.uint64)
    assert td.owner_of(k[:6], 2).tolist() == [0, 1, 0, 1, 0, 1]
    for w in (1, 2, 3, 8):
        for r in range(w):
            mine = k[:6][td.owner_of(k[:6], w) == r]
            assert (td.global_key(td.local_key(mine, w), r, w) == mine).all()
            assert td.num_local_keys(6, r, w) == mine.size
    s = td.shard_rows(1, 2, k, np.arange(7), np.arange(7) * 10, key_id2=k[::-1].copy())
    # rank 1 owns odd keys: a row is kept if either of its keys is odd; the other key is masked
    assert ((s["key_id"] != td.SKIP) | (s["key_id2"] != td.SKIP)).all()
    assert td.chan_merge([(0, 0, 0), (2, 1.5, 0.5), (1, 4.0, 0.0)])[0] == 3
    n, mean, m2 = td.chan_merge([(2, 1.5, 0.5), (1, 4.0, 0.0)])
    x = np.array([1.0, 2.0, 4.0])
    assert abs(mean - x.mean()) < 1e-15 and abs(m2 - ((x - x.mean()) ** 2).sum()) < 1e-12


def test_sentinel_decision_single_rank():
    red = td.JobReducer()
    out = red.reduce({"n_anomalies": 0, "n_keys": 3, "n_points": 10, "pts_mean": 2.0, "pts_m2": 9.0})
    assert out["write_sentinel"] and out["global_sigma"] == 1.0


def pipelined_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        red = td.JobReducer()
        outs, pending = [], None
        for job in range(5):        # job j on rank r reports counters that identify (j, r); collected one job late
            stats = {"n_anomalies": 10 * job + rank, "n_keys": 100 + job, "n_points": 1000 * (rank + 1), "rows_used": job,
                     "keys_no_result": 0, "rows_in": 7, "pts_mean": float(job + rank), "pts_m2": float(job)}
            nxt = red.start(stats)
            if pending is not None:
                outs.append(pending.result())
            pending = nxt
        outs.append(pending.result())
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_pipelined_reducer_keeps_jobs_apart():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for job in range(5):
        for r in range(world):
            o = got[r][job]
            assert o["n_anomalies"] == sum(10 * job + x for x in range(world)) and o["n_keys"] == world * (100 + job)
            assert o["n_points"] == sum(1000 * (x + 1) for x in range(world)) and o["rows_used"] == world * job
            n, mean, m2 = td.chan_merge([(1000.0 * (x + 1), float(job + x), float(job)) for x in range(world)])
            assert o["global_mean"] == mean and o["global_sigma"] == (m2 / (n - 1.0)) ** 0.5
        assert {k: v for k, v in got[0][job].items() if k != "write_sentinel"} == {k: v for k, v in got[1][job].items() if k != "write_sentinel"}
