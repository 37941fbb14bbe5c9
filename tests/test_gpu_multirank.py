"""GPU: the N>1 code path of bench.py on ONE GPU (ranks share the device, collectives over gloo on host tensors).

BASELINE.json configs[4] (C5: EWMA then ARIMA on a table split over the ranks) at a reduced size: the row-sharded
variant — every rank starts from an arbitrary slice of the rows, pre-aggregates it (tad_aggregate), buckets the partial
points by owner on the GPU (tad_shard_rows) and ships them with the all-to-all(v) — must produce, over 2 ranks, exactly
the anomaly rows of the 1-rank run on the whole table: every column of every row, bit for bit, for both detectors."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import tad_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(world, rows, keys, dump, extra=()):
    env = dict(os.environ, TAD_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    common = ["--config", "c5", "--rows", str(rows // world), "--keys", str(keys // world), "--buckets", "60", "--steps", "1", "--warmup", "0",
              "--no-cpu-baseline", "--dump-rows", dump] + list(extra)
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + common
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT).stdout
    return json.loads(out.strip().splitlines()[-1])


def merged_rows(dump, world, algo):
    parts = [np.load(dump + ".rank%d.npz" % r) for r in range(world)]
    cols = {f: np.concatenate([p["%s_%s" % (algo, f)] for p in parts]) for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
    order = np.lexsort((cols["flow_end_s"], cols["key_id"]))
    return {f: v[order] for f, v in cols.items()}


def test_c5_row_sharded_two_ranks_equal_one_rank(tmp_path):
    rows, keys = 600_000, 600
    one = run_bench(1, rows, keys, str(tmp_path / "w1"), ["--ingest", "rows"])
    two = run_bench(2, rows, keys, str(tmp_path / "w2"), ["--ingest", "rows"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["algo"] == "EWMA+ARIMA" and "row-sharded x2" in two["config"]["parallelism"]
    for f in ("anomalies", "keys", "points"):
        assert one["result"][f] == two["result"][f], f
    for algo in ("EWMA", "ARIMA"):
        a, b = merged_rows(str(tmp_path / "w1"), 1, algo), merged_rows(str(tmp_path / "w2"), 2, algo)
        assert a["key_id"].size == b["key_id"].size > 0
        for f in a:
            assert np.array_equal(a[f], b[f], equal_nan=True), (algo, f)
    # and the 1-rank EWMA rows are the oracle's rows on the whole table
    k, t, v = orc.synth_rows(0, rows, keys, 60)
    want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
    a = merged_rows(str(tmp_path / "w1"), 1, "EWMA")
    assert (a["key_id"] == want["key_id"]).all() and (a["flow_end_s"] == want["flow_end_s"]).all() and (a["algo_calc"] == want["algo_calc"]).all()


def test_shard_rows_buckets_by_owner(engine):
    from theia_amd.engine import DeviceArray
    k, t, v = orc.synth_rows(5, 300_000, 5000, 40)
    k[::97] = orc.KEY_SKIP
    # engine-owned device arrays (no torch in this process: the torch wheel carries its own HIP runtime, and which of two
    # runtimes in one process gets the device is not something a test should depend on; bench.py initialises torch first)
    tk, tt, tv = (DeviceArray.from_host(engine, x) for x in (k, t, v))
    for world in (1, 3, 8):
        (dk, dt, dv), counts = engine.shard_rows(tk, tt, tv, world)
        hk, ht, hv = dk.to_host(), dt.to_host(), dv.to_host()
        live = k != orc.KEY_SKIP
        assert sum(counts) == int(live.sum())
        pos = 0
        for d in range(world):
            sel = live & (k % np.uint64(world) == d)
            assert counts[d] == int(sel.sum())
            got = np.stack([hk[pos:pos + counts[d]].astype(np.int64), ht[pos:pos + counts[d]], hv[pos:pos + counts[d]].astype(np.int64)], axis=1)
            want = np.stack([(k[sel] // np.uint64(world)).astype(np.int64), t[sel], v[sel].astype(np.int64)], axis=1)
            assert (got[np.lexsort(got.T[::-1])] == want[np.lexsort(want.T[::-1])]).all(), (world, d)     # same multiset of rows
            pos += counts[d]
        for a in (dk, dt, dv):
            a.free()


def test_default_config_key_sharded_two_ranks(tmp_path):
    # the weak-scaling path under an EXTERNAL torchrun (`bench.py --gpus N`, key-sharded, one all-gather per job), reduced in size,
    # two ranks on one GPU: the line must describe the whole job (both ranks' rows), and the reduced counters must add up
    env = dict(os.environ, TAD_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "3000000", "--keys", "3000",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT).stdout
    d = json.loads(out.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["metric"] == "flow-records/sec"
    assert d["result"]["rows_used"] == 2 * 3_000_000 and d["result"]["keys"] == 2 * 3000
    assert abs(d["value"] - 2 * 3_000_000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9       # whole-job rate over both ranks
    assert "other_configs" not in d and "cpu_baseline" not in d and d["roofline"]["frac"] > 0


def _self_launched(extra):
    """`python bench.py --gpus 2 ...` with NO launcher around it and no WORLD_SIZE in the environment — the form the driver uses."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["TAD_BENCH_BACKEND"] = "gloo"          # two ranks share this box's one GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert sum(1 for l in lines if l.startswith("{")) == 1 and lines[-1].startswith("{"), r.stdout[-2000:]   # ONE line, last, rank 0's
    return json.loads(lines[-1])


def test_bench_gpus_2_starts_two_ranks_by_itself():
    d = _self_launched(["--rows", "3000000", "--keys", "3000", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["result"]["rows_used"] == 2 * 3_000_000 and d["result"]["keys"] == 2 * 3000
    assert abs(d["value"] - 2 * 3_000_000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    assert "other_configs" not in d


def test_bench_gpus_2_default_line_carries_c5_and_rccl_evidence():
    """What the driver's `bench.py --gpus N` must put on the line at N > 1 (reduced sizes here, two ranks on one GPU over gloo): the
    weak-scaled headline, BASELINE configs[4] (C5: EWMA then ARIMA, strong scaling) key-sharded AND row-sharded in other_configs, and
    the evidence that the collectives crossed ranks: ranks seen, all-gather wait, all-to-all bytes / time, per-rank step times."""
    d = _self_launched(["--rows", "2000000", "--keys", "2000", "--steps", "3", "--warmup", "1", "--c5-shape", "600000,600,60"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["tables"] == 4
    rc = d["rccl"]
    assert rc["world"] == 2 and rc["ranks_seen"] == 2 and rc["allgather_us_p50"] > 0 and rc["alltoall_bytes"] == 0
    assert 0 < rc["per_rank_ms_per_step"]["min"] <= rc["per_rank_ms_per_step"]["max"] == d["ms_per_step"]
    c5 = d["other_configs"]["c5"]
    for ing in ("keys", "rows"):
        c = c5[ing]
        assert "error" not in c, c
        assert c["config"]["algo"] == "EWMA+ARIMA" and c["result"]["keys"] == 600 and c["rccl"]["ranks_seen"] == 2
        assert c["scaling"].startswith("strong") and c["config"]["rows_per_gpu"] == 300000
    assert c5["rows"]["rccl"]["alltoall_bytes"] > 0 and c5["rows"]["rccl"]["alltoall_ms"] > 0 and c5["keys"]["rccl"]["alltoall_bytes"] == 0
    assert "row-sharded x2" in c5["rows"]["config"]["parallelism"]
    assert c5["keys"]["result"]["rows_used"] == 600000           # (row-sharded: the owners' jobs run on the shipped partial points)
    assert 0 < c5["rows"]["result"]["rows_used"] <= 600000 and c5["rows"]["rccl"]["alltoall_bytes"] % 24 == 0


def test_bench_c5_gpus_2_row_sharded_starts_two_ranks_by_itself():
    d = _self_launched(["--config", "c5", "--rows", "300000", "--keys", "300", "--buckets", "60", "--steps", "1", "--warmup", "0",
                        "--ingest", "rows"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["algo"] == "EWMA+ARIMA"
    assert "row-sharded x2" in d["config"]["parallelism"] and d["result"]["keys"] == 600


def test_rccl_one_rank_group_runs_the_collectives_on_device_tensors(tmp_path):
    """The only way a single-GPU box can execute this repository's RCCL calls before the driver's 8-GPU run does:
    `bench.py --force-collectives` creates a ONE-rank process group with backend "nccl" (= RCCL) and pushes the job's
    9-double all-gather, the all-to-all of the send counts and the three all-to-all(v) of the partial points through it, on
    device tensors (no TAD_BENCH_BACKEND=gloo here).  API / dtype / stream mistakes in theia_amd/distributed.py surface
    here; the rows must be those of the plain one-rank run, bit for bit."""
    rows, keys = 400_000, 400
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("TAD_BENCH_BACKEND", None)
    common = ["--config", "c5", "--rows", str(rows), "--keys", str(keys), "--buckets", "60", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = {}
    for name, extra in (("plain", ["--ingest", "rows"]), ("rccl", ["--ingest", "rows", "--force-collectives"]), ("rccl_keys", ["--force-collectives"])):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common + extra + ["--dump-rows", str(tmp_path / name)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        assert lines, (r.stdout[-2000:], r.stderr[-2000:])
        out[name] = json.loads(lines[-1])
    assert out["rccl"]["result"] == out["plain"]["result"] and out["rccl_keys"]["result"]["anomalies"] == out["plain"]["result"]["anomalies"]
    assert out["rccl"]["result"]["global_sigma"] is not None
    for algo in ("EWMA", "ARIMA"):
        a, b, c = (merged_rows(str(tmp_path / n), 1, algo) for n in ("plain", "rccl", "rccl_keys"))
        assert a["key_id"].size > 0
        for f in a:
            assert np.array_equal(a[f], b[f], equal_nan=True) and np.array_equal(a[f], c[f], equal_nan=True), (algo, f)
