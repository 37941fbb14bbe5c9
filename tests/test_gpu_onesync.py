"""GPU: the one-synchronisation form of a job (ABI 8, tad_stats.host_syncs).  A job of the same shape as the engine's previous one is
issued with that job's lattice and result capacity while the device verifies both; results must be those of the plain form bit for
bit, and every way the speculation can miss (another time range, more rows than the block holds, a table that needs a retry anyway)
must fall back to the three-synchronisation form and still give the oracle's rows."""
import numpy as np
import pytest

from oracle import tad_oracle as orc
from theia_amd.engine import DeviceArray

pytestmark = pytest.mark.gpu
FIELDS = ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")


@pytest.fixture()
def v2(engine):
    engine.set_plan(stage0="v2", partition_pass="wc")      # small tables take the partition path too
    yield engine
    engine.set_plan()


def dev(engine, *cols):
    return [DeviceArray.from_host(engine, c) for c in cols]


def same_rows(res, want):
    assert res.n_rows == want["n_anomalies"]
    for f in FIELDS:
        assert (res[f] == want[f]).all(), f


@pytest.mark.parametrize("algo,agg", [("EWMA", "svc"), ("DBSCAN", "")])
def test_second_job_of_the_same_shape_has_one_host_synchronisation(v2, algo, agg):
    K, T = 700, 60
    k, t, v = orc.synth_rows(3, 200_000, K, T)
    want = orc.run_job(algo, k, t, v, agg_flow=agg)
    dk, dt, dv = dev(v2, k, t, v)
    first = v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="device")
    assert first.stats["host_syncs"] == 3 and first.stats["stage0_attempts"] == 1
    same_rows(first, want)
    for _ in range(3):
        again = v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="device")
        assert again.stats["host_syncs"] == 1 and again.stats["stage0_attempts"] == 1
        same_rows(again, want)
        for f in ("n_keys", "n_points", "rows_used", "t0", "step", "n_buckets", "pts_mean", "pts_m2"):
            assert again.stats[f] == first.stats[f], f
    # host-resident output, emit-all and a lattice hint are never speculated
    assert v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="host").stats["host_syncs"] == 3
    assert v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="device", emit_all=True).stats["host_syncs"] == 3
    # plan override: never
    v2.set_plan(stage0="v2", partition_pass="wc", one_sync="never")
    v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="device")
    assert v2.run(algo, dk, dt, dv, K, agg_flow=agg, out="device").stats["host_syncs"] == 3


def test_another_time_range_misses_and_is_redone(v2):
    K, T = 500, 40
    k, t, v = orc.synth_rows(5, 150_000, K, T)
    dk, dt, dv = dev(v2, k, t, v)
    v2.run("EWMA", dk, dt, dv, K, agg_flow="svc", out="device")
    # same rows and keys, every timestamp a day later / a coarser step / one bucket more: the remembered lattice is wrong each time
    for t2 in (t + 86400, t.min() + (t - t.min()) * 2, np.where(np.arange(t.size) == 7, t.max() + 60, t)):
        want = orc.run_job("EWMA", k, t2, v, agg_flow="svc")
        d2 = DeviceArray.from_host(v2, t2)
        res = v2.run("EWMA", dk, d2, dv, K, agg_flow="svc", out="device")
        assert res.stats["host_syncs"] == 3 and res.stats["stage0_attempts"] == 2      # the speculated attempt, then the plain one
        same_rows(res, want)
        again = v2.run("EWMA", dk, d2, dv, K, agg_flow="svc", out="device")              # ... which is remembered in turn
        assert again.stats["host_syncs"] == 1
        same_rows(again, want)


def test_more_rows_than_the_speculated_block_holds(v2):
    K, T = 2000, 60
    k, t, v = orc.synth_rows(9, 300_000, K, T)
    flat = np.full(v.size, 5_000_000, dtype=np.uint64)      # constant series: sigma = 0 and the EWMA (from 0) never equals x -> every point is a row
    dk, dt, dv, df = dev(v2, k, t, v, flat)
    few = v2.run("EWMA", dk, dt, dv, K, agg_flow="", out="device")
    want = orc.run_job("EWMA", k, t, flat, agg_flow="")
    assert want["n_anomalies"] > few.n_rows + few.n_rows // 8 + 4096    # does not fit the block sized from the previous job's rows
    res = v2.run("EWMA", dk, dt, df, K, agg_flow="", out="device")
    assert res.stats["host_syncs"] == 3 and res.stats["stage0_attempts"] == 2
    same_rows(res, want)
    again = v2.run("EWMA", dk, dt, df, K, agg_flow="", out="device")
    assert again.stats["host_syncs"] == 1
    same_rows(again, want)
    # fewer rows than remembered is no miss
    back = v2.run("EWMA", dk, dt, dv, K, agg_flow="", out="device")
    assert back.stats["host_syncs"] == 1
    same_rows(back, orc.run_job("EWMA", k, t, v, agg_flow=""))


def test_errors_still_surface_from_a_speculated_job(v2):
    from theia_amd import TadError
    K, T = 300, 30
    k, t, v = orc.synth_rows(2, 90_000, K, T)
    dk, dt, dv = dev(v2, k, t, v)
    v2.run("EWMA", dk, dt, dv, K, agg_flow="svc", out="device")
    bad = k.copy()
    bad[1234] = K + 5                                               # a key id out of range: same shape, so the job is speculated
    db = DeviceArray.from_host(v2, bad)
    with pytest.raises(TadError):
        v2.run("EWMA", db, dt, dv, K, agg_flow="svc", out="device")
    same_rows(v2.run("EWMA", dk, dt, dv, K, agg_flow="svc", out="device"), orc.run_job("EWMA", k, t, v, agg_flow="svc"))
