"""GPU: the placement calibration of pass B's record buffer is visible and controllable (ABI 11: tad_stats.placement_*, tad_plan.placement).

Pass B's rate depends on where its record buffer lies relative to the job's columns in physical memory: the first big job
of an engine times the buffer against further allocations and keeps the fastest.  Whatever it keeps, the RESULT is the same — checked here against the
oracle with and without the calibration."""
import numpy as np
import pytest

from oracle import tad_oracle as orc
from theia_amd import TadEngine

pytestmark = pytest.mark.gpu

N, K, T = 20_000_000, 20_000, 120       # >= 2^24 rows: the calibration's threshold


@pytest.fixture(scope="module")
def table():
    return orc.synth_rows_parallel(N, K, T)


def rows_of(res):
    return {f: np.array(res[f]) for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}


def test_first_big_job_reports_its_calibration_and_later_jobs_do_none(table):
    k, t, v = table
    eng = TadEngine(device=0)
    try:
        dk, dt, dv = k, t, v          # host columns: staged into the engine's own device buffers, whose addresses are stable from job to job
        r1 = eng.run("EWMA", dk, dt, dv, K, agg_flow="svc")
        s1 = r1.stats
        assert s1["stage0_path"] in (2, 3)
        assert s1["placement_candidates"] >= 1 and s1["placement_ms"] > 0.0
        assert 0.0 < s1["placement_kept_ms"] <= s1["placement_worst_ms"]
        r2 = eng.run("EWMA", dk, dt, dv, K, agg_flow="svc")
        s2 = r2.stats
        assert s2["placement_candidates"] == 0 and s2["placement_ms"] == 0.0          # the buffer is in place
        a, b = rows_of(r1), rows_of(r2)
        assert all((a[f] == b[f]).all() for f in a)
        want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
        assert r1.n_rows == want["n_anomalies"] and all((a[f] == want[f]).all() for f in a)
    finally:
        eng.close()


def test_placement_can_be_turned_off(table):
    k, t, v = table
    eng = TadEngine(device=0, plan={"placement": "never"})
    try:
        r = eng.run("EWMA", k, t, v, K, agg_flow="svc")
        assert r.stats["placement_candidates"] == 0 and r.stats["placement_ms"] == 0.0
        want = orc.run_job("EWMA", k, t, v, agg_flow="svc")
        got = rows_of(r)
        assert r.n_rows == want["n_anomalies"] and all((got[f] == want[f]).all() for f in got)
    finally:
        eng.close()
