"""theia_amd/controller.py — the in-process job runner against the reference controller's own tests:
the invalid-spec table of pkg/controller/anomalydetector/controller_test.go:318-441 (exact messages), the argument vector
of controller.go:525-623, the NEW -> SCHEDULED -> RUNNING -> COMPLETED / FAILED walk (controller.go:370-381), progress,
finishJob's EndTime and cleanupTADetector's DELETE statement (controller.go:385-398).  CPU tests inject the job body; the
`-m gpu` test runs the real engine between an in-process ClickHouse and the state machine."""
import threading
import uuid
from datetime import datetime, timedelta

import numpy as np
import pytest

from theia_amd import controller as ctl

NS = "flow-visibility"


def tad(name=None, **spec):
    return ctl.ThroughputAnomalyDetector(name=name or "tad-" + str(uuid.uuid4()), namespace=NS, spec=ctl.ThroughputAnomalyDetectorSpec(**spec))


# controller_test.go:318-441, one row per test case: (name, spec fields, expected message)
NOW = datetime(2022, 8, 11, 7, 26, 54)
INVALID = [
    ("tad-invalid-job-type", dict(jobType="nonexistent-job-type"),
     "invalid request: Throughput Anomaly Detector algorithm type should be 'EWMA' or 'ARIMA' or 'DBSCAN'"),
    ("tad-invalid-end-interval", dict(jobType="ARIMA", startInterval=NOW + timedelta(seconds=10), endInterval=NOW),
     "invalid request: EndInterval should be after StartInterval"),
    ("tad-invalid-executor-instances", dict(jobType="ARIMA", executorInstances=-1),
     "invalid request: ExecutorInstances should be an integer >= 0"),
    ("tad-invalid-driver-core-request", dict(jobType="ARIMA", executorInstances=1, driverCoreRequest="m200"),
     "invalid request: DriverCoreRequest should conform to the Kubernetes resource quantity convention"),
    ("tad-invalid-driver-memory", dict(jobType="ARIMA", executorInstances=1, driverCoreRequest="200m", driverMemory="m512"),
     "invalid request: DriverMemory should conform to the Kubernetes resource quantity convention"),
    ("tad-invalid-executor-core-request", dict(jobType="ARIMA", executorInstances=1, driverCoreRequest="200m", driverMemory="512M",
                                               executorCoreRequest="m200"),
     "invalid request: ExecutorCoreRequest should conform to the Kubernetes resource quantity convention"),
    ("tad-invalid-executor-memory", dict(jobType="ARIMA", executorInstances=1, driverCoreRequest="200m", driverMemory="512M",
                                         executorCoreRequest="200m", executorMemory="m512"),
     "invalid request: ExecutorMemory should conform to the Kubernetes resource quantity convention"),
    ("tad-invalid-agg-flow-pod-podNamespace-combo", dict(jobType="ARIMA", aggFlow="pod", podNameSpace="podNameSpace", podName="", podLabel=""),
     "invalid request: 'pod-namespace' argument can not be used alone"),
    ("tad-invalid-agg-flow", dict(jobType="ARIMA", aggFlow="nonexistent-agg-flow"),
     "invalid request: Throughput Anomaly Detector aggregated flow type should be 'pod' or 'external' or 'svc'"),
]


@pytest.fixture()
def controller():
    ran = []
    c = ctl.AnomalyDetectorController(run_job=lambda args, t: ran.append(args), progress=lambda: (3, 5))
    c.ran = ran
    yield c
    c.shutdown()


@pytest.mark.parametrize("name,spec,msg", INVALID, ids=[c[0] for c in INVALID])
def test_invalid_specs_fail_with_the_reference_messages(controller, name, spec, msg):
    controller.create(tad(name, **spec))
    got = controller.wait(NS, name, states=(ctl.STATE_FAILED,), timeout=10)
    assert got.status.state == ctl.STATE_FAILED
    assert msg in got.status.errorMsg and got.status.errorMsg.startswith("error in creating AnomalyDetector: ")   # controller.go:510
    assert controller.ran == [] and got.status.sparkApplication == ""


def test_name_must_be_tad_uuid(controller):
    controller.create(tad("tad-not-a-uuid", jobType="EWMA"))
    got = controller.wait(NS, "tad-not-a-uuid", states=(ctl.STATE_FAILED,), timeout=10)
    assert "invalid request: Throughput Anomaly Detector Querier job name is invalid" in got.status.errorMsg


def test_argument_vector_is_the_spark_applications():
    jid = str(uuid.uuid4())
    t = tad("tad-" + jid, jobType="DBSCAN", startInterval=datetime(2022, 8, 11, 7, 0, 0), endInterval=datetime(2022, 8, 11, 9, 30, 5),
            nsIgnoreList=["kube-system", "flow-visibility"], aggFlow="pod", podLabel="app:web", podNameSpace="default")
    assert ctl.job_arguments(t) == ["--algo", "DBSCAN", "--start_time", "2022-08-11 07:00:00", "--end_time", "2022-08-11 09:30:05",
                                    "--ns-ignore-list", '["kube-system","flow-visibility"]', "--agg-flow", "pod", "--pod-label", "app:web",
                                    "--pod-namespace", "default", "--id", jid]
    assert ctl.job_arguments(tad("tad-" + jid, jobType="EWMA", aggFlow="svc", servicePortName="p")) == \
        ["--algo", "EWMA", "--agg-flow", "svc", "--svc-port-name", "p", "--id", jid]
    assert ctl.job_arguments(tad("tad-" + jid, jobType="ARIMA", aggFlow="external", externalIp="10.0.0.1")) == \
        ["--algo", "ARIMA", "--agg-flow", "external", "--external-ip", "10.0.0.1", "--id", jid]
    # an EndInterval alone is fine: a zero StartInterval is before everything (controller.go:537-541)
    assert "--end_time" in ctl.job_arguments(tad("tad-" + jid, jobType="EWMA", endInterval=NOW))
    # the argument vector is what the job's own command line accepts (anomaly_detection.py:781-870)
    from theia_amd import anomaly_detection as ad
    assert ad.RESULT_TABLE_NAME.endswith(ctl.RESULT_TABLE)
    assert ctl.reference_cleanup_query(jid) == "ALTER TABLE tadetector ON CLUSTER '{cluster}' DELETE WHERE id = (" + jid + ");"   # controller.go:396, verbatim
    assert ctl.cleanup_query(jid) == "ALTER TABLE tadetector ON CLUSTER '{cluster}' DELETE WHERE id = ('" + jid + "');"   # what is sent: the id quoted


def test_states_progress_end_time_and_cleanup():
    gate, seen_states, commands = threading.Event(), [], []

    class FakeCH:
        def command(self, sql):
            commands.append(sql)

    def job(args, t):
        gate.wait(10)

    c = ctl.AnomalyDetectorController(clickhouse=FakeCH(), run_job=job, progress=lambda: (3, 5))
    try:
        t = tad(jobType="EWMA", aggFlow="svc")
        created = c.create(t)
        assert created.status.state in ("", ctl.STATE_NEW, ctl.STATE_SCHEDULED, ctl.STATE_RUNNING)
        running = c.wait(NS, t.name, states=(ctl.STATE_RUNNING,), timeout=10)
        assert running.status.state == ctl.STATE_RUNNING and running.status.sparkApplication == t.name[4:]
        assert running.status.startTime is not None and running.status.endTime is None
        # RUNNING resources get CompletedStages / TotalStages from the progress source (controller.go:426-453)
        for _ in range(200):
            running = c.get(NS, t.name)
            if running.status.totalStages:
                break
            threading.Event().wait(0.01)
        assert (running.status.completedStages, running.status.totalStages) == (3, 5)
        gate.set()
        done = c.wait(NS, t.name, timeout=10)
        assert done.status.state == ctl.STATE_COMPLETED and done.status.errorMsg == ""
        assert done.status.startTime < done.status.endTime                      # controller_test.go:306
        assert [x.name for x in c.list(NS)] == [t.name]                         # controller_test.go:308-311
        c.delete(NS, t.name)                                                    # -> cleanupTADetector
        assert commands == [ctl.cleanup_query(t.name[4:])]
        assert c.list(NS) == []
    finally:
        gate.set()
        c.shutdown()


def test_a_failing_job_ends_failed_with_the_reference_wording():
    def job(args, t):
        raise RuntimeError("tad error -6: dense point grid needs 9 bytes")

    c = ctl.AnomalyDetectorController(run_job=job, progress=lambda: (0, 4))
    try:
        t = tad(jobType="EWMA")
        c.create(t)
        got = c.wait(NS, t.name, states=(ctl.STATE_FAILED,), timeout=10)
        assert got.status.state == ctl.STATE_FAILED
        assert got.status.errorMsg == "Throughput Anomaly Detector job failed, state: FAILED, error message: tad error -6: dense point grid needs 9 bytes"
    finally:
        c.shutdown()


def test_four_workers_run_jobs_concurrently():
    """controller.go:199-201: 4 workers; four jobs are in flight at once (the engine serialises tad_run internally)."""
    barrier = threading.Barrier(4, timeout=10)
    c = ctl.AnomalyDetectorController(run_job=lambda a, t: barrier.wait(), progress=lambda: (0, 4))
    try:
        names = []
        for _ in range(4):
            t = tad(jobType="DBSCAN")
            names.append(t.name)
            c.create(t)
        for n in names:
            assert c.wait(NS, n, timeout=15).status.state == ctl.STATE_COMPLETED
    finally:
        c.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("algo,agg", [("EWMA", "svc"), ("DBSCAN", ""), ("ARIMA", "pod")])
def test_job_walks_the_states_on_the_gpu_engine(engine, algo, agg):
    """ClickHouse (in-process) -> controller -> tad_run on the MI355X -> rows in tadetector -> COMPLETED; the rows are the
    oracle's (oracle/job_oracle.py), then delete -> the DELETE statement of the reference."""
    from oracle import job_oracle as jo
    from theia_amd import anomaly_detection as ad
    from theia_amd import clickhouse as ch
    from test_clickhouse_http import FakeClickHouse, arrow_table
    server = FakeClickHouse()
    c = None
    try:
        flows = jo.synth_flows(6000)
        kw = dict(start_time="", end_time="", ns_ignore_list=[], agg_flow=agg, pod_label="app1" if agg == "pod" else "", external_ip="",
                  svc_port_name="", pod_name="", pod_namespace="")
        sql = ch.rows_query(kw["start_time"], kw["end_time"], kw["ns_ignore_list"], kw["agg_flow"], kw["pod_label"], kw["external_ip"],
                            kw["svc_port_name"], kw["pod_name"], kw["pod_namespace"])
        cols = sql[len("SELECT "):sql.index(" FROM ")].split(", ")
        server.responses[sql] = arrow_table({name: flows[name] for name in cols})
        client = ch.ClickHouseHTTP(server.url, user="", password="")
        c = ctl.AnomalyDetectorController(clickhouse=client, engine=engine)
        t = tad(jobType=algo, aggFlow=agg, podLabel=kw["pod_label"])
        c.create(t)
        done = c.wait(NS, t.name, timeout=120)
        assert done.status.state == ctl.STATE_COMPLETED, done.status.errorMsg
        assert done.status.totalStages == 4 and done.status.completedStages == 4 and done.status.startTime < done.status.endTime
        want = jo.run(flows, algo, tad_id=t.name[4:], **kw)
        got = [r for _, rows in server.inserted for r in rows]
        assert len(got) == len(want) and len(got) > 0
        # whole rows, every column of the tadetector schema, in (key columns, time) order: a key-decoding mistake cannot hide
        # behind equal multisets of numbers
        def canon(rows):
            out = []
            for r in rows:
                # (stddev_samp of a one-point key is null in Spark; the Float64 column of tadetector takes its default 0 for it, which
                # is what the engine emits: DESIGN.md section 1)
                d = {k: (0.0 if v is None else float(v)) if k in ("throughput", "algoCalc", "throughputStandardDeviation") else
                         (int(v) if k == "flowEndSeconds" else str(v)) for k, v in r.items()}
                out.append(d)
            return sorted(out, key=lambda d: tuple(str(d[k]) for k in sorted(d) if k not in ("throughput", "algoCalc", "throughputStandardDeviation")) +
                          (d["flowEndSeconds"],))
        cg, cw = canon(got), canon(want)
        assert [sorted(d) for d in cg] == [sorted(d) for d in cw]
        for a, b in zip(cg, cw):
            assert a == b, (a, b)
        assert all(r["id"] == t.name[4:] and r["algoType"] == algo for r in got)
        c.delete(NS, t.name)
        assert server.commands == [ctl.cleanup_query(t.name[4:])]
    finally:
        if c is not None:
            c.shutdown()
        server.close()


def test_delete_while_the_job_runs_writes_nothing_and_cleans_up_again():
    """cleanupTADetector stops the application before it can write (controller.go:385-398): a job body that is still running when
    its resource is deleted is cancelled — run_engine_job asks `cancelled()` before the INSERT — and the cleanup statement is
    issued once more when the body returns, so nothing it wrote can stay behind."""
    started, release, commands, wrote = threading.Event(), threading.Event(), [], []

    class FakeCH:
        def command(self, sql):
            commands.append(sql)

    c = ctl.AnomalyDetectorController(clickhouse=FakeCH())

    def job(args, t):          # what run_engine_job does around the engine call: check the tombstone, then write
        started.set()
        release.wait(10)
        if c._is_cancelled(t.name[4:]):
            raise ctl.JobCancelled(t.name[4:])
        wrote.append(t.name)

    c._run_job = job
    try:
        t = tad(jobType="EWMA", aggFlow="svc")
        c.create(t)
        assert started.wait(10)
        c.delete(NS, t.name)
        assert commands == [ctl.cleanup_query(t.name[4:])]
        release.set()
        for _ in range(500):
            if len(commands) == 2:
                break
            threading.Event().wait(0.01)
        assert commands == [ctl.cleanup_query(t.name[4:])] * 2 and wrote == []
        assert not c._is_cancelled(t.name[4:]) and c.list(NS) == []
    finally:
        c.shutdown()


def test_job_bodies_run_on_a_bounded_pool_and_cleanup_errors_do_not_escape():
    lock, running, peak, gate = threading.Lock(), [0], [0], threading.Event()

    class BrokenCH:
        def command(self, sql):
            raise RuntimeError("server says no")

    def job(args, t):
        with lock:
            running[0] += 1
            peak[0] = max(peak[0], running[0])
        gate.wait(10)
        with lock:
            running[0] -= 1

    c = ctl.AnomalyDetectorController(clickhouse=BrokenCH(), run_job=job, workers=2)
    try:
        names = []
        for _ in range(6):
            t = tad(jobType="EWMA")
            names.append(t.name)
            c.create(t)
        for _ in range(300):
            with lock:
                if running[0] == 2:
                    break
            threading.Event().wait(0.01)
        threading.Event().wait(0.1)
        assert peak[0] == 2                                   # never more bodies than workers
        gate.set()
        for n in names:
            assert c.wait(NS, n, timeout=15).status.state == ctl.STATE_COMPLETED
        c.delete(NS, names[0])                                # the DELETE fails on the server: logged, not raised (controller.go:262-280)
        assert isinstance(c._last_error, RuntimeError) and len(c.list(NS)) == 5
    finally:
        c.shutdown()


def test_a_progress_update_cannot_undo_a_completed_state():
    c = ctl.AnomalyDetectorController(run_job=lambda a, t: None, progress=lambda: (1, 4))
    try:
        t = tad(jobType="EWMA")
        c.create(t)
        done = c.wait(NS, t.name, timeout=10)
        assert done.status.state == ctl.STATE_COMPLETED
        key = (NS, t.name)
        c._update_status(key, only_if_state=(ctl.STATE_SCHEDULED, ctl.STATE_RUNNING), state=ctl.STATE_RUNNING, completedStages=1)
        assert c.get(NS, t.name).status.state == ctl.STATE_COMPLETED
        # a stale snapshot in RUNNING state synced late: update_progress must leave COMPLETED and endTime alone
        stale = c.get(NS, t.name)
        stale.status.state = ctl.STATE_RUNNING
        c.update_progress(key, stale)
        after = c.get(NS, t.name)
        assert after.status.state == ctl.STATE_COMPLETED and after.status.endTime == done.status.endTime
    finally:
        c.shutdown()


def test_start_job_for_a_deleted_key_starts_nothing_and_leaves_no_periodic_entry():
    """round-4 advisor finding: `_jobs[id]` was set before the status named the application — a delete in that window found no id to
    cancel, start_job then switched the periodic resync on for a dead key and ran the body.  Now: one critical section, and a key
    that has left the store starts nothing."""
    ran = []
    c = ctl.AnomalyDetectorController(run_job=lambda a, t: ran.append(t.name))
    try:
        t = tad(jobType="EWMA")
        key = (NS, t.name)
        c.start_job(key, t)                                   # never created (= created and deleted before its first sync ran)
        threading.Event().wait(0.1)
        assert ran == [] and key not in c._periodic and c._jobs == {} and c._alive == {}
    finally:
        c.shutdown()


def test_a_resource_recreated_under_the_same_name_waits_for_the_old_body_and_keeps_its_rows():
    """delete + create of the same name while the old body still runs: the new run has its own token (no inherited tombstone) and
    does not start before the old body — whose cleanup statement deletes by id — has returned."""
    release, commands, wrote, bodies = threading.Event(), [], [], []

    class FakeCH:
        def command(self, sql):
            commands.append((sql, len(wrote)))

    c = ctl.AnomalyDetectorController(clickhouse=FakeCH())

    def job(args, t):
        bodies.append(len(bodies))
        me = bodies[-1]
        if me == 0:
            release.wait(10)
        if c._is_cancelled(t.name[4:]):
            raise ctl.JobCancelled(t.name[4:])
        wrote.append(me)

    c._run_job = job
    try:
        t = tad(jobType="EWMA", aggFlow="svc")
        c.create(t)
        for _ in range(500):
            if bodies:
                break
            threading.Event().wait(0.01)
        c.delete(NS, t.name)
        c.create(tad(name=t.name, jobType="EWMA", aggFlow="svc"))    # same name, while body 0 is still blocked
        threading.Event().wait(0.3)
        assert bodies == [0] and wrote == []                  # the new run has not started
        release.set()
        done = c.wait(NS, t.name, timeout=10)
        assert done.status.state == ctl.STATE_COMPLETED
        assert bodies == [0, 1] and wrote == [1]              # body 0 was cancelled, body 1 wrote
        # both cleanup statements (the delete's and the cancelled body's) were issued BEFORE the new run wrote anything
        assert [n for _, n in commands] == [0, 0] and all(sql == ctl.cleanup_query(t.name[4:]) for sql, _ in commands)
        assert not c._is_cancelled(t.name[4:]) and c._alive == {}
    finally:
        c.shutdown()


def test_a_failing_sync_is_retried_with_exponential_backoff_and_forgotten_on_success():
    """The reference's queue is workqueue.NewItemExponentialFailureRateLimiter(MinRetryDelay, MaxRetryDelay) (controller.go:95,
    util.go:40-41: 5 s, 300 s); round 4 retried every 50 ms forever."""
    assert (ctl.MIN_RETRY_DELAY, ctl.MAX_RETRY_DELAY) == (5.0, 300.0)
    import time
    c = ctl.AnomalyDetectorController(run_job=lambda a, t: None, retry_min_delay=0.02, retry_max_delay=0.16)
    calls, fail = [], [True]
    real_sync = c.sync

    def sync(key):
        calls.append(time.monotonic())
        if fail[0]:
            raise RuntimeError("engine unavailable")
        return real_sync(key)

    c.sync = sync
    try:
        t = tad(jobType="EWMA")
        c.create(t)
        for _ in range(1000):                                 # (timers never fire early, so only LOWER bounds are asserted: a loaded box cannot fail this)
            if len(calls) >= 7:
                break
            threading.Event().wait(0.01)
        gaps = [b - a for a, b in zip(calls, calls[1:])][:6]
        # 0.02, 0.04, 0.08, 0.16, 0.16, ...: seven attempts take >= 0.62 s, where the fixed 50 ms retry of round 4 made them in 0.3 s
        assert len(gaps) == 6, len(calls)
        for g, nominal in zip(gaps, (0.02, 0.04, 0.08, 0.16, 0.16, 0.16)):
            assert g >= 0.75 * nominal, gaps
        assert calls[6] - calls[0] >= 0.55, gaps
        fail[0] = False
        assert c.wait(NS, t.name, timeout=10).status.state == ctl.STATE_COMPLETED
        assert c._failures == {}                              # Forget
    finally:
        c.shutdown()
