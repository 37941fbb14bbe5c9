"""In-process job runner for Throughput Anomaly Detection — the host half of SURVEY.md §8f rank 2.

The reference's AnomalyDetectorController (pkg/controller/anomalydetector/controller.go) turns a ThroughputAnomalyDetector
custom resource into a SparkApplication and then watches it.  With the MI355X engine there is no Spark application to
launch: `startJob` validates the spec exactly as `startSparkApplication` does (controller.go:525-623 — same checks in the
same order, same messages, the same argument vector), and then runs the job in this process: ClickHouse read ->
`tad_run` on the engine -> insert into `tadetector` (anomaly_detection.py:647-726), with the resource walking the same
states the CRD clients see today:

    "" / NEW -> SCHEDULED -> RUNNING -> COMPLETED | FAILED          (types.go:33-37, controller.go:370-381)

`Status.SparkApplication` keeps holding the bare job uuid (controller.go:622, 694): it is the `id` column of the result
rows, what the REST handler queries by (rest.go:143) and what cleanup deletes by (controller.go:385-398).  Progress comes
from `tad_progress` (the engine's four stages) instead of the Spark monitoring service (controller.go:426-453).

Go is not available in this image, so this is the Python statement of that controller logic; go/tadengine/tadengine.go is
the cgo binding a Go host would put under the same logic.  No Kubernetes client here: the resources live
in this object's store, `create/get/list/delete` mirror the REST verbs of pkg/apiserver/registry/intelligence/
throughputanomalydetector/rest.go.
"""
import copy
import queue
import re
import threading
import time
import uuid
from dataclasses import dataclass, field
from datetime import datetime, timezone
from typing import Callable, Dict, List, Optional

# ThroughputAnomalyDetector states (pkg/apis/crd/v1alpha1/types.go:33-37)
STATE_NEW = "NEW"
STATE_SCHEDULED = "SCHEDULED"
STATE_RUNNING = "RUNNING"
STATE_COMPLETED = "COMPLETED"
STATE_FAILED = "FAILED"

INPUT_TIME_FORMAT = "%Y-%m-%d %H:%M:%S"                                   # pkg/controller/util.go:45 "2006-01-02 15:04:05"
K8S_QUANTITIES_REG = r"^([+-]?[0-9.]+)([eEinumkKMGTP]*[-+]?[0-9]*)$"      # pkg/controller/util.go:46
DEFAULT_WORKERS = 4                                                       # pkg/controller/util.go:43, controller.go:199-201
RESULT_TABLE = "tadetector"


@dataclass
class ThroughputAnomalyDetectorSpec:
    """pkg/apis/crd/v1alpha1/types.go:96-112 (times: datetime or None for the zero metav1.Time)."""
    jobType: str = ""
    startInterval: Optional[datetime] = None
    endInterval: Optional[datetime] = None
    nsIgnoreList: List[str] = field(default_factory=list)
    aggFlow: str = ""
    podLabel: str = ""
    podName: str = ""
    podNameSpace: str = ""
    externalIp: str = ""
    servicePortName: str = ""
    executorInstances: int = 1
    driverCoreRequest: str = "200m"
    driverMemory: str = "512M"
    executorCoreRequest: str = "200m"
    executorMemory: str = "512M"


@dataclass
class ThroughputAnomalyDetectorStatus:
    """pkg/apis/crd/v1alpha1/types.go:114-122."""
    state: str = ""
    sparkApplication: str = ""     # the job uuid (name kept: REST, CLI and cleanup read this field)
    completedStages: int = 0
    totalStages: int = 0
    errorMsg: str = ""
    startTime: Optional[datetime] = None
    endTime: Optional[datetime] = None


@dataclass
class ThroughputAnomalyDetector:
    name: str
    namespace: str = "flow-visibility"
    spec: ThroughputAnomalyDetectorSpec = field(default_factory=ThroughputAnomalyDetectorSpec)
    status: ThroughputAnomalyDetectorStatus = field(default_factory=ThroughputAnomalyDetectorStatus)


class IllegalArgumentError(ValueError):
    """illeagelArguementError (controller.go:77-79): the resource is marked FAILED and not retried (controller.go:505-514)."""


def parse_ad_algorithm_id(tad_name):
    """pkg/util/utils.go:48-58: 'tad-<uuid>'."""
    if not tad_name.startswith("tad-"):
        raise ValueError("input name %s is not a valid Throughput Anomaly Detection job name" % tad_name)
    try:
        uuid.UUID(tad_name[4:])
    except ValueError:
        raise ValueError("input name %s does not contain a valid UUID" % tad_name)


def _is_zero(t):
    return t is None


def job_arguments(tad):
    """startSparkApplication's validation and argument vector (controller.go:525-623), check for check; raises
    IllegalArgumentError with the reference's messages (pinned by controller_test.go:326-441).  Returns the `Arguments` list
    the SparkApplication would have carried — what plugins/anomaly-detection/anomaly_detection.py:729-900 parses."""
    spec = tad.spec
    args = []
    if spec.jobType not in ("EWMA", "ARIMA", "DBSCAN"):
        raise IllegalArgumentError("invalid request: Throughput Anomaly Detector algorithm type should be 'EWMA' or 'ARIMA' or 'DBSCAN'")
    args += ["--algo", spec.jobType]
    if not _is_zero(spec.startInterval):
        args += ["--start_time", spec.startInterval.strftime(INPUT_TIME_FORMAT)]
    if not _is_zero(spec.endInterval):
        # EndInterval.After(StartInterval): a zero StartInterval is the year 1, so any EndInterval is after it
        if not _is_zero(spec.startInterval) and not spec.endInterval > spec.startInterval:
            raise IllegalArgumentError("invalid request: EndInterval should be after StartInterval")
        args += ["--end_time", spec.endInterval.strftime(INPUT_TIME_FORMAT)]
    if len(spec.nsIgnoreList) > 0:
        args += ["--ns-ignore-list", '["' + '","'.join(spec.nsIgnoreList) + '"]']
    if spec.aggFlow != "":
        if spec.aggFlow == "pod":
            args += ["--agg-flow", spec.aggFlow]
            if spec.podLabel != "":
                args += ["--pod-label", spec.podLabel]
            if spec.podName != "":
                args += ["--pod-name", spec.podName]
            if spec.podNameSpace != "":
                if spec.podName == "" and spec.podLabel == "":
                    raise IllegalArgumentError("invalid request: 'pod-namespace' argument can not be used alone, should be specified along pod-label or pod-name")
                args += ["--pod-namespace", spec.podNameSpace]
        elif spec.aggFlow == "external":
            args += ["--agg-flow", spec.aggFlow]
            if spec.externalIp != "":
                args += ["--external-ip", spec.externalIp]
        elif spec.aggFlow == "svc":
            args += ["--agg-flow", spec.aggFlow]
            if spec.servicePortName != "":
                args += ["--svc-port-name", spec.servicePortName]
        else:
            raise IllegalArgumentError("invalid request: Throughput Anomaly Detector aggregated flow type should be 'pod' or 'external' or 'svc'")
    # The Spark sizing fields have no meaning for the GPU engine; they stay validated so that a spec the reference rejects is
    # rejected here with the same message (the CRD and the CLI still carry them).
    if spec.executorInstances < 0:
        raise IllegalArgumentError("invalid request: ExecutorInstances should be an integer >= 0")
    for label, value in (("DriverCoreRequest", spec.driverCoreRequest), ("DriverMemory", spec.driverMemory),
                         ("ExecutorCoreRequest", spec.executorCoreRequest), ("ExecutorMemory", spec.executorMemory)):
        if not re.match(K8S_QUANTITIES_REG, value):
            raise IllegalArgumentError("invalid request: %s should conform to the Kubernetes resource quantity convention" % label)
    try:
        parse_ad_algorithm_id(tad.name)
    except ValueError as exc:
        raise IllegalArgumentError("invalid request: Throughput Anomaly Detector Querier job name is invalid: %s" % exc)
    args += ["--id", tad.name[4:]]
    return args


def reference_cleanup_query(job_id):
    """cleanupTADetector's statement (controller.go:396), verbatim — kept for the record: the uuid is NOT quoted there, which a
    ClickHouse server parses as an arithmetic expression over identifiers (and rejects, or misreads)."""
    return "ALTER TABLE tadetector ON CLUSTER '{cluster}' DELETE WHERE id = (" + job_id + ");"


def cleanup_query(job_id):
    """The statement this controller sends: the reference's (controller.go:396) with the id as a string literal.  `job_id` has
    passed parse_ad_algorithm_id (a canonical uuid: hex digits and dashes), so quoting is all the escaping it needs."""
    uuid.UUID(job_id)
    return "ALTER TABLE tadetector ON CLUSTER '{cluster}' DELETE WHERE id = ('" + job_id + "');"


MIN_RETRY_DELAY, MAX_RETRY_DELAY = 5.0, 300.0      # seconds: controllerutil.MinRetryDelay / MaxRetryDelay (pkg/controller/util.go:40-41)


class JobCancelled(Exception):
    """The resource was deleted while its job was running (DeleteSparkApplication, controller.go:387): nothing is written."""


def run_engine_job(args, client, engine=None, pushdown=False, cancelled=None, connections=0):
    """The job body the SparkApplication ran (anomaly_detection.py:647-726) on the GPU engine: parse the argument vector, read
    through `client` (theia_amd.clickhouse.ClickHouseHTTP), detect, append the rows to `tadetector`.  Returns the row count.
    `cancelled()` is asked right before the rows are written: a deleted job must not leave rows nobody tracks."""
    from . import anomaly_detection as ad
    opt = {}
    it = iter(args)
    for name in it:
        opt[name] = next(it)
    import json
    tad_id = opt["--id"]
    _, cols = ad.anomaly_detection(opt["--algo"], client, opt.get("--start_time", ""), opt.get("--end_time", ""), tad_id,
                                   json.loads(opt["--ns-ignore-list"]) if "--ns-ignore-list" in opt else [], opt.get("--agg-flow", ""),
                                   opt.get("--pod-label", ""), opt.get("--external-ip", ""), opt.get("--svc-port-name", ""),
                                   opt.get("--pod-name", ""), opt.get("--pod-namespace", ""), engine=engine, pushdown=pushdown, columnar=True,
                                   connections=connections)
    if cancelled is not None and cancelled():
        raise JobCancelled(tad_id)
    return ad.store_result_columns(client, cols)


class AnomalyDetectorController:
    """Work queue + workers + per-resource periodic resync, like the reference controller (controller.go:150-260), for resources
    held in this object.  `run_job(args, tad)` does the work of the Spark application (default: run_engine_job on `engine` through
    `clickhouse`); `progress()` returns (completed, total) stages of the job that is running (default: engine.progress)."""

    def __init__(self, clickhouse=None, engine=None, run_job: Optional[Callable] = None, progress: Optional[Callable] = None,
                 workers=DEFAULT_WORKERS, resync_period=0.05, pushdown=False, retry_min_delay=MIN_RETRY_DELAY, retry_max_delay=MAX_RETRY_DELAY,
                 connections=0):
        self.clickhouse = clickhouse
        self.engine = engine
        # connections > 0: the device ingest (that many parallel dictionary-encoded reads straight into HBM, theia_amd.clickhouse.fetch_flows_device)
        self._run_job = run_job or (lambda args, tad: run_engine_job(args, self.clickhouse, self.engine, pushdown,
                                                                      cancelled=lambda: self._is_cancelled(tad.name[4:]), connections=connections))
        self._tls = threading.local()             # .run = the run token of the job body this pool thread is executing
        self._progress = progress or self._engine_progress
        self._lock = threading.Lock()
        self._store: Dict[tuple, ThroughputAnomalyDetector] = {}
        # job id -> the RUN TOKEN of its application: {"id", "state": SUBMITTED|RUNNING|COMPLETED|FAILED, "error", "cancelled"}.  The
        # tombstone of a resource deleted while its body runs is the token's own "cancelled" flag — per run, not per id: a resource
        # re-created under the same name gets a new token, so it neither inherits the old tombstone nor has its rows removed by the
        # old body's cleanup (round-4 advisor finding)
        self._jobs: Dict[str, dict] = {}
        self._alive: Dict[str, list] = {}         # job id -> tokens whose body has been submitted and has not returned yet
        self._failures: Dict[tuple, int] = {}     # key -> consecutive failed syncs (the rate limiter's per-item count; Forget = pop)
        self._queue: "queue.Queue" = queue.Queue()
        self._queued: set = set()                 # keys waiting in the queue (the workqueue's dedup, controller.go:150-160)
        self._active: set = set()                 # keys a worker is processing: one key is never synced by two workers at once
        self._dirty: set = set()                  # keys re-added while active: requeued when the worker is done (workqueue semantics)
        self._periodic: Dict[tuple, bool] = {}
        self._stop = threading.Event()
        self._resync = resync_period
        self._retry_min, self._retry_max = retry_min_delay, retry_max_delay
        self._last_error = None
        # job bodies run on a bounded pool: `workers` concurrent ClickHouse reads / engine jobs at most (controller.go:199-201's
        # defaultWorkers bound what the reference starts at once; Spark bounded the rest)
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="tad-job")
        self._threads = [threading.Thread(target=self._worker, daemon=True) for _ in range(workers)]
        self._threads.append(threading.Thread(target=self._resync_loop, daemon=True))
        for t in self._threads:
            t.start()

    # ---- REST verbs (rest.go:95-140, 249-315) ----
    def create(self, tad):
        with self._lock:
            key = (tad.namespace, tad.name)
            if key in self._store:
                raise KeyError("ThroughputAnomalyDetector %s/%s already exists" % key)
            self._store[key] = copy.deepcopy(tad)
        self._enqueue(key)
        return self.get(tad.namespace, tad.name)

    def get(self, namespace, name):
        with self._lock:
            return copy.deepcopy(self._store[(namespace, name)])

    def list(self, namespace):
        with self._lock:
            return [copy.deepcopy(t) for (ns, _), t in sorted(self._store.items()) if ns == namespace]

    def delete(self, namespace, name):
        """DeleteThroughputAnomalyDetector + the delete handler's cleanup (controller.go:385-398): the application is stopped
        (a running job body is marked cancelled and writes nothing), then the result rows of the job go."""
        with self._lock:
            tad = self._store.pop((namespace, name))
            self._periodic.pop((namespace, name), None)
            job_id = tad.status.sparkApplication
            self._failures.pop((namespace, name), None)
            job = self._jobs.get(job_id) if job_id else None
            if job is not None and job["state"] in ("SUBMITTED", "RUNNING"):
                job["cancelled"] = True              # _execute re-issues the cleanup when the body returns
        if job_id:
            self.cleanup(namespace, job_id)

    def cleanup(self, namespace, job_id):
        with self._lock:
            self._jobs.pop(job_id, None)           # DeleteSparkApplication
        if self.clickhouse is not None:
            try:
                self.clickhouse.command(cleanup_query(job_id))
            except Exception as exc:               # the reference's delete handler logs the error and carries on (controller.go:262-280)
                self._last_error = exc

    def handle_stale_resources(self, namespace="flow-visibility", add_resync=True, remove_stale_db_entries=True):
        """handleStaleResources (controller.go:232-276), what the reference's garbage-collection worker runs once at start-up
        (controller.go:188-192): SCHEDULED / RUNNING resources go back on the periodic resync list, and result rows whose job has no
        resource any more are deleted (controllerutil.HandleStaleDbEntries, pkg/controller/util.go:239-270: `SELECT DISTINCT id FROM
        tadetector`, then the cleanup statement for every id without a `tad-<id>` resource).  The third leg, stale SparkApplications,
        has nothing to act on here — a job body dies with the process that ran it.  Returns the list of errors (the reference requeues
        the key with the legs that failed; callers here may simply call again)."""
        errors = []
        if add_resync:
            with self._lock:
                for key, tad in self._store.items():
                    if key[0] == namespace and tad.status.state in (STATE_SCHEDULED, STATE_RUNNING):
                        self._periodic[key] = True
        if remove_stale_db_entries and self.clickhouse is not None:
            try:
                got = self.clickhouse.query_columns("SELECT DISTINCT id FROM %s" % RESULT_TABLE)
                ids = [str(v) for v in got.get("id", [])]
            except Exception as exc:
                return errors + ["failed to get %s ids from ClickHouse: %s" % (RESULT_TABLE, exc)]
            for job_id in ids:
                with self._lock:
                    exists = (namespace, "tad-" + job_id) in self._store
                if exists:
                    continue
                try:
                    self.clickhouse.command(cleanup_query(job_id))
                except Exception as exc:           # an id that is not a uuid, or a server error: reported, the other ids still go
                    errors.append("%s: %s" % (job_id, exc))
        return errors

    def _is_cancelled(self, job_id):
        """Asked by a job body right before it writes: has THIS run's resource been deleted?  (From any other thread: is a cancelled
        body of that id still running?)"""
        run = getattr(self._tls, "run", None)
        with self._lock:
            if run is not None and run["id"] == job_id:
                return run["cancelled"]
            return any(r["cancelled"] for r in self._alive.get(job_id, ()))

    def shutdown(self):
        self._stop.set()
        for _ in self._threads:
            self._queue.put(None)
        self._pool.shutdown(wait=False)

    # ---- the controller ----
    def _enqueue(self, key):
        """workqueue.Add: a key waits in the queue once; one that is being processed is marked dirty and re-added afterwards."""
        with self._lock:
            if key in self._queued:
                return
            if key in self._active:
                self._dirty.add(key)
                return
            self._queued.add(key)
        self._queue.put(key)

    def _worker(self):
        while not self._stop.is_set():
            key = self._queue.get()
            if key is None:
                return
            with self._lock:
                self._queued.discard(key)
                self._active.add(key)
            failed = False
            try:
                self.sync(key)
            except Exception as exc:                # the reference requeues with rate limiting (controller.go:330-345)
                self._last_error = exc
                failed = True
            with self._lock:
                self._active.discard(key)
                dirty = key in self._dirty
                self._dirty.discard(key)
                retry = failed and key in self._store
                if retry:
                    self._failures[key] = n_failed = self._failures.get(key, 0) + 1
                else:
                    self._failures.pop(key, None)            # Forget (controller.go:339)
            if retry:
                # AddRateLimited: ItemExponentialFailureRateLimiter(MinRetryDelay, MaxRetryDelay) (controller.go:95, util.go:40-41) —
                # base * 2^(failures - 1), capped; a timer re-adds the key, no worker sleeps on it
                delay = min(self._retry_min * 2 ** min(n_failed - 1, 30), self._retry_max)
                t = threading.Timer(delay, self._enqueue, args=(key,))
                t.daemon = True
                t.start()
            if dirty:       # client-go re-adds an item marked dirty during processing on Done(), independently of AddRateLimited
                self._enqueue(key)

    def _resync_loop(self):
        while not self._stop.wait(self._resync):
            with self._lock:
                keys = [k for k, on in self._periodic.items() if on]
            for k in keys:
                self._enqueue(k)

    def _update_status(self, key, only_if_state=None, **changes):
        """updateTADetectorStatus (controller.go:700-730): only the fields a caller names change; ErrorMsg is overwritten when given.
        `only_if_state`: the write happens only while the stored state is one of these (a progress update must not undo a
        COMPLETED / FAILED another sync has written since this one took its snapshot)."""
        with self._lock:
            tad = self._store.get(key)
            if tad is None or (only_if_state is not None and tad.status.state not in only_if_state):
                return
            for name, value in changes.items():
                setattr(tad.status, name, value)

    def sync(self, key):
        """syncTADetector (controller.go:355-383)."""
        with self._lock:
            tad = copy.deepcopy(self._store.get(key))
        if tad is None:                              # already deleted
            return
        state = tad.status.state
        if state in ("", STATE_NEW):
            self.start_job(key, tad)
        elif state == STATE_SCHEDULED:
            self.check_job_status(key, tad)
        elif state == STATE_RUNNING:
            self.update_progress(key, tad)
        elif state == STATE_COMPLETED:
            if tad.status.endTime is None:
                self.finish_job(key, tad)

    def start_job(self, key, tad):
        """startJob + startSparkApplication (controller.go:499-523, 525-698)."""
        try:
            args = job_arguments(tad)
        except IllegalArgumentError as exc:
            self._update_status(key, state=STATE_FAILED, errorMsg="error in creating AnomalyDetector: %s" % exc)
            return
        job_id = tad.name[4:]
        run = {"id": job_id, "state": "SUBMITTED", "error": "", "cancelled": False}
        # ONE critical section: the application exists, the status names it and the periodic resync is on — or none of it.  A delete
        # can only come before (the key has left the store: nothing is started, no periodic entry for a dead key) or after (it finds
        # the id in the status, sets the token's tombstone and cleans up).
        with self._lock:
            cur = self._store.get(key)
            if cur is None:
                return
            if self._alive.get(job_id):
                # the body of a deleted resource of the same name is still running: one body per id at a time — its cleanup
                # statement (DELETE WHERE id) must not meet this run's rows.  The resync tick brings the key back.
                self._periodic[key] = True
                return
            self._jobs[job_id] = run
            self._alive.setdefault(job_id, []).append(run)
            cur.status.state, cur.status.sparkApplication, cur.status.startTime = STATE_SCHEDULED, job_id, datetime.now(timezone.utc)
            self._periodic[key] = True               # addPeriodicSync
        self._pool.submit(self._execute, run, args, tad)

    def _execute(self, run, args, tad):
        job_id = run["id"]
        with self._lock:
            started = not run["cancelled"]           # deleted while it waited for a pool slot: never started
            if started:
                run["state"] = "RUNNING"
        outcome = None
        if started:
            self._tls.run = run
            try:
                self._run_job(args, tad)
                outcome = ("COMPLETED", "")
            except Exception as exc:                 # TadError, ClickHouse errors, ...: the application failed
                outcome = ("FAILED", str(exc))
            finally:
                self._tls.run = None
        with self._lock:
            if outcome is not None:
                run["state"], run["error"] = outcome
            deleted = started and run["cancelled"]
        try:
            if deleted and self.clickhouse is not None:
                # the resource went away while the body ran: whatever it managed to write before noticing is removed again — BEFORE the
                # run leaves _alive: start_job waits for that, so a resource re-created under the same name cannot insert rows that this
                # DELETE WHERE id would take with it (round-5 advisor finding)
                try:
                    self.clickhouse.command(cleanup_query(job_id))
                except Exception as exc:
                    self._last_error = exc
        finally:
            with self._lock:
                alive = self._alive.get(job_id, [])
                if run in alive:
                    alive.remove(run)
                if not alive:
                    self._alive.pop(job_id, None)

    def check_job_status(self, key, tad):
        """checkSparkApplicationStatus (controller.go:455-497)."""
        if tad.status.sparkApplication == "":
            self._update_status(key, state=STATE_FAILED, errorMsg="Spark Application should be started before status checking")
            return ""
        with self._lock:
            job = dict(self._jobs.get(tad.status.sparkApplication, {"state": "", "error": ""}))
        state, msg = job["state"], job["error"]
        if state == "RUNNING":
            self._update_status(key, only_if_state=("", STATE_NEW, STATE_SCHEDULED, STATE_RUNNING), state=STATE_RUNNING, errorMsg=msg)
        elif state == "COMPLETED":
            self._update_status(key, state=STATE_COMPLETED, errorMsg=msg)
        elif state in ("FAILED", "SUBMISSION_FAILED", "FAILING", "INVALIDATING"):
            self._update_status(key, state=STATE_FAILED,
                                errorMsg="Throughput Anomaly Detector job failed, state: %s, error message: %s" % (state, msg))
            with self._lock:
                self._periodic[key] = False
        return state

    def _engine_progress(self, job_id=None):
        """(completed, total) stages of job `job_id` — the reference asks the Spark monitoring service of THAT application
        (controller.go:426-453); several jobs may be in flight on the engine (tad.h ABI 12: job contexts), so the engine is asked by id
        (tad_job_progress); a job that is not in flight any more reports what the engine finished last (tad_progress)."""
        if self.engine is None:
            return (0, 0)
        if job_id and hasattr(self.engine, "job_progress"):
            done, total = self.engine.job_progress(job_id)
            if total:
                return done, total
        return self.engine.progress()

    def _progress_of(self, tad):
        try:
            return self._progress(tad.status.sparkApplication)
        except TypeError:                            # a caller-supplied progress() without arguments
            return self._progress()

    def update_progress(self, key, tad):
        """updateProgress (controller.go:426-453)."""
        state = self.check_job_status(key, tad)
        if state != "RUNNING":
            return
        try:
            done, total = self._progress_of(tad)
        except Exception:                            # the monitoring endpoint may not be up: not requeued (controller.go:437-443)
            return
        self._update_status(key, only_if_state=(STATE_SCHEDULED, STATE_RUNNING), state=STATE_RUNNING, completedStages=int(done),
                            totalStages=int(total))

    def finish_job(self, key, tad):
        """finishJob (controller.go:400-424)."""
        with self._lock:
            self._periodic[key] = False              # stopPeriodicSync
        if tad.status.sparkApplication == "":
            self._update_status(key, state=STATE_FAILED, errorMsg="Spark Application should be started before updating results")
            return
        try:
            done, total = self._progress_of(tad)
            self._update_status(key, completedStages=int(done), totalStages=int(total))
        except Exception:
            pass
        self._update_status(key, state=STATE_COMPLETED, endTime=datetime.now(timezone.utc))

    def wait(self, namespace, name, states=(STATE_COMPLETED, STATE_FAILED), timeout=30.0, need_end_time=True):
        """Poll like the reference's tests do (controller_test.go:285-301)."""
        deadline = time.time() + timeout
        while time.time() < deadline:
            tad = self.get(namespace, name)
            if tad.status.state in states and (tad.status.state != STATE_COMPLETED or not need_end_time or tad.status.endTime is not None):
                return tad
            time.sleep(0.01)
        return self.get(namespace, name)
