"""TadEngine — Python host over the C ABI (include/tad.h).  Plumbing only: every number is computed
by the HIP kernels in libtad_mi355x.so; there is no CPU fallback and no import of oracle/.

Columns may be numpy arrays (host memory), torch CUDA tensors or DeviceArray objects (device
memory).  torch is optional here: it is only touched when the caller hands in tensors.
"""
import ctypes as C

import numpy as np

from . import _capi as capi

SYNTH_SEED = 0x7AD05EED


class TadError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("tad error %d: %s" % (code, message))
        self.code = code
        self.message = message


class DeviceArray:
    """n 8-byte elements in HBM, owned by an engine (tad_device_alloc / tad_device_free)."""

    def __init__(self, engine, n, dtype):
        self.engine = engine
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        ptr = C.c_void_p()
        engine._check(engine._lib.tad_device_alloc(engine._h, self.n * self.dtype.itemsize, C.byref(ptr)))
        self.ptr = ptr.value

    @classmethod
    def from_host(cls, engine, array):
        """device copy of an 8-byte numpy array (tad_device_alloc + tad_copy_to_device)"""
        a = np.ascontiguousarray(array)
        d = cls(engine, a.size, a.dtype)
        if a.size:
            engine._check(engine._lib.tad_copy_to_device(engine._h, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            self.engine._check(self.engine._lib.tad_copy_to_host(self.engine._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.engine._h is not None:
            self.engine._lib.tad_device_free(self.engine._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Page-locked host memory owned by an engine (tad_host_alloc): a reader receives an HTTP body straight into `view`, and
    copies from it to the device run at PCIe rate.  Kept and reused by the ingest client between jobs (pinning is slow)."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = int(nbytes)
        ptr = C.c_void_p()
        engine._check(engine._lib.tad_host_alloc(engine._h, self.nbytes, C.byref(ptr)))
        self.ptr = ptr.value
        self.view = memoryview((C.c_ubyte * self.nbytes).from_address(self.ptr)).cast("B")

    def free(self):
        if self.ptr is not None and self.engine._h is not None:
            self.view = None
            self.engine._lib.tad_host_free(self.engine._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _as_column(x, dtype, n_expected=None):
    """-> (pointer, n, is_device, keepalive)"""
    if x is None:
        return None, 0, None, None
    if isinstance(x, DeviceArray):
        return x.ptr, x.n, True, x
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):  # torch tensor
        if not x.is_contiguous():
            x = x.contiguous()
        if x.element_size() != 8:
            raise TypeError("tensor columns must be 8-byte integers")
        if x.is_cuda:
            return x.data_ptr(), x.numel(), True, x
        x = x.numpy()
    a = np.ascontiguousarray(x)
    if a.dtype != np.dtype(dtype):
        if a.dtype.kind in "iu" and a.dtype.itemsize == 8:
            a = a.view(dtype)
        else:
            a = a.astype(dtype)
    return a.ctypes.data, a.size, False, a


class KeyHistogram:
    """tad_factorize_hist's by-product: the key-bin histogram of a batch per Stage-0 workgroup, in HBM.  Hand it to TadEngine.run(key_hist=...)
    with the SAME batch: Stage 0 then sizes pass B's regions from it instead of reading the key column a second time."""

    def __init__(self, engine):
        self.engine = engine
        self.bins = DeviceArray(engine, capi.TAD_KEY_HIST_BYTES // 8, np.uint64)
        self.c = capi.KeyHist(bins=self.bins.ptr)

    @property
    def valid(self):
        return self.c.n_rows != 0

    def free(self):
        self.bins.free()


class PreparedJob:
    """TadEngine.prepare(...): one job over one set of live columns, ready to be submitted any number of times."""

    def __init__(self, engine, job, cols, out_memory, keep):
        self._engine, self._job, self._cols, self._out, self._keep = engine, job, cols, out_memory, keep
        self._jref, self._cref = C.byref(job), C.byref(cols)

    def run(self):
        res = C.POINTER(capi.Result)()
        e = self._engine
        e._check(e._lib.tad_run(e._h, self._jref, self._cref, self._out, C.byref(res)))
        return TadResult(e, res)


class TadResult:
    """Anomalous points ordered by (key_id, flow_end_s) + the run's counters and stage timings."""

    FIELDS = (("key_id", np.uint64), ("flow_end_s", np.int64), ("throughput", np.float64),
              ("algo_calc", np.float64), ("stddev", np.float64))

    def __init__(self, engine, res_ptr):
        self._engine = engine
        self._ptr = res_ptr
        r = res_ptr.contents
        self.n_rows = int(r.n_rows)
        self.memory = "device" if r.memory == capi.TAD_MEM_DEVICE else "host"
        self.id = r.id.decode()
        self.stats = {name: getattr(r.stats, name) for name, _ in capi.Stats._fields_}
        self._host = None
        if self.memory == "host":
            self._host = self._copy_host(r, direct=True)
            self.close()

    def _copy_host(self, r, direct):
        out = {}
        n = self.n_rows
        cols = list(self.FIELDS) + ([("anomaly", np.uint8)] if r.anomaly else [])
        for name, dt in cols:
            arr = np.empty(n, dtype=dt)
            src = getattr(r, name)
            if n:
                if direct:
                    C.memmove(arr.ctypes.data, src, arr.nbytes)
                else:
                    self._engine._check(self._engine._lib.tad_copy_to_host(self._engine._h, arr.ctypes.data, src, arr.nbytes))
            out[name] = arr
        return out

    def device_pointers(self):
        if self._ptr is None or self.memory != "device":
            raise ValueError("no live device result")
        r = self._ptr.contents
        return {name: getattr(r, name) for name, _ in self.FIELDS + (("anomaly", np.uint8),)}

    def to_host(self):
        """dict of numpy arrays: key_id, flow_end_s, throughput, algo_calc, stddev (, anomaly)."""
        if self._host is None:
            self._host = self._copy_host(self._ptr.contents, direct=False)
        return self._host

    def __getitem__(self, name):
        return self.to_host()[name]

    def close(self):
        if self._ptr is not None and self._engine._h is not None:
            self._engine._lib.tad_result_free(self._engine._h, self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TadPoints:
    """Stage-0 output (tad_aggregate): aggregated points ordered by (key_id, flow_end_s), values as raw uint64."""

    FIELDS = (("key_id", np.uint64), ("flow_end_s", np.int64), ("value", np.uint64))

    def __init__(self, engine, ptr):
        self._engine = engine
        self._ptr = ptr
        p = ptr.contents
        self.n_points = int(p.n_points)
        self.memory = "device" if p.memory == capi.TAD_MEM_DEVICE else "host"
        self.stats = {name: getattr(p.stats, name) for name, _ in capi.Stats._fields_}
        self._host = None
        if self.memory == "host":
            self._host = self._copy(direct=True)
            self.close()

    def _copy(self, direct):
        p = self._ptr.contents
        out = {}
        for name, dt in self.FIELDS:
            arr = np.empty(self.n_points, dtype=dt)
            if self.n_points:
                if direct:
                    C.memmove(arr.ctypes.data, getattr(p, name), arr.nbytes)
                else:
                    self._engine._check(self._engine._lib.tad_copy_to_host(self._engine._h, arr.ctypes.data, getattr(p, name), arr.nbytes))
            out[name] = arr
        return out

    def device_pointers(self):
        if self._ptr is None or self.memory != "device":
            raise ValueError("no live device points")
        p = self._ptr.contents
        return {name: getattr(p, name) for name, _ in self.FIELDS}

    def to_host(self):
        if self._host is None:
            self._host = self._copy(direct=False)
        return self._host

    def __getitem__(self, name):
        return self.to_host()[name]

    def close(self):
        if self._ptr is not None and self._engine._h is not None:
            self._engine._lib.tad_points_free(self._engine._h, self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TadState:
    """Per-key running state of the streaming EWMA detector (tad_state), resident in HBM."""

    def __init__(self, engine, num_keys):
        self._engine = engine
        self.num_keys = int(num_keys)
        h = C.c_void_p()
        engine._check(engine._lib.tad_state_create(engine._h, self.num_keys, C.byref(h)))
        self._h = h

    def export(self):
        """dict of numpy arrays: n, avg, m2, ewma, last_t (one entry per key)."""
        K = self.num_keys
        out = {"n": np.zeros(K, np.uint32), "avg": np.zeros(K), "m2": np.zeros(K), "ewma": np.zeros(K), "last_t": np.zeros(K, np.int64)}
        self._engine._check(self._engine._lib.tad_state_export(self._engine._h, self._h, *(out[f].ctypes.data for f in
                                                                                             ("n", "avg", "m2", "ewma", "last_t"))))
        return out

    def close(self):
        if self._h is not None and self._engine._h is not None:
            self._engine._lib.tad_state_destroy(self._engine._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TadEngine:
    """One engine per GPU.  Thread-safe: up to max_jobs_in_flight jobs (0 = the library's default, 4) run concurrently, each on its own
    job context (HIP stream + workspace) inside the library; further callers wait."""

    def __init__(self, device=0, stream=None, workspace_limit=0, plan=None, library_path=None, max_jobs_in_flight=0):
        """plan: dict of tad_plan overrides (include/tad.h), e.g. {"stage0": "v2", "partition_pass": "sort"}; None = the
        engine decides everything (production).  library_path: another build of the library (A/B measurements, tools/ab_plans.py)."""
        self._lib = capi.load_library(path=library_path)
        self._h = None
        self.device = int(device)
        self._plan = dict(plan or {})
        opts = capi.EngineOpts(device=int(device), stream=C.c_void_p(stream) if stream else None,
                               workspace_limit=int(workspace_limit), plan=capi.make_plan(**self._plan),
                               max_jobs_in_flight=int(max_jobs_in_flight))
        h = C.c_void_p()
        rc = self._lib.tad_engine_create(C.byref(opts), C.byref(h))
        if rc != capi.TAD_OK:
            raise TadError(rc, (self._lib.tad_last_error(None) or b"").decode())
        self._h = h

    # ---- plumbing ----
    def _check(self, rc):
        if rc != capi.TAD_OK:
            raise TadError(rc, (self._lib.tad_last_error(self._h) or b"").decode())

    def set_plan(self, **overrides):
        """Replace the engine's plan overrides (tad_engine_set_plan); no arguments = back to automatic."""
        p = capi.make_plan(**overrides)
        self._check(self._lib.tad_engine_set_plan(self._h, C.byref(p)))
        self._plan = dict(overrides)

    def plan(self, **overrides):
        """Context manager: the jobs inside run with these overrides ON TOP of the current ones, which are restored after."""
        engine = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.saved = dict(engine._plan)
                merged = dict(engine._plan)
                merged.update(overrides)
                engine.set_plan(**merged)
                return engine

            def __exit__(self_inner, *exc):
                engine.set_plan(**self_inner.saved)
                return False
        return _Scope()

    def close(self):
        if self._h is not None:
            self._lib.tad_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def progress(self):
        d, t = capi.i32(), capi.i32()
        self._lib.tad_progress(self._h, C.byref(d), C.byref(t))
        return d.value, t.value

    def job_progress(self, job_id):
        """(done, total) of the job in flight whose id is job_id; (0, 0) when there is none"""
        d, t = capi.i32(), capi.i32()
        self._lib.tad_job_progress(self._h, job_id.encode()[:63], C.byref(d), C.byref(t))
        return d.value, t.value

    def jobs_in_flight(self):
        return int(self._lib.tad_jobs_in_flight(self._h))

    # ---- the job (anomaly_detection.py:647-710) ----
    def run(self, algo, key_id, flow_end_s, value, num_keys, agg_flow="", value_op="auto", key_id2=None,
            flow_start_s=None, start_time=0, end_time=0, lattice=None, emit_all=False, out="host", job_id="",
            alpha=0.0, eps=0.0, min_samples=0, maxiter=0, drop_nsigma=0.0, drop_min_samples=0, key_hist=None, _prepare_only=False):
        if algo not in capi.TAD_ALGO:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "algo must be EWMA, ARIMA, DBSCAN or DROP")
        if agg_flow not in capi.TAD_AGG:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "agg_flow must be '', pod, svc or external")
        pk, n, dev, keep1 = _as_column(key_id, np.uint64)
        pt, nt, dev_t, keep2 = _as_column(flow_end_s, np.int64)
        pv, nv, dev_v, keep3 = _as_column(value, np.uint64)
        pk2, nk2, dev_k2, keep4 = _as_column(key_id2, np.uint64)
        ps, ns, dev_s, keep5 = _as_column(flow_start_s, np.int64)
        for m, d in ((nt, dev_t), (nv, dev_v)) + (((nk2, dev_k2),) if key_id2 is not None else ()) + \
                (((ns, dev_s),) if flow_start_s is not None else ()):
            if m != n or d != dev:
                raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "columns must have equal length and live in the same memory")
        job = capi.Job(algo=capi.TAD_ALGO[algo], agg_flow=capi.TAD_AGG[agg_flow], value_op=capi.TAD_OP[value_op],
                       start_time=int(start_time), end_time=int(end_time), ewma_alpha=float(alpha),
                       dbscan_eps=float(eps), dbscan_min_samples=int(min_samples), arima_maxiter=int(maxiter),
                       drop_nsigma=float(drop_nsigma), drop_min_samples=int(drop_min_samples),
                       flags=capi.TAD_FLAG_EMIT_ALL_POINTS if emit_all else 0, id=job_id.encode()[:63])
        cols = capi.Columns(n_rows=n, key_id=pk, key_id2=pk2, flow_end_s=pt, flow_start_s=ps, value=pv,
                            num_keys=int(num_keys), memory=capi.TAD_MEM_DEVICE if dev else capi.TAD_MEM_HOST)
        if lattice is not None:
            cols.t0, cols.step, cols.n_buckets = int(lattice[0]), int(lattice[1]), int(lattice[2])
        if key_hist is not None and key_hist.valid:
            cols.key_hist = C.pointer(key_hist.c)
        if _prepare_only:
            return PreparedJob(self, job, cols, capi.TAD_MEM_DEVICE if out == "device" else capi.TAD_MEM_HOST,
                               (keep1, keep2, keep3, keep4, keep5, key_hist))
        res = C.POINTER(capi.Result)()
        rc = self._lib.tad_run(self._h, C.byref(job), C.byref(cols),
                               capi.TAD_MEM_DEVICE if out == "device" else capi.TAD_MEM_HOST, C.byref(res))
        del keep1, keep2, keep3, keep4, keep5
        self._check(rc)
        return TadResult(self, res)

    def prepare(self, *args, **kw):
        """Same arguments as run(): the tad_job / tad_columns structs built ONCE, for a host that submits the same job over the
        same (live) columns repeatedly — PreparedJob.run() is then the bare tad_run call (a cgo host pays no more either; building
        the two structs and inspecting five column objects in Python costs ~15 us per call, 1 % of a C2 job)."""
        return self.run(*args, _prepare_only=True, **kw)

    # ---- streaming EWMA: one new batch against the per-key running state ----
    def state_create(self, num_keys):
        return TadState(self, num_keys)

    def run_stream(self, state, key_id, flow_end_s, value, agg_flow="", value_op="auto", lattice=None, emit_all=False, out="host",
                   alpha=0.0, job_id="", num_keys=None):
        pk, n, dev, keep1 = _as_column(key_id, np.uint64)
        pt, nt, dev_t, keep2 = _as_column(flow_end_s, np.int64)
        pv, nv, dev_v, keep3 = _as_column(value, np.uint64)
        if nt != n or nv != n or dev_t != dev or dev_v != dev:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "columns must have equal length and live in the same memory")
        job = capi.Job(algo=capi.TAD_ALGO["EWMA"], agg_flow=capi.TAD_AGG[agg_flow], value_op=capi.TAD_OP[value_op], ewma_alpha=float(alpha),
                       flags=capi.TAD_FLAG_EMIT_ALL_POINTS if emit_all else 0, id=job_id.encode()[:63])
        cols = capi.Columns(n_rows=n, key_id=pk, flow_end_s=pt, value=pv, num_keys=state.num_keys if num_keys is None else int(num_keys),
                            memory=capi.TAD_MEM_DEVICE if dev else capi.TAD_MEM_HOST)
        if lattice is not None:
            cols.t0, cols.step, cols.n_buckets = int(lattice[0]), int(lattice[1]), int(lattice[2])
        res = C.POINTER(capi.Result)()
        rc = self._lib.tad_run_stream(self._h, state._h, C.byref(job), C.byref(cols),
                                      capi.TAD_MEM_DEVICE if out == "device" else capi.TAD_MEM_HOST, C.byref(res))
        del keep1, keep2, keep3
        self._check(rc)
        return TadResult(self, res)

    # ---- row-sharded ingest: bucket device rows by owner = key mod world (tad_shard_rows) ----
    def shard_rows(self, key_id, flow_end_s, value, world):
        """Device columns (torch CUDA tensors or DeviceArray) -> ((key_local, flow_end_s, value) DeviceArrays grouped by
        destination rank, counts per destination): the payload and the send splits of the all-to-all(v)."""
        pk, n, dev, keep1 = _as_column(key_id, np.uint64)
        pt, nt, dev_t, keep2 = _as_column(flow_end_s, np.int64)
        pv, nv, dev_v, keep3 = _as_column(value, np.uint64)
        if nt != n or nv != n or not (dev and dev_t and dev_v):
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "shard_rows: three device columns of equal length")
        outs = (DeviceArray(self, n, np.uint64), DeviceArray(self, n, np.int64), DeviceArray(self, n, np.uint64))
        counts = np.zeros(int(world), dtype=np.uint64)
        cols = capi.Columns(n_rows=n, key_id=pk, flow_end_s=pt, value=pv, num_keys=0, memory=capi.TAD_MEM_DEVICE)
        rc = self._lib.tad_shard_rows(self._h, C.byref(cols), int(world), outs[0].ptr, outs[1].ptr, outs[2].ptr, counts.ctypes.data)
        del keep1, keep2, keep3
        self._check(rc)
        return outs, [int(c) for c in counts]

    # ---- columnar ingest: Arrow buffers -> 8-byte device columns (tad_widen_column / tad_mask_rows) ----
    def widen_into(self, dst, dst_offset, src_ptr, bits, signed, n, src_device=False, table=None):
        """dst[dst_offset + i] = table[src[i]] (table: DeviceArray of int64) or src[i] widened, for i < n.  dst: DeviceArray of 8-byte
        elements; src_ptr: address of n integers of `bits` bits in host memory (or device memory with src_device)."""
        if dst_offset < 0 or dst_offset + n > dst.n:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "widen_into: rows %d..%d do not fit a column of %d" % (dst_offset, dst_offset + n, dst.n))
        self._check(self._lib.tad_widen_column(self._h, src_ptr, int(bits), 1 if signed else 0, capi.TAD_MEM_DEVICE if src_device else capi.TAD_MEM_HOST,
                                               int(n), table.ptr if table is not None else None, table.n if table is not None else 0,
                                               dst.ptr + 8 * int(dst_offset)))

    def gather(self, column, rows):
        """column[rows] for a device column (DeviceArray of 8-byte elements) and device row numbers (DeviceArray u64) -> numpy array"""
        out = DeviceArray(self, max(rows.n, 1), column.dtype)
        out.n = rows.n
        if rows.n:
            self._check(self._lib.tad_widen_column(self._h, rows.ptr, 64, 0, capi.TAD_MEM_DEVICE, rows.n, column.ptr, column.n, out.ptr))
        host = out.to_host()
        out.free()
        return host

    def mask_rows(self, n, terms, keep=None):
        """terms: list of (codes DeviceArray int64[n], mask numpy bool[D]) -> DeviceArray uint8-as-bytes keep[n] = AND of mask[codes[i]] (ANDed into
        `keep` when one is given).  The masks are per DISTINCT value (the host evaluated the SQL's string predicates on the dictionaries)."""
        if not 0 <= len(terms) <= 8:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "mask_rows: at most 8 terms")
        combine = keep is not None
        if keep is None:
            keep = DeviceArray(self, (n + 7) // 8 + 1, np.uint64)     # n bytes, allocated in 8-byte elements
        masks = []
        for _, m in terms:
            a = np.ascontiguousarray(np.asarray(m, dtype=bool).astype(np.uint8))
            masks.append((DeviceArray.from_host(self, np.frombuffer(a.tobytes() + b"\0" * (-a.size % 8 or 8), dtype=np.uint64)), a.size))
        k = len(terms)
        codes_p = (C.c_void_p * max(k, 1))(*[c.ptr for c, _ in terms])
        masks_p = (C.c_void_p * max(k, 1))(*[d.ptr for d, _ in masks])
        lens = (capi.u64 * max(k, 1))(*[ln for _, ln in masks])
        if k == 0 and not combine:      # no predicate: every row is kept
            ones = np.ones(((n + 7) // 8 + 1) * 8, dtype=np.uint8)
            self._check(self._lib.tad_copy_to_device(self._h, keep.ptr, ones.ctypes.data, ones.size))
        elif k:
            self._check(self._lib.tad_mask_rows(self._h, int(n), k, codes_p, masks_p, lens, 1 if combine else 0, keep.ptr))
        for d, _ in masks:
            d.free()
        return keep

    # ---- ingest: key tuples -> dense ids in order of first appearance (tad_factorize) ----
    def factorize(self, cols_a, keep_a=None, cols_b=None, keep_b=None, max_keys=None, with_hist=False):
        """cols_a: list of 1..8 equally long int64 arrays (numpy on the host, or DeviceArray / device pointers all on the device) —
        the key tuple of every row; keep_a: bool / uint8 mask (None = every row); cols_b / keep_b: the second tuple of every row
        (pod mode).  Returns (key_id u64[n], key_id2 u64[n] or None, first_row u64[num_keys]) — ids in order of first appearance over
        the virtual rows [side a ++ side b], TAD_KEY_SKIP where the mask is 0 — in the memory the inputs live in.  with_hist: a fourth
        return value, the KeyHistogram of the ids (tad_factorize_hist) for TadEngine.run(key_hist=...) on the same batch."""
        ncol = len(cols_a)
        if not 1 <= ncol <= 8 or (cols_b is not None and len(cols_b) != ncol):
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "factorize: 1..8 key columns, the same number on both sides")
        keepalive = []

        def col(x, dtype):
            p, n, dev, keep = _as_column(x, dtype)
            keepalive.append(keep)
            return p, n, dev

        pa = [col(c, np.int64) for c in cols_a]
        n, dev = pa[0][1], pa[0][2]
        pb = [col(c, np.int64) for c in cols_b] if cols_b is not None else None
        if any(q[1] != n or q[2] != dev for q in pa + (pb or [])):
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "factorize: columns must have equal length and live in the same memory")

        def mask(m):
            if m is None:
                return None
            if isinstance(m, DeviceArray):
                return m.ptr
            a = np.ascontiguousarray(np.asarray(m).astype(np.uint8, copy=False))
            if a.size != n:
                raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "factorize: mask length")
            if dev:
                d = DeviceArray.from_host(self, np.frombuffer(a.tobytes() + b"\0" * (-a.size % 8), dtype=np.uint64))
                keepalive.append(d)
                return d.ptr
            keepalive.append(a)
            return a.ctypes.data

        ka, kb = mask(keep_a), mask(keep_b)
        arr_a = (C.c_void_p * ncol)(*[q[0] for q in pa])
        arr_b = (C.c_void_p * ncol)(*[q[0] for q in pb]) if pb is not None else None
        sides = 2 if pb is not None else 1
        cap = int(max_keys) if max_keys is not None else n * sides
        kc = capi.KeyColumns(n_rows=n, n_cols=ncol, cols_a=arr_a, keep_a=ka, cols_b=arr_b, keep_b=kb,
                             memory=capi.TAD_MEM_DEVICE if dev else capi.TAD_MEM_HOST)
        nk = capi.u64()
        hist = KeyHistogram(self) if with_hist else None

        def call(k1, k2, fr):
            if hist is not None:
                return self._lib.tad_factorize_hist(self._h, C.byref(kc), k1, k2, fr, cap, C.byref(nk), C.byref(hist.c))
            return self._lib.tad_factorize(self._h, C.byref(kc), k1, k2, fr, cap, C.byref(nk))
        if dev:
            key1 = DeviceArray(self, n, np.uint64)
            key2 = DeviceArray(self, n, np.uint64) if pb is not None else None
            first = DeviceArray(self, max(cap, 1), np.uint64)
            rc = call(key1.ptr, key2.ptr if key2 is not None else None, first.ptr)
            del keepalive
            self._check(rc)
            first.n = min(int(nk.value), cap)
            return (key1, key2, first, hist) if with_hist else (key1, key2, first)
        key1 = np.empty(n, dtype=np.uint64)
        key2 = np.empty(n, dtype=np.uint64) if pb is not None else None
        first = np.empty(max(cap, 1), dtype=np.uint64)
        rc = call(key1.ctypes.data, key2.ctypes.data if key2 is not None else None, first.ctypes.data)
        del keepalive
        self._check(rc)
        first = first[:min(int(nk.value), cap)]
        return (key1, key2, first, hist) if with_hist else (key1, key2, first)

    # ---- ingest, one step earlier: an Arrow string column -> dictionary codes (tad_encode_strings) ----
    def encode_strings(self, column, max_values=None):
        """column: a pyarrow string / large_string / binary / large_binary Array (host memory; slices and nulls are fine: a null encodes like
        ""), or a tuple (offsets, data) / (offsets, data, validity, validity_offset) — numpy arrays on the host (offsets int32 or int64, data
        uint8), or DeviceArrays on the device (offsets as int32 / int64 elements).  Returns (codes int64[n], first_row u64[num_values]): codes
        in order of first appearance (pyarrow.compute.dictionary_encode's, pandas.factorize's), first_row[k] = the row where value k
        first appears — in the memory the input lives in."""
        keepalive = []
        validity, voff = None, 0
        if hasattr(column, "combine_chunks") and hasattr(column, "chunks"):     # a pyarrow ChunkedArray: one contiguous column first
            column = column.combine_chunks() if column.num_chunks != 1 else column.chunk(0)
        if hasattr(column, "buffers") and hasattr(column, "type"):        # a pyarrow Array
            import pyarrow as pa
            t = column.type
            if pa.types.is_string(t) or pa.types.is_binary(t):
                bits = 32
            elif pa.types.is_large_string(t) or pa.types.is_large_binary(t):
                bits = 64
            else:
                raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "encode_strings: not a string / binary column: %s" % t)
            vbuf, obuf, dbuf = column.buffers()
            n, dev = len(column), False
            off_ptr = obuf.address + column.offset * (bits // 8) if obuf is not None else None
            data_ptr, data_bytes = (dbuf.address, dbuf.size) if dbuf is not None else (None, 0)
            if vbuf is not None and column.null_count:
                validity, voff = vbuf.address, column.offset
            keepalive.append(column)
            if n and off_ptr is None:
                raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "encode_strings: the column has no offsets buffer")
        else:
            offsets, data = column[0], column[1]
            if isinstance(offsets, DeviceArray):
                dev, bits, n = True, offsets.dtype.itemsize * 8, offsets.n - 1
                off_ptr, data_ptr, data_bytes = offsets.ptr, data.ptr, data.n * data.dtype.itemsize
                if len(column) > 2 and column[2] is not None:
                    validity, voff = column[2].ptr, int(column[3]) if len(column) > 3 else 0
            else:
                dev = False
                offsets = np.ascontiguousarray(offsets)
                if offsets.dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
                    raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "encode_strings: offsets must be int32 or int64")
                data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data)
                bits, n = offsets.dtype.itemsize * 8, offsets.size - 1
                off_ptr, data_ptr, data_bytes = offsets.ctypes.data, (data.ctypes.data if data.size else None), data.size
                if len(column) > 2 and column[2] is not None:
                    v = np.ascontiguousarray(column[2], dtype=np.uint8)
                    keepalive.append(v)
                    validity, voff = v.ctypes.data, int(column[3]) if len(column) > 3 else 0
            keepalive += [offsets, data]
        if n < 0:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "encode_strings: the offsets hold n + 1 entries")
        cap = int(max_values) if max_values is not None else n
        sc = capi.StringColumn(n_rows=n, offsets=off_ptr, offset_bits=bits, data=data_ptr, data_bytes=data_bytes, validity=validity,
                               validity_offset=voff, memory=capi.TAD_MEM_DEVICE if dev else capi.TAD_MEM_HOST)
        nv = capi.u64()
        if dev:
            codes = DeviceArray(self, max(n, 1), np.int64)
            first = DeviceArray(self, max(cap, 1), np.uint64)
            rc = self._lib.tad_encode_strings(self._h, C.byref(sc), codes.ptr, first.ptr, cap, C.byref(nv))
            del keepalive
            self._check(rc)
            codes.n, first.n = n, min(int(nv.value), cap)
            return codes, first
        codes = np.empty(n, dtype=np.int64)
        first = np.empty(max(cap, 1), dtype=np.uint64)
        rc = self._lib.tad_encode_strings(self._h, C.byref(sc), codes.ctypes.data, first.ctypes.data, cap, C.byref(nv))
        del keepalive
        self._check(rc)
        return codes, first[:min(int(nv.value), cap)]

    # ---- Stage 0 alone: the GROUP BY (anomaly_detection.py:507-614) ----
    def aggregate(self, key_id, flow_end_s, value, num_keys, agg_flow="", value_op="auto", key_id2=None, flow_start_s=None,
                  start_time=0, end_time=0, lattice=None, out="host"):
        if agg_flow not in capi.TAD_AGG:
            raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "agg_flow must be '', pod, svc or external")
        pk, n, dev, keep1 = _as_column(key_id, np.uint64)
        pt, nt, dev_t, keep2 = _as_column(flow_end_s, np.int64)
        pv, nv, dev_v, keep3 = _as_column(value, np.uint64)
        pk2, nk2, dev_k2, keep4 = _as_column(key_id2, np.uint64)
        ps, ns, dev_s, keep5 = _as_column(flow_start_s, np.int64)
        for m, d in ((nt, dev_t), (nv, dev_v)) + (((nk2, dev_k2),) if key_id2 is not None else ()) + \
                (((ns, dev_s),) if flow_start_s is not None else ()):
            if m != n or d != dev:
                raise TadError(capi.TAD_ERR_INVALID_ARGUMENT, "columns must have equal length and live in the same memory")
        job = capi.Job(algo=0, agg_flow=capi.TAD_AGG[agg_flow], value_op=capi.TAD_OP[value_op],
                       start_time=int(start_time), end_time=int(end_time))
        cols = capi.Columns(n_rows=n, key_id=pk, key_id2=pk2, flow_end_s=pt, flow_start_s=ps, value=pv,
                            num_keys=int(num_keys), memory=capi.TAD_MEM_DEVICE if dev else capi.TAD_MEM_HOST)
        if lattice is not None:
            cols.t0, cols.step, cols.n_buckets = int(lattice[0]), int(lattice[1]), int(lattice[2])
        res = C.POINTER(capi.Points)()
        rc = self._lib.tad_aggregate(self._h, C.byref(job), C.byref(cols),
                                     capi.TAD_MEM_DEVICE if out == "device" else capi.TAD_MEM_HOST, C.byref(res))
        del keep1, keep2, keep3, keep4, keep5
        self._check(rc)
        return TadPoints(self, res)

    # ---- the reference's per-series pure functions, on the GPU ----
    @staticmethod
    def _series(x):
        a = np.ascontiguousarray(np.asarray([int(v) for v in x] if not isinstance(x, np.ndarray) else x, dtype=np.uint64))
        return a

    def series_ewma(self, x, alpha=0.0):
        a = self._series(x)
        out = np.empty(a.size, dtype=np.float64)
        self._check(self._lib.tad_series_ewma(self._h, a.ctypes.data, a.size, float(alpha), out.ctypes.data))
        return out

    def series_ewma_anomaly(self, x, stddev, alpha=0.0):
        a = self._series(x)
        out = np.zeros(a.size, dtype=np.uint8)
        self._check(self._lib.tad_series_ewma_anomaly(self._h, a.ctypes.data, a.size, float(alpha),
                                                      0 if stddev is None else 1, 0.0 if stddev is None else float(stddev),
                                                      out.ctypes.data))
        return out.astype(bool)

    def series_stddev(self, x):
        a = self._series(x)
        has, sd = C.c_int(), capi.f64()
        self._check(self._lib.tad_series_stddev(self._h, a.ctypes.data, a.size, C.byref(has), C.byref(sd)))
        return sd.value if has.value else None

    def series_dbscan_anomaly(self, x, eps=0.0, min_samples=0):
        a = self._series(x)
        out = np.zeros(a.size, dtype=np.uint8)
        self._check(self._lib.tad_series_dbscan_anomaly(self._h, a.ctypes.data, a.size, float(eps), int(min_samples), out.ctypes.data))
        return out.astype(bool)

    def series_drop(self, x, nsigma=0.0, min_samples=0):
        """DropDetection.end_partition on one partition -> None (too few samples) or (mean, std, verdict bool[n])."""
        a = self._series(x)
        out = np.zeros(max(a.size, 1), dtype=np.uint8)
        has, mean, sd = C.c_int(), capi.f64(), capi.f64()
        self._check(self._lib.tad_series_drop(self._h, a.ctypes.data, a.size, float(nsigma), int(min_samples), C.byref(has),
                                              C.byref(mean), C.byref(sd), out.ctypes.data))
        return (mean.value, sd.value, out[:a.size].astype(bool)) if has.value else None

    def series_arima(self, x, maxiter=0):
        a = self._series(x)
        out = np.empty(a.size, dtype=np.float64)
        has = C.c_int()
        self._check(self._lib.tad_series_arima(self._h, a.ctypes.data, a.size, int(maxiter), C.byref(has), out.ctypes.data))
        return out if has.value else None

    def series_arima_anomaly(self, x, stddev, maxiter=0):
        a = self._series(x)
        out = np.zeros(max(a.size, 1), dtype=np.uint8)
        nv = capi.u64()
        self._check(self._lib.tad_series_arima_anomaly(self._h, a.ctypes.data, a.size, int(maxiter),
                                                       0 if stddev is None else 1, 0.0 if stddev is None else float(stddev),
                                                       out.ctypes.data, C.byref(nv)))
        return out[:nv.value].astype(bool)

    # ---- synthetic table straight into HBM ----
    def synth(self, first_row, n_rows, num_keys, n_buckets, seed=SYNTH_SEED, into=None):
        """Returns (key_id, flow_end_s, value) as DeviceArray, or fills the 3 given torch CUDA tensors."""
        if into is None:
            cols = (DeviceArray(self, n_rows, np.uint64), DeviceArray(self, n_rows, np.int64), DeviceArray(self, n_rows, np.uint64))
            ptrs = [c.ptr for c in cols]
        else:
            cols = into
            ptrs = [t.data_ptr() for t in into]
        self._check(self._lib.tad_synth_generate(self._h, int(seed), int(first_row), int(n_rows), int(num_keys), int(n_buckets), *ptrs))
        return cols
