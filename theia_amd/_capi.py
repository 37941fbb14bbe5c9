"""ctypes declarations for include/tad.h — field for field what a cgo `import "C"` block sees."""
import ctypes as C
import os

from . import build as _build

u64, i64, i32, u32, f64, f32 = C.c_uint64, C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.c_float

TAD_ABI_VERSION = 12
TAD_KEY_SKIP = (1 << 64) - 1
TAD_OK = 0
TAD_ERR_INVALID_ARGUMENT, TAD_ERR_NO_DEVICE, TAD_ERR_OUT_OF_MEMORY, TAD_ERR_HIP = -1, -2, -3, -4
TAD_ERR_KEY_RANGE, TAD_ERR_GRID_TOO_LARGE, TAD_ERR_BUSY = -5, -6, -7
TAD_ALGO = {"EWMA": 0, "ARIMA": 1, "DBSCAN": 2, "DROP": 3}
TAD_AGG = {"": 0, None: 0, "None": 0, "pod": 1, "svc": 2, "external": 3}
TAD_OP = {"auto": 0, "max": 1, "sum": 2}
TAD_MEM_HOST, TAD_MEM_DEVICE = 0, 1
TAD_FLAG_EMIT_ALL_POINTS = 1


class Plan(C.Structure):
    """tad_plan: plan overrides, every field 0 = the engine decides (tests and A/B measurements set them)."""
    _fields_ = [("stage0", i32), ("partition_pass", i32), ("histogram", i32), ("sparse", i32), ("sparse_classes", i32),
                ("ewma_emit", i32), ("ewma_emit_rows", u32), ("reserved0", i32), ("tile_cells", i32), ("sparse_sort", i32), ("reserved1", i32)]


PLAN_VALUES = {   # symbolic values accepted by TadEngine(plan=...) / TadEngine.plan(...)
    "stage0": {"auto": 0, "v1": 1, "v2": 2}, "partition_pass": {"auto": 0, "sort": 1, "wc": 2, "wc_sectors": 3}, "histogram": {"auto": 0, "exact": 1, "sampled": 2},
    "sparse": {"auto": 0, "never": 1, "always": 2}, "sparse_classes": {"auto": 0, "always": 1}, "ewma_emit": {"auto": 0, "staged": 0, "lane": 1}, "tile_cells": {"auto": 0, "wide": 1},
    "sparse_sort": {"auto": 0, "lsd": 1, "partition": 2},
}


def make_plan(**kw):
    p = Plan()
    for name, v in kw.items():
        if name not in dict((f[0], 1) for f in Plan._fields_):
            raise ValueError("unknown tad_plan field %r" % name)
        if isinstance(v, str):
            v = PLAN_VALUES[name][v]
        setattr(p, name, int(v))
    return p


class KeyColumns(C.Structure):
    """tad_key_columns: the key tuples of a batch for tad_factorize (up to 8 int64 columns, optional keep masks, optional second side)."""
    _fields_ = [("n_rows", u64), ("n_cols", i32), ("cols_a", C.POINTER(C.c_void_p)), ("keep_a", C.c_void_p),
                ("cols_b", C.POINTER(C.c_void_p)), ("keep_b", C.c_void_p), ("memory", C.c_int)]


class StringColumn(C.Structure):
    """tad_string_column: one Arrow string column (offsets + bytes + optional validity bitmap) for tad_encode_strings."""
    _fields_ = [("n_rows", u64), ("offsets", C.c_void_p), ("offset_bits", i32), ("data", C.c_void_p), ("data_bytes", u64),
                ("validity", C.c_void_p), ("validity_offset", u64), ("memory", C.c_int)]


class EngineOpts(C.Structure):
    _fields_ = [("device", i32), ("stream", C.c_void_p), ("workspace_limit", u64), ("plan", Plan), ("max_jobs_in_flight", i32), ("reserved", i32)]


class Job(C.Structure):
    _fields_ = [("algo", C.c_int), ("agg_flow", C.c_int), ("value_op", C.c_int),
                ("start_time", i64), ("end_time", i64), ("ewma_alpha", f64), ("dbscan_eps", f64),
                ("dbscan_min_samples", i32), ("arima_maxiter", i32), ("drop_nsigma", f64), ("drop_min_samples", i32),
                ("flags", u32), ("id", C.c_char * 64)]


TAD_KEY_HIST_BYTES = 256 * 16384 * 4


class KeyHist(C.Structure):
    """tad_key_hist: tad_factorize_hist's by-product (key-bin histogram per Stage-0 workgroup); bins is device memory of TAD_KEY_HIST_BYTES."""
    _fields_ = [("n_rows", u64), ("num_keys", u64), ("chunk_rows", u64), ("workgroups", u32), ("nbins", u32), ("shift", u32), ("sides", u32),
                ("bins", C.c_void_p)]


class Columns(C.Structure):
    _fields_ = [("n_rows", u64), ("key_id", C.c_void_p), ("key_id2", C.c_void_p),
                ("flow_end_s", C.c_void_p), ("flow_start_s", C.c_void_p), ("value", C.c_void_p),
                ("num_keys", u64), ("memory", C.c_int), ("t0", i64), ("step", i64), ("n_buckets", u64), ("key_hist", C.POINTER(KeyHist))]


class Stats(C.Structure):
    _fields_ = [("rows_in", u64), ("rows_used", u64), ("n_keys", u64), ("n_points", u64),
                ("n_anomalies", u64), ("keys_no_result", u64), ("kalman_steps", u64),
                ("arima_fits", u64), ("arima_nan_fits", u64), ("pts_mean", f64), ("pts_m2", f64), ("t0", i64), ("step", i64), ("n_buckets", u64),
                ("ms_meta", f32), ("ms_stage0", f32), ("ms_scatter", f32), ("ms_detect", f32),
                ("ms_total", f32), ("stage0_path", i32), ("stage0_attempts", i32), ("hist_sampled", i32), ("host_syncs", i32),
                ("job_context", i32), ("arima_relaunches", i32)]


class Result(C.Structure):
    _fields_ = [("n_rows", u64), ("key_id", C.c_void_p), ("flow_end_s", C.c_void_p),
                ("throughput", C.c_void_p), ("algo_calc", C.c_void_p), ("stddev", C.c_void_p),
                ("anomaly", C.c_void_p), ("memory", C.c_int), ("stats", Stats), ("id", C.c_char * 64)]


class Points(C.Structure):
    _fields_ = [("n_points", u64), ("key_id", C.c_void_p), ("flow_end_s", C.c_void_p), ("value", C.c_void_p),
                ("memory", C.c_int), ("stats", Stats)]


# every symbol include/tad.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tad_abi_version": (C.c_int, []),
    "tad_engine_create": (C.c_int, [C.POINTER(EngineOpts), C.POINTER(C.c_void_p)]),
    "tad_engine_destroy": (None, [C.c_void_p]),
    "tad_engine_set_plan": (C.c_int, [C.c_void_p, C.POINTER(Plan)]),
    "tad_last_error": (C.c_char_p, [C.c_void_p]),
    "tad_run": (C.c_int, [C.c_void_p, C.POINTER(Job), C.POINTER(Columns), C.c_int, C.POINTER(C.POINTER(Result))]),
    "tad_result_free": (None, [C.c_void_p, C.POINTER(Result)]),
    "tad_state_create": (C.c_int, [C.c_void_p, u64, C.POINTER(C.c_void_p)]),
    "tad_state_destroy": (None, [C.c_void_p, C.c_void_p]),
    "tad_state_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tad_run_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Job), C.POINTER(Columns), C.c_int, C.POINTER(C.POINTER(Result))]),
    "tad_aggregate": (C.c_int, [C.c_void_p, C.POINTER(Job), C.POINTER(Columns), C.c_int, C.POINTER(C.POINTER(Points))]),
    "tad_points_free": (None, [C.c_void_p, C.POINTER(Points)]),
    "tad_shard_rows": (C.c_int, [C.c_void_p, C.POINTER(Columns), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tad_factorize": (C.c_int, [C.c_void_p, C.POINTER(KeyColumns), C.c_void_p, C.c_void_p, C.c_void_p, u64, C.POINTER(u64)]),
    "tad_factorize_hist": (C.c_int, [C.c_void_p, C.POINTER(KeyColumns), C.c_void_p, C.c_void_p, C.c_void_p, u64, C.POINTER(u64), C.POINTER(KeyHist)]),
    "tad_encode_strings": (C.c_int, [C.c_void_p, C.POINTER(StringColumn), C.c_void_p, C.c_void_p, u64, C.POINTER(u64)]),
    "tad_widen_column": (C.c_int, [C.c_void_p, C.c_void_p, i32, i32, C.c_int, u64, C.c_void_p, u64, C.c_void_p]),
    "tad_mask_rows": (C.c_int, [C.c_void_p, u64, i32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(u64), i32, C.c_void_p]),
    "tad_host_alloc": (C.c_int, [C.c_void_p, u64, C.POINTER(C.c_void_p)]),
    "tad_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tad_progress": (C.c_int, [C.c_void_p, C.POINTER(i32), C.POINTER(i32)]),
    "tad_job_progress": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(i32), C.POINTER(i32)]),
    "tad_jobs_in_flight": (C.c_int, [C.c_void_p]),
    "tad_series_ewma": (C.c_int, [C.c_void_p, C.c_void_p, u64, f64, C.c_void_p]),
    "tad_series_ewma_anomaly": (C.c_int, [C.c_void_p, C.c_void_p, u64, f64, C.c_int, f64, C.c_void_p]),
    "tad_series_stddev": (C.c_int, [C.c_void_p, C.c_void_p, u64, C.POINTER(C.c_int), C.POINTER(f64)]),
    "tad_series_dbscan_anomaly": (C.c_int, [C.c_void_p, C.c_void_p, u64, f64, C.c_int, C.c_void_p]),
    "tad_series_drop": (C.c_int, [C.c_void_p, C.c_void_p, u64, f64, C.c_int, C.POINTER(C.c_int), C.POINTER(f64), C.POINTER(f64), C.c_void_p]),
    "tad_series_arima": (C.c_int, [C.c_void_p, C.c_void_p, u64, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "tad_series_arima_anomaly": (C.c_int, [C.c_void_p, C.c_void_p, u64, C.c_int, C.c_int, f64, C.c_void_p, C.POINTER(u64)]),
    "tad_synth_generate": (C.c_int, [C.c_void_p, u64, u64, u64, u64, u64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tad_device_alloc": (C.c_int, [C.c_void_p, u64, C.POINTER(C.c_void_p)]),
    "tad_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tad_copy_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64]),
    "tad_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64]),
}

_libs = {}


def load_library(build_if_missing=True, path=None):
    """dlopen theia_amd/lib/libtad_mi355x.so (building it in-tree first if needed).  `path` (or TAD_LIBRARY_PATH) names another build
    of the library — measurement variants of tools/build_variants.py; one handle per path, so two builds can run in one process."""
    path = path or os.environ.get("TAD_LIBRARY_PATH") or _build.LIB_PATH   # override: A/B two builds of the library
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        if not build_if_missing or path != os.path.abspath(_build.LIB_PATH):
            raise OSError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        _build.build_library()
    lib = C.CDLL(path)
    shipped = path == os.path.abspath(_build.LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if not shipped and not hasattr(lib, name):     # an older build loaded for an A/B measurement may lack the newest entry points
            continue
        fn = getattr(lib, name)  # AttributeError here = the library does not export the header's symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tad_abi_version() != TAD_ABI_VERSION:
        raise OSError("%s ABI %d != binding ABI %d" % (os.path.basename(path), lib.tad_abi_version(), TAD_ABI_VERSION))
    _libs[path] = lib
    return lib
