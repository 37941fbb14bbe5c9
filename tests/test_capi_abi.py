"""CPU tests of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/tad.h declares; struct layouts of the ctypes view match the compiler's.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tad.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tad_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("tad_engine_create", "tad_run", "tad_result_free", "tad_progress", "tad_last_error",
                 "tad_engine_destroy", "tad_series_ewma", "tad_series_dbscan_anomaly", "tad_series_arima", "tad_aggregate",
                 "tad_points_free"):
        assert must in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    from theia_amd import _capi, build
    path = build.build_library()
    assert os.path.exists(path)
    lib = _capi.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), "libtad_mi355x.so does not export %s" % name
        assert name in _capi.SYMBOLS, "ctypes binding lacks %s" % name
    assert lib.tad_abi_version() == _capi.TAD_ABI_VERSION


def test_ctypes_struct_layout_matches_the_c_compiler(tmp_path):
    from theia_amd import _capi
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "tad.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(tad_plan),sizeof(tad_engine_opts),sizeof(tad_job),sizeof(tad_columns),sizeof(tad_stats),sizeof(tad_result),sizeof(tad_points),'
                    'sizeof(tad_key_columns),sizeof(tad_string_column),sizeof(tad_key_hist));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mine = [ctypes.sizeof(c) for c in (_capi.Plan, _capi.EngineOpts, _capi.Job, _capi.Columns, _capi.Stats, _capi.Result, _capi.Points, _capi.KeyColumns,
                                       _capi.StringColumn, _capi.KeyHist)]
    assert sizes == mine


def test_library_reads_no_environment_variable():
    """ABI 7: plan overrides are fields of tad_engine_opts / tad_engine_set_plan; a host with several workers
    (controller.go:199-201) cannot scope process-global switches, so the library has none."""
    for dp, _, files in os.walk(os.path.join(ROOT, "theia_amd", "csrc")):
        for fn in files:
            assert "getenv" not in open(os.path.join(dp, fn), errors="ignore").read(), fn
    from theia_amd import _capi
    with pytest.raises(ValueError):
        _capi.make_plan(no_such_field=1)
    p = _capi.make_plan(stage0="v2", partition_pass="sort", ewma_emit_rows=64)
    assert (p.stage0, p.partition_pass, p.ewma_emit_rows, p.sparse) == (2, 1, 64, 0)


def test_engine_create_fails_loudly_without_a_gpu():
    """No CPU fallback: without a device the engine refuses to exist."""
    from theia_amd import TadEngine, TadError
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    with pytest.raises(TadError) as ei:
        TadEngine(device=0)
    assert ei.value.code == -2   # TAD_ERR_NO_DEVICE


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under theia_amd/ or include/ may import or link it."""
    bad = []
    for base in ("theia_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".cpp", ".h")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+\"[^\"]*oracle", txt, flags=re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_host_emulator_is_a_development_aid_only():
    """tools/hipemu executes the kernels' logic on the host to debug them before GPU time is spent.  It must never become a
    code path: nothing in the product, the tests, bench.py or __graft_entry__ refers to it, and its build never reaches the GPU box."""
    bad = []
    for base in ("theia_amd", "include", "tests", "oracle"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".cpp", ".h", ".c")) and fn != os.path.basename(__file__):
                    if "hipemu" in open(os.path.join(dp, fn), errors="ignore").read():
                        bad.append(os.path.join(dp, fn))
    for fn in ("bench.py", "__graft_entry__.py"):
        if "hipemu" in open(os.path.join(ROOT, fn)).read():
            bad.append(fn)
    assert not bad, bad
    assert "tools/hipemu/_build/" in open(os.path.join(ROOT, ".gpurunignore")).read().split()


def _build_c_driver(tmp_path):
    from theia_amd import build
    build.build_library()
    exe = tmp_path / "capi_driver"
    lib_dir = os.path.join(ROOT, "theia_amd", "lib")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "capi_driver.c"), "-L", lib_dir, "-ltad_mi355x",
                           "-Wl,-rpath," + lib_dir, "-o", str(exe)])
    return exe


def test_plain_c_caller_compiles_against_the_header_and_fails_loudly_without_a_gpu(tmp_path):
    """include/tad.h is C (what cgo parses): a C11 translation unit must compile -Werror clean and link."""
    exe = _build_c_driver(tmp_path)
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_plain_c_caller_runs_the_job(tmp_path):
    import numpy as np
    from oracle import tad_oracle as orc
    exe = _build_c_driver(tmp_path)
    for algo in ("EWMA", "DBSCAN"):
        r = subprocess.run([str(exe), algo], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        # the same table as tools/capi_driver.c builds
        key, tend, val = [], [], []
        for k in range(3):
            for t in range(30):
                for dup in range(2):
                    key.append(k); tend.append(1660202814 + 60 * t)
                    val.append(2000000000 + 1000 * t + dup + (30000000000 if t == 20 + k else 0))
        want = orc.run_job(algo, np.array(key, dtype=np.uint64), np.array(tend), np.array(val, dtype=np.uint64), agg_flow="svc")
        rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("row ")]
        assert len(rows) == want["n_anomalies"] > 0
        for i, f in enumerate(rows):
            d = dict(x.split("=") for x in f[1:])
            assert int(d["key"]) == int(want["key_id"][i]) and int(d["t"]) == int(want["flow_end_s"][i])
            assert float(d["x"]) == want["throughput"][i] and float(d["calc"]) == want["algo_calc"][i] and float(d["sd"]) == want["stddev"][i]
        assert "illegal algo -> -1: invalid request: Throughput Anomaly Detector algorithm type should be" in r.stdout
