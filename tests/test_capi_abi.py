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
                 "tad_engine_destroy", "tad_series_ewma", "tad_series_dbscan_anomaly", "tad_series_arima"):
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
    prog.write_text('#include <stdio.h>\n#include "tad.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                    'sizeof(tad_engine_opts),sizeof(tad_job),sizeof(tad_columns),sizeof(tad_stats),sizeof(tad_result));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mine = [ctypes.sizeof(c) for c in (_capi.EngineOpts, _capi.Job, _capi.Columns, _capi.Stats, _capi.Result)]
    assert sizes == mine


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
