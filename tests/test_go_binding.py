"""go/tadengine/tadengine.go has never seen a compiler (no Go toolchain in the image), so what CAN be checked is checked statically against
include/tad.h: every C function, struct field, enum constant and macro the Go file touches is declared by the header, the entry points a
theia-manager host needs are bound, and the library's ABI version is verified before the first struct crosses the boundary."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "tad.h")).read()
GO = open(os.path.join(ROOT, "go", "tadengine", "tadengine.go")).read()


def header_structs():
    out = {}
    for body, name in re.findall(r"typedef struct(?: \w+)? \{(.*?)\} (tad_\w+);", HEADER, flags=re.S):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out[name] = set(re.findall(r"\b(\w+)(?:\[\d+\])?\s*;", body))
    return out


def test_every_c_function_the_go_file_calls_is_declared():
    declared = set(re.findall(r"\b(tad_[a-z0-9_]+)\s*\(", HEADER))
    types = set(re.findall(r"\}\s*(tad_\w+);", HEADER)) | set(re.findall(r"typedef struct (tad_\w+) tad_\w+;", HEADER)) | \
        set(re.findall(r"typedef enum \{.*?\} (tad_\w+);", HEADER, flags=re.S))
    called = set(re.findall(r"C\.(tad_[a-z0-9_]+)\(", GO))
    conversions = {c for c in called if c in types}          # C.tad_algo(x) is a type conversion, not a call
    assert called - conversions <= declared, sorted(called - conversions - declared)
    for must in ("tad_abi_version", "tad_engine_create", "tad_engine_destroy", "tad_engine_set_plan", "tad_last_error", "tad_run", "tad_result_free",
                 "tad_aggregate", "tad_points_free", "tad_shard_rows", "tad_factorize", "tad_encode_strings", "tad_progress", "tad_state_create",
                 "tad_state_destroy", "tad_run_stream", "tad_state_export", "tad_device_alloc", "tad_device_free", "tad_copy_to_device",
                 "tad_copy_to_host", "tad_job_progress", "tad_jobs_in_flight", "tad_widen_column", "tad_mask_rows", "tad_host_alloc", "tad_host_free", "tad_factorize_hist"):
        assert must in called, must
    for t in set(re.findall(r"C\.(tad_[a-z0-9_]+)\b(?!\()", GO)):
        assert t in types or t in declared, t


def test_every_constant_the_go_file_uses_is_defined():
    consts = set(re.findall(r"\b(TAD_[A-Z0-9_]+)\b", HEADER))
    for c in set(re.findall(r"C\.(TAD_[A-Z0-9_]+)\b", GO)):
        assert c in consts, c


def test_struct_fields_exist_in_the_header():
    structs = header_structs()
    # variables the Go file declares with a C struct type, and the fields it reads or writes on them
    var_types = dict(re.findall(r"var (\w+) C\.(tad_\w+)\n", GO))
    var_types.update({"opts": "tad_engine_opts"})
    for var, st in var_types.items():
        for field in set(re.findall(r"\b%s\.(\w+)\b" % re.escape(var), GO)):
            assert field in structs[st], (var, st, field)
    # composite literals
    for st, body in re.findall(r"C\.(tad_\w+)\{(.*?)\}", GO, flags=re.S):
        for field in re.findall(r"(\w+):", body):
            assert field in structs[st], (st, field)
    # result / points / stats accessed through pointers
    for field in set(re.findall(r"\bres\.stats\.(\w+)", GO)):
        assert field in structs["tad_stats"], field
    for field in set(re.findall(r"\bres\.(\w+)", GO)) - {"stats"}:
        assert field in structs["tad_result"], field
    for field in set(re.findall(r"\bpts\.(\w+)", GO)):
        assert field in structs["tad_points"], field


def test_the_go_plan_mirrors_tad_plan_field_for_field():
    fields = re.findall(r"\b(\w+);", re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct \{(.*?)\} tad_plan;", HEADER, flags=re.S).group(1), flags=re.S))
    lit = re.search(r"return C\.tad_plan\{(.*?)\}", GO, flags=re.S).group(1)
    assert re.findall(r"(\w+):", lit) == [f for f in fields if f in lit], "order / names of the plan literal"
    live = {f for f in fields if not f.startswith("reserved")}      # reserved fields stay zero: the literal leaves them out
    assert set(re.findall(r"(\w+):", lit)) == live, sorted(live - set(re.findall(r"(\w+):", lit)))


def test_abi_version_is_checked_before_an_engine_is_created():
    body = GO[GO.index("func NewEngineWithOptions"):]
    assert body.index("C.tad_abi_version()") < body.index("C.tad_engine_create(")


def test_go_pointers_stored_in_c_structs_are_pinned():
    """cgo pointer rules (round-4 advisor finding): a Go struct passed to C may only hold Go pointers to PINNED memory.  Every
    `<struct var>.<field> = ...unsafe.Pointer(&slice[0])` store must be preceded, in the same function, by `pin.Pin(&slice[0])`;
    pointers handed over as plain call arguments need nothing."""
    funcs = re.split(r"\nfunc ", GO)
    stores = 0
    for f in funcs:
        for var, field, sl in re.findall(r"\b(\w+)\.(\w+) = (?:\(\*C\.\w+\)\()?unsafe\.Pointer\(&(\w+)\[0\]\)", f):
            stores += 1
            pin = f.find("pin.Pin(&%s[0])" % sl)
            store = f.find("%s.%s = " % (var, field))
            assert 0 <= pin < store, (var, field, sl)
            assert "var pin runtime.Pinner" in f and "defer pin.Unpin()" in f
    assert stores >= 3          # EncodeStrings: offsets, data, validity
