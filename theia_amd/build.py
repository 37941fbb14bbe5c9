"""Build libtad_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is the product: HIP kernels + the C ABI of include/tad.h.  `-ffp-contract=off` is part
of the numerics contract (EWMA / stddev_samp must round like the reference's Python floats).
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtad_mi355x.so")
SOURCES = ["tad_kernels.hip", "tad_stage0_part.hip", "tad_dbscan.hip", "tad_arima.hip", "tad_drop.hip", "tad_synth.hip", "tad_shard.hip", "tad_sparse.hip", "tad_factorize.hip", "tad_ingest.hip", "tad_engine.cpp", "tad_capi.cpp", "tad_capi_ingest.cpp", "tad_capi_series.cpp"]
HEADERS = [os.path.join(CSRC, "tad_internal.h"), os.path.join(CSRC, "tad_engine.h"), os.path.join(CSRC, "tad_detmath.h"), os.path.join(REPO_ROOT, "include", "tad.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libtad_mi355x.so")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in sources() + HEADERS)


def _obj_path(src):
    return os.path.join(LIB_DIR, "obj", os.path.basename(src) + ".o")


def _compile_one(src):
    obj = _obj_path(src)
    deps = [src] + HEADERS + [os.path.join(CSRC, "tad_detmath.h")]
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps if os.path.exists(d)):
        return None
    cmd = [_hipcc()] + [f for f in FLAGS if f != "-shared"] + ["-I" + os.path.join(REPO_ROOT, "include"), "-c", "-o", obj, src]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
    return obj


def build_library(force=False, verbose=False):
    """Compile every HIP source (one object per file, in parallel, only what changed) and link
    theia_amd/lib/libtad_mi355x.so.  Returns the path.  force=True recompiles everything."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(os.path.join(LIB_DIR, "obj"), exist_ok=True)
    srcs = sources()
    if force:
        for src in srcs:
            if os.path.exists(_obj_path(src)):
                os.remove(_obj_path(src))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        list(ex.map(_compile_one, srcs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [_obj_path(src) for src in srcs]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in __import__("sys").argv, verbose=True))
