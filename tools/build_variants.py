#!/usr/bin/env python3
"""Build extra variants of libtad_mi355x.so with compile-time defines for same-box A/B runs (bench.py / tests load one through
TAD_LIBRARY_PATH).  Measurement aid: the shipped library is theia_amd/lib/libtad_mi355x.so, built by theia_amd/build.py.
usage: python tools/build_variants.py name:DEF1,DEF2,-rawflag [name2:DEF ...]   ->  theia_amd/lib/variants/libtad_<name>.so"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theia_amd import build as b  # noqa: E402


def build_variant(name, defines):
    out_dir = os.path.join(b.LIB_DIR, "variants")
    obj_dir = os.path.join("/tmp", "tad_variants", "obj_" + name)   # (objects stay out of the tree: they would travel to the GPU box)
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    flags = [f for f in b.FLAGS if f != "-shared"] + [d if d.startswith("-") else "-D" + d for d in defines if d]   # (an entry starting with "-" is a raw compiler flag)

    def cc(src):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        r = subprocess.run([b._hipcc()] + flags + ["-I" + os.path.join(ROOT, "include"), "-c", "-o", obj, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-4000:])
        return obj
    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(cc, b.sources()))
    lib = os.path.join(out_dir, "libtad_%s.so" % name)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        name, _, defs = spec.partition(":")
        print(build_variant(name, defs.split(",")))
