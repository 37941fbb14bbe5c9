#!/usr/bin/env python3
"""Does pass B's speed depend on WHERE its buffers lie?  Same table, same engine, the three input columns carved out of one allocation
at varying distances from each other / from the allocation's base; prints the pass-B time (tad_stats.ms_scatter) per placement.
usage: python tools/ubench_align.py [--config c2|c4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from theia_amd import TadEngine  # noqa: E402
from theia_amd.engine import DeviceArray  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4")
args = ap.parse_args()
cfg = {"c2": ("EWMA", 100_000_000, 100_000, 250, "svc"), "c4": ("DBSCAN", 100_000_000, 1_000_000, 100, "")}[args.config]
algo, n, K, T, agg = cfg
eng = TadEngine(device=0)
S = n * 8
slack = 64 << 20
big = DeviceArray(eng, (3 * S + 4 * slack) // 8, np.uint64)


class Ptr:
    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


def view(ptr, dtype):
    d = object.__new__(DeviceArray)
    d.engine, d.n, d.dtype, d.ptr = eng, n, np.dtype(dtype), ptr
    return d


for shift, gap in ((0, 0), (0, 256), (0, 4096), (0, 65536), (0, 1 << 20), (0, (1 << 21) + 4096), (4096, 0), (65536, 0), (1 << 20, 0), (1 << 20, 1 << 20), (768, 768),
                   (0, 8 << 20), (0, (16 << 20) + 65536)):
    pk = big.ptr + shift
    pt = pk + S + gap
    pv = pt + S + gap
    eng.synth(0, n, K, T, into=(Ptr(pk), Ptr(pt), Ptr(pv)))
    cols = (view(pk, np.uint64), view(pt, np.int64), view(pv, np.uint64))
    job = eng.prepare(algo, cols[0], cols[1], cols[2], K, agg_flow=agg, out="device")
    acc = []
    for i in range(14):
        r = job.run()
        if i >= 4:
            acc.append((r.stats["ms_scatter"], r.stats["ms_total"]))
        r.close()
    for c in cols:
        c.ptr = None
    print("%s: base + %8d, gap between columns %9d: pass B %.4f ms (min %.4f), job device %.4f ms"
          % (args.config, shift, gap, float(np.median([a[0] for a in acc])), min(a[0] for a in acc), float(np.median([a[1] for a in acc]))), flush=True)
