#!/usr/bin/env python3
"""Timing of the C2 job when a fraction of the rows carries ONE hot key (real flow tables have heavy hitters).
usage: python tools/skew_check.py [rows] [hot_fraction ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from theia_amd import TadEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
fracs = [float(x) for x in sys.argv[2:]] or [0.0, 0.1, 0.5]
K, T = 100_000, 250
dev = torch.device("cuda", 0)
eng = TadEngine(0)
key = torch.empty(n, dtype=torch.int64, device=dev)
tend = torch.empty(n, dtype=torch.int64, device=dev)
val = torch.empty(n, dtype=torch.int64, device=dev)
eng.synth(0, n, K, T, into=(key, tend, val))
base = key.clone()
for f in fracs:
    key.copy_(base)
    if f > 0:
        hot = torch.rand(n, device=dev) < f
        key[hot] = 7
    torch.cuda.synchronize()
    for _ in range(2):
        eng.run("EWMA", key, tend, val, K, agg_flow="svc", out="device").close()
    t0 = time.perf_counter()
    for _ in range(5):
        r = eng.run("EWMA", key, tend, val, K, agg_flow="svc", out="device")
        st = r.stats
        r.close()
    dt = (time.perf_counter() - t0) / 5
    print("hot fraction %.2f: %.3f ms/job (meta %.3f, stage0 %.3f [partition %.3f], detect %.3f), points %d" %
          (f, dt * 1e3, st["ms_meta"], st["ms_stage0"], st["ms_scatter"], st["ms_detect"], st["n_points"]))
