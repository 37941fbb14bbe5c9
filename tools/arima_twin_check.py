"""Debugging aid: host instantiation of the device ARIMA source (tools/arima_twin.cpp) vs oracle/arima_exact.c,
bit for bit, on the reference's golden series and seeded random series.  Run from the repo root after building
/tmp/libarima_twin.so (command in tools/arima_twin.cpp)."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import arima_oracle as ao

tw = ctypes.CDLL("/tmp/libarima_twin.so")
tw.twin_series.restype = ctypes.c_int
tw.twin_series.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def twin(x, maxiter=50):
    x = np.ascontiguousarray(x, dtype=np.float64)
    pred = np.empty(x.size); info = np.zeros(4)
    rc = tw.twin_series(x.ctypes.data, x.size, maxiter, pred.ctypes.data, info.ctypes.data)
    return (pred if rc == 1 else None), info


g = json.load(open("tests/golden/reference_golden.json"))
series = {"golden": np.array(g["throughput_list"], dtype=np.float64)}
rng = np.random.default_rng(7)
for i in range(20):
    n = int(rng.integers(4, 120))
    base = 10 ** rng.uniform(3, 10)
    x = base * np.exp(rng.normal(0, rng.uniform(0.001, 0.6), n))
    if rng.random() < 0.5:
        x[rng.integers(0, n)] *= rng.uniform(2, 12)
    series["rand%d" % i] = np.floor(x) + 1.0
bad = 0
for name, x in series.items():
    t0 = time.time(); c = {}
    a = ao.calculate_arima_exact(x, counters=c); t1 = time.time()
    b, info = twin(x)
    if a is None or b is None:
        same = a is None and b is None
        print(name, "None", same); bad += not same; continue
    a = np.array(a)
    same = bool((a.view(np.uint64) == b.view(np.uint64)).all())
    print(name, len(x), "bit-equal" if same else "DIFF max rel %.3g at %s" % (np.max(np.abs(a - b) / np.abs(a)), np.flatnonzero(a != b)[:5]),
          "steps", c.get("kalman_steps"), int(info[1]), "%.2fs" % (t1 - t0))
    bad += not same
print("mismatching series:", bad)
