#!/usr/bin/env python3
"""Trace of scipy's fmin_l_bfgs_b on one ARIMA(1,1,1) fit (every f/g evaluation), for diffing with
tools/arima_trace.cpp.  usage: arima_trace_scipy.py <golden-index t>  (history = boxcox(golden)[:t])"""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arima_oracle as ao
from scipy import optimize
g = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_golden.json")))
x = np.array(g["throughput_list"], dtype=float)
lam = ao.boxcox_mle_lambda(x); y = ao.boxcox_transform(x, lam)
t = int(sys.argv[1]); h = np.ascontiguousarray(y[:t])
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write("%d\n" % t + "\n".join("%.17g" % v for v in h) + "\n")
x0 = ao.untransform_params(*ao.start_params(h))
print("start %.17g %.17g %.17g" % tuple(x0))
calls = [0]
def obj(u):
    llf, _ = ao.kalman_fast(h, *ao.transform_params(u)); return -llf / t
def fg(u):
    f0 = obj(u); gr = np.zeros(3)
    for i in range(3):
        xe = u.copy(); xe[i] = xe[i] + 1e-5; gr[i] = (obj(xe) - f0) / (xe[i] - u[i])
    calls[0] += 1
    print("eval %d x %.17g %.17g %.17g f %.17g g %.17g %.17g %.17g" % (calls[0], *u, f0, *gr))
    return f0, gr
xo, fv, info = optimize.fmin_l_bfgs_b(fg, x0, m=10, factr=1e7, pgtol=1e-5, maxiter=50, bounds=[(None, None)] * 3,
                                      callback=lambda xk: print("  iter done"))
print("final x %.17g %.17g %.17g f %.17g forecast %.17g nit %d %s" % (*xo, fv, ao.kalman_fast(h, *ao.transform_params(xo))[1], info["nit"], info["task"]))
