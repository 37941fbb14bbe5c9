"""The queued, default-off work at the FULL size of the bench configurations, on the host emulator (development aid — see README.md):
C2 (EWMA, 1e8 rows / 1e5 keys / 250 buckets, the plan of the bench line: 782 partitions, write-combining pass B, sampled histogram)
with k_ewma_fused / the pass-A prefetch / a small LDS capacity, and C4 (DBSCAN, 1e8 rows / 1e6 keys / 100 buckets, max) with the tile
statistics, key rounds + skipped grid columns and the wave list / list emit.  Every emitted row and column against the numpy oracle on
the complete table.  ~7 minutes each on one host core, ~15 GB of host memory.
    python tools/hipemu/build.py && python tools/hipemu/check_fullsize.py [c2|c4]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["TAD_LIBRARY_PATH"] = os.path.join(HERE, "_build", "libtad_hipemu.so")

import numpy as np  # noqa: E402

from oracle import tad_oracle as orc  # noqa: E402
from theia_amd import TadEngine  # noqa: E402

def check_c2():
    t0 = time.time()
    N, K, T = 100_000_000, 100_000, 250
    k, t, v = orc.synth_rows_parallel(N, K, T)
    pk, pt, pv = orc.stage0_dense(k, t, v, "sum", K, orc.SYNTH_T_BASE, orc.SYNTH_T_STEP, T)
    eng = TadEngine(device=0)
    dk, dt, dv = eng.synth(0, N, K, T)
    del k, t, v
    keys, ptr = orc.series_offsets(pk)
    pvf = orc.u64_to_f64(pv)
    sigma, has = orc.stddev_samp_all(pvf, ptr)
    calc = orc.ewma_all(pvf, ptr)
    verdict = np.repeat(has, np.diff(ptr)) & (np.abs(pvf - calc) > np.repeat(sigma, np.diff(ptr)))
    sg_rows = np.repeat(sigma, np.diff(ptr))[verdict]
    print("oracle ready, %d anomalies, %.0f s" % (verdict.sum(), time.time()-t0))
    for label, plan in (("default plan", {}), ("exact histogram", {"histogram": "exact"}), ("staged emit, 64 LDS rows per wavefront", {"ewma_emit_rows": 64}),
                        ("sort-by-tile pass B, lane-per-key emit", {"partition_pass": "sort", "ewma_emit": "lane"})):
        with eng.plan(**plan):
            res = eng.run("EWMA", dk, dt, dv, K, agg_flow="svc", out="device")
        assert res.stats["stage0_path"] == (2 if plan.get("partition_pass") == "sort" else 3) and res.n_rows == int(verdict.sum()), (label, res.n_rows)
        h = res.to_host()
        for f, want in (("key_id", pk[verdict]), ("flow_end_s", pt[verdict]), ("throughput", pvf[verdict]), ("algo_calc", calc[verdict]), ("stddev", sg_rows)):
            assert (h[f] == want).all(), (label, f)
        print("ok full-size C2 %-70s rows %d sampled %d  %.0f s" % (label, res.n_rows, res.stats["hist_sampled"], time.time()-t0))



def check_c4():
    t0 = time.time()
    N, K, T = 100_000_000, 1_000_000, 100
    k, t, v = orc.synth_rows_parallel(N, K, T)
    pk, pt, pv = orc.stage0_dense(k, t, v, "max", K, orc.SYNTH_T_BASE, orc.SYNTH_T_STEP, T)
    eng = TadEngine(device=0)
    dk, dt, dv = eng.synth(0, N, K, T)
    del k, t, v
    keys, ptr = orc.series_offsets(pk)
    pvf = orc.u64_to_f64(pv)
    sigma, has = orc.stddev_samp_all(pvf, ptr)
    noise = orc.dbscan_noise_all(pvf, ptr)
    sg_rows = np.repeat(sigma, np.diff(ptr))[noise]
    print("oracle ready, %d noise points, %.0f s" % (noise.sum(), time.time()-t0))
    for label, plan in (("default plan", {}), ("write-combining pass B forced", {"partition_pass": "wc"})):
        with eng.plan(**plan):
            res = eng.run("DBSCAN", dk, dt, dv, K, agg_flow="", out="device")
        assert res.n_rows == int(noise.sum()), (label, res.n_rows, int(noise.sum()))
        assert res.stats["n_points"] == pk.size and res.stats["n_keys"] == K
        h = res.to_host()
        for f, want in (("key_id", pk[noise]), ("flow_end_s", pt[noise]), ("throughput", pvf[noise]), ("stddev", sg_rows)):
            assert (h[f] == want).all(), (label, f)
        assert (h["algo_calc"] == 0.0).all()
        print("ok full-size C4 %-60s rows %d path %d  %.0f s" % (label, res.n_rows, res.stats["stage0_path"], time.time()-t0))



if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("c2", "all"):
        check_c2()
    if what in ("c4", "all"):
        check_c4()
