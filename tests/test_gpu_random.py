"""GPU: randomized whole-job parity.  Forty seeded configurations drawn over table shape (1 .. 3e5 rows, 1 .. 5000 keys,
1 .. 400 buckets), lattice (step 1 / 7 / 60 / 3600 s, arbitrary origin), aggregation operator, second key column,
rejected rows, time-window filter, values beyond 2^49 / wrap-around sums and Stage-0 strategy; every one must equal the
oracle bit for bit (integers, sigma, EWMA, verdicts)."""

import numpy as np
import pytest

from oracle import tad_oracle as orc

from test_gpu_parity import check_job

pytestmark = pytest.mark.gpu
SKIP = np.uint64(orc.MASK64)


@pytest.mark.parametrize("seed", range(40))
def test_random_job(engine, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 17, 300, 5000, 40000, 300000]))
    K = int(rng.choice([1, 2, 9, 64, 65, 700, 5000]))
    T = int(rng.choice([1, 2, 5, 33, 250, 400]))
    step = int(rng.choice([1, 7, 60, 3600]))
    t0 = int(rng.integers(0, 2_000_000_000))
    algo = str(rng.choice(["EWMA", "DBSCAN"]))
    agg = str(rng.choice(["svc", ""]))          # sum / max
    key = rng.integers(0, K, size=n).astype(np.uint64)
    t = (t0 + step * rng.integers(0, T, size=n)).astype(np.int64)
    v = rng.integers(1, 2**33, size=n).astype(np.uint64)
    if rng.random() < 0.4:
        big = rng.random(n) < 0.05
        v = np.where(big, rng.integers(2**49, 2**64 - 1, size=n, dtype=np.uint64), v)
    kw = {}
    if rng.random() < 0.3:
        key[rng.random(n) < 0.2] = SKIP
    if rng.random() < 0.25:
        k2 = rng.integers(0, K, size=n).astype(np.uint64)
        k2[rng.random(n) < 0.5] = SKIP
        kw["key_id2"] = k2
    if rng.random() < 0.25:
        kw["flow_start_s"] = t - rng.integers(0, 5 * step + 1, size=n)
        kw["start_time"] = int(t0 + step * (T // 4))
        kw["end_time"] = int(t0 + step * max(1, (3 * T) // 4) + 1)
    with engine.plan(stage0=str(rng.choice(["v1", "v2"])), partition_pass=str(rng.choice(["sort", "wc"]))):
        check_job(engine, algo, key, t, v, K, agg_flow=agg, **kw)
