"""GPU: tad_factorize (ingest) — the rows' GROUP BY key tuples -> dense ids in order of first appearance with a
hand-written HBM hash table.  The ids and the key tables must be those of the pandas factorisation (the host path of
theia_amd/anomaly_detection.py:prepare_columns) on every mode / filter case, for plain string columns and for dictionary-encoded
ones; the raw entry point is checked on random tuples, masks, two sides, device-resident columns and the edge cases."""
import numpy as np
import pytest

from oracle import job_oracle as jo
from theia_amd import _capi as capi
from theia_amd import anomaly_detection as ad
from theia_amd.engine import DeviceArray
from test_host_job import CASES

pytestmark = pytest.mark.gpu
SKIP = np.uint64(capi.TAD_KEY_SKIP)


def pandas_ids(cols, keep, cols_b=None, keep_b=None):
    """ids in order of first appearance over [kept rows of side a ++ kept rows of side b] (+ the first virtual rows)"""
    import pandas as pd
    n = len(cols[0])
    sides = [(cols, np.ones(n, bool) if keep is None else np.asarray(keep, bool), 0)]
    if cols_b is not None:
        sides.append((cols_b, np.ones(n, bool) if keep_b is None else np.asarray(keep_b, bool), 1))
    parts, vrows = [], []
    for cs, kp, side in sides:
        sel = np.flatnonzero(kp)
        parts.append([np.asarray(c)[sel] for c in cs] + [np.full(sel.size, side)])
        vrows.append(sel + side * n)
    cat = [np.concatenate([p[i] for p in parts]) for i in range(len(cols) + 1)]
    vrow = np.concatenate(vrows)
    if vrow.size == 0:
        return [np.full(n, SKIP) for _ in sides], np.zeros(0, np.uint64)
    codes, _ = pd.MultiIndex.from_arrays(cat).factorize()
    out, at = [], 0
    for cs, kp, side in sides:
        k = np.full(n, SKIP, dtype=np.uint64)
        m = int(kp.sum())
        k[np.flatnonzero(kp)] = codes[at:at + m].astype(np.uint64)
        at += m
        out.append(k)
    first = np.full(codes.max() + 1, -1, dtype=np.int64)
    for i in range(codes.size - 1, -1, -1):
        first[codes[i]] = vrow[i]
    return out, first.astype(np.uint64)


@pytest.mark.parametrize("dict_encoded", [False, True], ids=["strings", "dictionaries"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()))
def test_prepare_columns_on_the_gpu_equals_the_pandas_path(engine, case, dict_encoded):
    flows = jo.synth_flows(5000)
    if dict_encoded:
        enc = {}
        for name, col in flows.items():
            a = np.asarray(col)
            if a.dtype.kind in "US":
                values, codes = np.unique(a.astype(str), return_inverse=True)
                perm = np.random.default_rng(4).permutation(values.size)
                inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
                enc[name] = ad.DictColumn(inv[codes], values[perm])
            else:
                enc[name] = a
        flows = enc
    a = ad.prepare_columns(flows, **case)
    b = ad.prepare_columns(flows, **case, engine=engine)
    assert a.mode == b.mode and a.num_keys == b.num_keys and (a.start_time, a.end_time) == (b.start_time, b.end_time)
    assert (a.key_id == b.key_id).all()
    assert (a.key_id2 is None) == (b.key_id2 is None) and (a.key_id2 is None or (a.key_id2 == b.key_id2).all())
    assert set(a.key_table) == set(b.key_table)
    for name in a.key_table:
        assert (np.asarray(a.key_table[name]).astype(str) == np.asarray(b.key_table[name]).astype(str)).all(), name


@pytest.mark.parametrize("ncols,n,card", [(1, 100_000, 500), (3, 200_000, 40), (6, 150_000, 12), (8, 50_000, 5), (2, 300_000, 100_000)])
def test_random_tuples_masks_and_two_sides(engine, ncols, n, card):
    rng = np.random.default_rng(ncols * 1000 + n)
    span = np.array([card, 7, 3, 5, 2, 4, 2, 3][:ncols])
    cols = [(rng.integers(0, span[c], size=n) * (1 if c % 2 == 0 else -977) + (c << 40)).astype(np.int64) for c in range(ncols)]   # negative and large values
    keep = rng.random(n) < 0.8
    for kp in (None, keep):
        want, first = pandas_ids(cols, kp)
        k1, k2, fr = engine.factorize(cols, kp)
        assert k2 is None and (k1 == want[0]).all() and (fr == first).all()
    colsb = [c[rng.permutation(n)] for c in cols]
    keepb = rng.random(n) < 0.5
    want, first = pandas_ids(cols, keep, colsb, keepb)
    k1, k2, fr = engine.factorize(cols, keep, colsb, keepb)
    assert (k1 == want[0]).all() and (k2 == want[1]).all() and (fr == first).all()
    # device-resident columns: same ids
    dk1, dk2, dfr = engine.factorize([DeviceArray.from_host(engine, c) for c in cols], keep, [DeviceArray.from_host(engine, c) for c in colsb], keepb)
    assert (dk1.to_host() == want[0]).all() and (dk2.to_host() == want[1]).all() and (dfr.to_host() == first).all()


def test_edge_cases(engine):
    from theia_amd import TadError
    one = np.array([7], dtype=np.int64)
    k1, _, fr = engine.factorize([one])
    assert k1.tolist() == [0] and fr.tolist() == [0]
    k1, _, fr = engine.factorize([np.zeros(1000, np.int64)], np.zeros(1000, bool))          # nothing kept
    assert (k1 == SKIP).all() and fr.size == 0
    same = np.full(70_000, -5, dtype=np.int64)                                             # one key, every lane of every wavefront on one slot
    k1, _, fr = engine.factorize([same, same])
    assert (k1 == 0).all() and fr.tolist() == [0]
    alld = np.arange(200_000, dtype=np.int64)[::-1].copy()                                   # all distinct: id = row
    k1, _, fr = engine.factorize([alld])
    assert (k1 == np.arange(200_000, dtype=np.uint64)).all() and (fr == np.arange(200_000, dtype=np.uint64)).all()
    k1, _, fr = engine.factorize([alld], max_keys=10)                                       # the key table may be capped; the ids are not
    assert fr.tolist() == list(range(10)) and int(k1.max()) == 199_999
    with pytest.raises(TadError):
        engine.factorize([alld] * 9)
    with pytest.raises(TadError):
        engine.factorize([alld], None, [alld, alld])
