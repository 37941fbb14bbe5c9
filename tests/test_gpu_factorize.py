"""GPU: tad_factorize (ingest) — the rows' GROUP BY key tuples -> dense ids in order of first appearance with a
hand-written HBM hash table.  The ids and the key tables must be those of the pandas factorisation (the host path of
theia_amd/anomaly_detection.py:prepare_columns) on every mode / filter case, for plain string columns and for dictionary-encoded
ones; the raw entry point is checked on random tuples, masks, two sides, device-resident columns and the edge cases."""
import numpy as np
import pytest

from oracle import job_oracle as jo
from oracle import tad_oracle as orc
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


def test_millions_of_distinct_keys_on_the_full_size_table(engine):
    """Per-connection keys (mode None): nearly every row is a key of its own.  The full-size table (2 n slots) must take them all — with millions
    of keys at load 0.48 linear-probing clusters longer than the small tables' probe limit (32) are certain, and round 6's 5e7-connection
    ingest failed on exactly that (`tad_factorize: the full-size table filled up`)."""
    rng = np.random.default_rng(99)
    n = 4_000_000
    a = rng.integers(0, 1 << 40, size=n).astype(np.int64)
    b = rng.integers(0, 65536, size=n).astype(np.int64)
    a[n // 2:] = a[:n - n // 2]            # the second half repeats the first half's keys in order: ids 0 .. n/2 - 1 twice
    b[n // 2:] = b[:n - n // 2]
    k1, _, fr = engine.factorize([a, b])
    uniq = n // 2                           # (random 56-bit tuples: no accidental duplicates at this size)
    assert fr.size == uniq and (fr == np.arange(uniq, dtype=np.uint64)).all()
    assert (k1[:uniq] == np.arange(uniq, dtype=np.uint64)).all() and (k1[n // 2:] == k1[:n - n // 2]).all()


# ---- tad_factorize_hist (ABI 12): the key-bin histogram of the ids as a by-product, for the Stage 0 of the job that follows ----
@pytest.mark.parametrize("algo,agg,sides,plan", [
    ("DBSCAN", "", 1, {}), ("DBSCAN", "", 1, {"tile_cells": "wide"}), ("DBSCAN", "", 1, {"partition_pass": "sort"}),
    ("DBSCAN", "", 1, {"partition_pass": "wc_sectors"}), ("DBSCAN", "svc", 1, {}), ("DBSCAN", "", 1, {"histogram": "exact"}),
    ("EWMA", "svc", 1, {}), ("EWMA", "pod", 2, {}), ("DBSCAN", "pod", 2, {})])
def test_job_with_the_factorisation_histogram_equals_the_job_without(engine, algo, agg, sides, plan):
    """tad_run sizes pass B's regions from tad_factorize_hist's by-product instead of reading the key column again: the same rows, bit for bit,
    as the job that counts for itself — for the settle-mode variants of the DBSCAN job (tile cells, partition passes, operators), the EWMA job
    and pod mode's two keys per row; the histogram itself equals numpy's count per (workgroup, bin)."""
    from theia_amd import TadEngine
    rng = np.random.default_rng(17)
    n, card = 4_300_000, 100_000                  # >= 2^22 rows: the partition path on its own; ~3e5 keys x 20 buckets: a dense grid
    raw = [rng.integers(0, card, size=n).astype(np.int64), rng.integers(0, 3, size=n).astype(np.int64)]
    keep = rng.random(n) < 0.9
    rawb = [rng.integers(0, card, size=n).astype(np.int64), rng.integers(0, 3, size=n).astype(np.int64)] if sides == 2 else None
    keepb = (rng.random(n) < 0.8) if sides == 2 else None
    out = engine.factorize(raw, keep, rawb, keepb, with_hist=True)
    k1, k2, first, hist = out
    K = first.size
    assert hist.valid and hist.c.n_rows == n and hist.c.num_keys == K and hist.c.sides == sides and hist.c.workgroups == 256
    bins = np.frombuffer(hist.bins.to_host().tobytes(), dtype=np.uint32)[:256 * hist.c.nbins].reshape(256, hist.c.nbins)
    g = np.arange(n) // hist.c.chunk_rows
    want = np.zeros((256, hist.c.nbins), dtype=np.int64)
    for kk in (k1, k2) if sides == 2 else (k1,):
        live = kk != SKIP
        np.add.at(want, (g[live], (kk[live] >> np.uint64(hist.c.shift)).astype(np.int64)), 1)
    assert (bins == want).all() and hist.c.nbins == -(-K // (1 << hist.c.shift))
    _, t, v = orc.synth_rows(3, n, 1000, 20)
    eng = TadEngine(device=0, plan=plan)
    try:
        a = eng.run(algo, k1, t, v, K, agg_flow=agg, key_id2=k2)
        b = eng.run(algo, k1, t, v, K, agg_flow=agg, key_id2=k2, key_hist=hist)
        assert b.stats["hist_sampled"] == 2 and a.stats["hist_sampled"] in (0, 1) and b.stats["stage0_path"] in (2, 3)
        assert a.n_rows == b.n_rows and a.stats["n_points"] == b.stats["n_points"] and a.stats["rows_used"] == b.stats["rows_used"]
        for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
            assert (a[f] == b[f]).all(), f
        # a STALE histogram — right shape, another batch's counts (here: the key column reversed) — must cost an attempt, never memory or rows:
        # pass B writes nothing past a region, reports it, and the job counts for itself
        kr = np.ascontiguousarray(k1[::-1])
        kr2 = None if k2 is None else np.ascontiguousarray(k2[::-1])
        e1 = eng.run(algo, kr, t, v, K, agg_flow=agg, key_id2=kr2)
        e2 = eng.run(algo, kr, t, v, K, agg_flow=agg, key_id2=kr2, key_hist=hist)
        assert e2.stats["hist_sampled"] != 2 and e2.stats["stage0_attempts"] == e1.stats["stage0_attempts"] + 1 and e1.n_rows == e2.n_rows
        for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
            assert (e1[f] == e2[f]).all(), f
        # a histogram that does not belong to the batch (one row fewer) or a job with a time window: ignored, same rows
        c = eng.run(algo, k1[:-1], t[:-1], v[:-1], K, agg_flow=agg, key_id2=None if k2 is None else k2[:-1], key_hist=hist)
        assert c.stats["hist_sampled"] != 2
        if agg != "pod":
            d = eng.run(algo, k1, t, v, K, agg_flow=agg, end_time=int(t.max()), key_hist=hist)
            assert d.stats["hist_sampled"] != 2 and d.stats["rows_used"] < a.stats["rows_used"]
    finally:
        eng.close()
        hist.free()


def test_sparse_table_through_the_partition_sort_counts_for_itself(engine):
    """A table whose grid would be mostly empty (6e5 keys x 60 buckets for 4.3e6 rows) takes the sparse Stage 0 through pass A / pass B
    (stage0_path 8).  Its LDS sort rounds are planned from the histogram with sizes it relies on, so a caller's histogram is not used
    there: the job redoes pass A with its own count (one more attempt) and gives the same rows."""
    rng = np.random.default_rng(18)
    n = 4_300_000
    raw = [rng.integers(0, 200_000, size=n).astype(np.int64), rng.integers(0, 3, size=n).astype(np.int64)]
    k1, _, first, hist = engine.factorize(raw, rng.random(n) < 0.9, with_hist=True)
    _, t, v = orc.synth_rows(3, n, 1000, 60)
    a = engine.run("EWMA", k1, t, v, first.size, agg_flow="svc")
    b = engine.run("EWMA", k1, t, v, first.size, agg_flow="svc", key_hist=hist)
    assert a.stats["stage0_path"] == b.stats["stage0_path"] == 8 and b.stats["hist_sampled"] == 0 and a.stats["hist_sampled"] == 0
    assert a.stats["stage0_attempts"] == 1 and b.stats["stage0_attempts"] == 2
    assert a.n_rows == b.n_rows > 0 and a.stats["n_points"] == b.stats["n_points"]
    for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
        assert (a[f] == b[f]).all(), f
    hist.free()


def test_factorisation_histogram_of_a_small_or_empty_batch(engine):
    k1, _, first, hist = engine.factorize([np.arange(1000, dtype=np.int64) % 7], with_hist=True)
    assert first.size == 7 and hist.valid and hist.c.nbins == 7 and hist.c.shift == 0      # (a job this small never reads it: no pass A below 2^22 rows)
    _, t, v = orc.synth_rows(0, 1000, 7, 10)
    assert engine.run("EWMA", k1, t, v, 7, agg_flow="svc", key_hist=hist).stats["hist_sampled"] == 0
    k1, _, first, hist2 = engine.factorize([np.zeros(10, dtype=np.int64)], np.zeros(10, bool), with_hist=True)
    assert first.size == 0 and not hist2.valid
    hist.free(); hist2.free()


# ---- tad_encode_strings (ABI 10): an Arrow string column -> dictionary codes in order of first appearance ----
def _want_codes(arr):
    """pyarrow's own dictionary encode (first-appearance order), nulls as ''."""
    import pyarrow as pa
    import pyarrow.compute as pc
    a = arr.cast(pa.string()) if not (pa.types.is_string(arr.type) or pa.types.is_large_string(arr.type)) else arr
    d = pc.dictionary_encode(a.fill_null(""))
    codes = d.indices.to_numpy(zero_copy_only=False).astype(np.int64)
    first = np.full(len(d.dictionary), -1, np.int64)
    for i in range(codes.size - 1, -1, -1):
        first[codes[i]] = i
    return codes, first.astype(np.uint64), d.dictionary.to_pylist()


def _random_strings(rng, n, distinct, max_len=40):
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789-./:{}", dtype=np.uint8)
    vocab = []
    for i in range(distinct):
        ln = int(rng.integers(0, max_len + 1))
        vocab.append(bytes(rng.choice(alphabet, ln)).decode() + ("" if ln == 0 else "#%d" % i))     # distinct by construction (one may be "")
    return [vocab[j] for j in rng.integers(0, distinct, n)]


@pytest.mark.parametrize("kind", ["string", "large_string", "binary"])
@pytest.mark.parametrize("n,distinct", [(1, 1), (7, 3), (5000, 40), (20000, 7000)])
def test_encode_strings_equals_arrows_dictionary_encode(engine, kind, n, distinct):
    import pyarrow as pa
    rng = np.random.default_rng(n * 31 + distinct)
    vals = _random_strings(rng, n, distinct)
    t = {"string": pa.string(), "large_string": pa.large_string(), "binary": pa.binary()}[kind]
    arr = pa.array([v.encode() for v in vals] if kind == "binary" else vals, t)
    codes, first = engine.encode_strings(arr)
    want_codes, want_first, dictionary = _want_codes(arr)
    assert codes.dtype == np.int64 and (codes == want_codes).all()
    assert (first == want_first).all()
    # the dictionary the host builds from the first rows is Arrow's, value for value
    got_dict = arr.take(pa.array(first.astype(np.int64))).to_pylist()
    assert [g.decode() if isinstance(g, bytes) else g for g in got_dict] == dictionary


def test_encode_strings_slices_nulls_and_strings_that_differ_only_in_their_tail(engine):
    import pyarrow as pa
    base = ["pod-%s" % ("x" * k) for k in range(0, 20)] + ["", None, "a", "ab", "abc", "abcdefgh", "abcdefghi", "abcdefgh\0", "abcdefg"]
    vals = (base * 7)[3:]
    arr = pa.array(vals, pa.string())
    for sl in (arr, arr.slice(5), arr.slice(11, 60), arr.slice(len(arr) - 1)):
        codes, first = engine.encode_strings(sl)
        want_codes, want_first, _ = _want_codes(sl)
        assert (codes == want_codes).all() and (first == want_first).all()
    # a chunked column is one column
    chunked = pa.chunked_array([arr.slice(0, 40), arr.slice(40, 3), arr.slice(43)])
    codes, first = engine.encode_strings(chunked)
    want_codes, want_first, _ = _want_codes(arr)
    assert (codes == want_codes).all() and (first == want_first).all()
    # an all-null / all-empty column: one value
    codes, first = engine.encode_strings(pa.array([None, None, ""], pa.string()))
    assert codes.tolist() == [0, 0, 0] and first.tolist() == [0]
    # no rows
    codes, first = engine.encode_strings(pa.array([], pa.string()))
    assert codes.size == 0 and first.size == 0


def test_encode_strings_grows_its_table_for_a_high_cardinality_column(engine):
    """More distinct values than half the small (2^20-slot) table: the first attempt gives up on the device, the second uses 2 n slots."""
    import pyarrow as pa
    n = 1_300_000
    ids = np.arange(n) % 700_000
    rng = np.random.default_rng(3)
    rng.shuffle(ids)
    arr = pa.array(np.char.add("k", ids.astype(str)))
    codes, first = engine.encode_strings(arr)
    want_codes, want_first, _ = _want_codes(arr)
    assert (codes == want_codes).all() and (first == want_first).all() and first.size == 700_000


def test_encode_strings_on_device_resident_buffers_and_bad_offsets(engine):
    import pyarrow as pa
    from theia_amd import TadError
    rng = np.random.default_rng(9)
    vals = _random_strings(rng, 30000, 500)
    arr = pa.array(vals, pa.string())
    _, obuf, dbuf = arr.buffers()
    offsets = np.frombuffer(obuf, dtype=np.int32)[: len(arr) + 1].copy()
    data = np.frombuffer(dbuf, dtype=np.uint8).copy()
    want_codes, want_first, _ = _want_codes(arr)
    # host arrays
    codes, first = engine.encode_strings((offsets, data))
    assert (codes == want_codes).all() and (first == want_first).all()
    # device arrays (the bytes padded to a whole number of 8-byte elements)
    d_off = DeviceArray.from_host(engine, offsets)
    d_data = DeviceArray.from_host(engine, np.concatenate([data, np.zeros(-data.size % 8, np.uint8)]))
    dc, df = engine.encode_strings((d_off, d_data))
    assert (dc.to_host() == want_codes).all() and (df.to_host() == want_first).all()
    # int64 offsets
    codes, first = engine.encode_strings((offsets.astype(np.int64), data))
    assert (codes == want_codes).all()
    # malformed offsets are refused, not read
    bad = offsets.copy(); bad[100] = bad[101] + 5
    with pytest.raises(TadError):
        engine.encode_strings((bad, data))
    bad = offsets.copy(); bad[-1] = data.size + 64
    with pytest.raises(TadError):
        engine.encode_strings((bad, data))
    with pytest.raises(TadError):
        engine.encode_strings(pa.array([1, 2, 3]))


@pytest.mark.parametrize("n,keys", [(1_300_000, 700_000), (19_000_000, 9_000_000)], ids=["2^20->2^24", "2^24->full"])
def test_factorize_grows_its_table_when_the_device_asks(engine, n, keys):
    """The table starts at 2^20 slots and climbs 2^24 -> 2 n when a pass finds it filling up (tad_factorize.hip): ids and first rows must not
    depend on how many attempts it took."""
    import pandas as pd
    rng = np.random.default_rng(keys)
    a = rng.integers(0, keys, size=n).astype(np.int64)
    b = (a * 7 + 3) % 1000
    k1, _, first = engine.factorize([a, b])
    codes, uniq = pd.factorize(a)
    assert (k1 == codes.astype(np.uint64)).all()
    assert first.size == uniq.size and (a[first.astype(np.int64)] == uniq).all()
    assert (np.diff(first.astype(np.int64)) > 0).all()                 # order of first appearance


def test_encode_strings_long_strings_take_the_global_path_and_mixed_blocks_both(engine):
    """A block of 256 rows stages its bytes in LDS when they fit 24 KB; labels of hundreds of bytes do not, and a column may mix both kinds of
    block.  Null rows whose bytes are still in the buffer (validity cleared after the fact) read as ''."""
    import pyarrow as pa
    rng = np.random.default_rng(21)
    longs = ['{"app":"%s","tier":"%s"}' % ("x" * int(rng.integers(150, 400)), "y" * int(rng.integers(0, 90))) + str(i) for i in range(300)]
    vals = [longs[j] for j in rng.integers(0, 300, 3000)]
    arr = pa.array(vals, pa.string())
    codes, first = engine.encode_strings(arr)
    want_codes, want_first, _ = _want_codes(arr)
    assert (codes == want_codes).all() and (first == want_first).all()
    # short rows with a huge one every ~700 rows: some blocks staged, some not; the same strings must meet in the same slots
    mixed = [("p%d" % (i % 37)) if i % 701 else ("L" * 30000 + str(i % 3)) for i in range(6000)]
    arr = pa.array(mixed, pa.large_string())
    codes, first = engine.encode_strings(arr)
    want_codes, want_first, _ = _want_codes(arr)
    assert (codes == want_codes).all() and (first == want_first).all()
    # nulls that still own bytes
    base = pa.array(["aa", "bbb", "aa", "cccc", "bbb", "dd"] * 50, pa.string())
    _, obuf, dbuf = base.buffers()
    valid = np.packbits(np.array([i % 4 != 1 for i in range(len(base))], dtype=np.uint8), bitorder="little")
    holed = pa.Array.from_buffers(pa.string(), len(base), [pa.py_buffer(valid.tobytes()), obuf, dbuf])
    assert holed.null_count == len(base) // 4
    codes, first = engine.encode_strings(holed)
    want_codes, want_first, _ = _want_codes(holed)
    assert (codes == want_codes).all() and (first == want_first).all()
