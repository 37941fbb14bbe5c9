"""GPU: streaming EWMA (tad_run_stream).  The state after the last batch must equal the BATCH job's per-key statistics bit
for bit (that is what ties the streaming form to the reference-pinned batch semantics); rows and state must equal the
streaming oracle batch by batch; a late row is rejected and leaves the state untouched."""
import numpy as np
import pytest

from oracle import stream_oracle as so
from oracle import tad_oracle as orc
from theia_amd import TadError

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rows,K,T,cuts", [(60000, 200, 120, (40, 80)), (500000, 3000, 250, (50, 51, 200)), (3000, 7, 64, (1, 2, 3, 60))])
def test_batches_converge_to_the_batch_job(engine, n_rows, K, T, cuts):
    k, t, v = orc.synth_rows(0, n_rows, K, T)
    bucket = (t - orc.SYNTH_T_BASE) // orc.SYNTH_T_STEP
    edges = (0,) + tuple(cuts) + (T,)
    st = engine.state_create(K)
    ost = so.StreamState(K)
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = (bucket >= lo) & (bucket < hi)
        got = engine.run_stream(st, k[sel], t[sel], v[sel], agg_flow="svc")
        want = so.run_stream(ost, k[sel], t[sel], v[sel], "sum")
        assert got.n_rows == want["key_id"].size
        for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev"):
            assert (got[f] == want[f]).all(), (lo, hi, f)
    state = st.export()
    for f in ("n", "avg", "m2", "ewma", "last_t"):
        assert (state[f] == getattr(ost, f)).all(), f
    # ... and equals what the batch job computes over the whole table
    pk, pt, pv = orc.stage0(k, t, v, "sum")
    keys, ptr = orc.series_offsets(pk)
    xf = orc.u64_to_f64(pv)
    sigma, has = orc.stddev_samp_all(xf, ptr)
    ew = orc.ewma_all(xf, ptr)
    kk = keys.astype(np.int64)
    n = state["n"][kk].astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        stream_sigma = np.sqrt(state["m2"][kk] / (n - 1.0))
    assert (np.diff(ptr) == state["n"][kk]).all()
    assert (stream_sigma[has] == sigma[has]).all()
    assert (state["ewma"][kk] == ew[ptr[1:] - 1]).all()
    st.close()


def test_late_row_is_rejected_and_state_kept(engine):
    k, t, v = orc.synth_rows(0, 5000, 20, 50)
    st = engine.state_create(20)
    engine.run_stream(st, k, t, v, agg_flow="svc")
    before = st.export()
    with pytest.raises(TadError) as ei:
        engine.run_stream(st, k[:10], t[:10], v[:10], agg_flow="svc")        # same timestamps again
    assert ei.value.code == -1 and "not newer" in ei.value.message
    after = st.export()
    for f in before:
        assert (before[f] == after[f]).all()
    # an empty batch changes nothing either
    engine.run_stream(st, k[:0], t[:0], v[:0], agg_flow="svc")
    assert all((st.export()[f] == before[f]).all() for f in before)
    # newer rows are fine
    res = engine.run_stream(st, k[:10], t[:10] + 60 * 100, v[:10], agg_flow="svc", emit_all=True)
    assert res.n_rows == np.unique(np.stack([k[:10].astype(np.int64), t[:10]]), axis=1).shape[1]
    st.close()


def test_batch_must_declare_the_states_key_space(engine):
    # a batch that declared fewer keys than the state holds would flip the double buffer with the other keys' candidate
    # state unwritten (tad.h: cols->num_keys must equal the state's num_keys)
    k, t, v = orc.synth_rows(0, 5000, 20, 50)
    st = engine.state_create(20)
    engine.run_stream(st, k, t, v, agg_flow="svc")
    before = st.export()
    for bad in (10, 21):
        with pytest.raises(TadError) as ei:
            engine.run_stream(st, k[:0], t[:0], v[:0], agg_flow="svc", num_keys=bad)
        assert ei.value.code == -1 and "must be equal" in ei.value.message
    assert all((st.export()[f] == before[f]).all() for f in before)
    st.close()
