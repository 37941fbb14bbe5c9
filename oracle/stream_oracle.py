"""Streaming-EWMA oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The append-only form of the reference's EWMA detector (include/tad.h: tad_run_stream).  The reference has no
streaming job; what pins this file is the batch semantics it must converge to: after the last batch the per-key state
(n, avg, m2, ewma) must equal, bit for bit, Spark's streaming stddev_samp update (SURVEY.md appendix A.2,
anomaly_detection.py:674-684) and calculate_ewma (:146-165) over the concatenated series — tests/test_gpu_stream.py checks
both the GPU and this oracle against oracle/tad_oracle.py for that.  Verdicts use the RUNNING sigma (points seen so far)."""
import numpy as np

from . import tad_oracle as orc


class StreamState:
    def __init__(self, num_keys):
        self.n = np.zeros(num_keys, dtype=np.uint32)
        self.avg = np.zeros(num_keys)
        self.m2 = np.zeros(num_keys)
        self.ewma = np.zeros(num_keys)
        self.last_t = np.zeros(num_keys, dtype=np.int64)
        self.seen = np.zeros(num_keys, dtype=bool)


def run_stream(state, key_id, flow_end_s, value, op="sum", alpha=0.5):
    """One batch.  Returns the emitted rows (dict of arrays ordered by key, time) and updates `state` in place.
    Raises ValueError (state untouched) when a point is not newer than its key's last_t."""
    pk, pt, pv = orc.stage0(key_id, flow_end_s, value, op)
    keys, ptr = orc.series_offsets(pk)
    xf = orc.u64_to_f64(pv)
    for k, a in zip(keys.tolist(), ptr[:-1].tolist()):
        if state.seen[k] and pt[a] <= state.last_t[k]:
            raise ValueError("late row for key %d" % k)
    rows = {f: [] for f in ("key_id", "flow_end_s", "throughput", "algo_calc", "stddev")}
    for k, a, b in zip(keys.tolist(), ptr[:-1].tolist(), ptr[1:].tolist()):
        n, cnt = int(state.n[k]), float(state.n[k])
        avg, m2, e = float(state.avg[k]), float(state.m2[k]), float(state.ewma[k])
        for i in range(a, b):
            x = float(xf[i])
            cnt = cnt + 1.0
            n += 1
            d = x - avg
            dn = d / cnt
            avg = avg + dn
            m2 = m2 + d * (d - dn)
            e = (1 - alpha) * e + alpha * x
            if n >= 2:
                sg = float(np.sqrt(np.float64(m2 / (cnt - 1.0))))
                if abs(x - e) > sg:
                    for f, val in zip(rows, (k, int(pt[i]), x, e, sg)):
                        rows[f].append(val)
        state.n[k], state.avg[k], state.m2[k], state.ewma[k] = n, avg, m2, e
        state.last_t[k], state.seen[k] = pt[b - 1], True
    return {"key_id": np.array(rows["key_id"], dtype=np.uint64), "flow_end_s": np.array(rows["flow_end_s"], dtype=np.int64),
            "throughput": np.array(rows["throughput"]), "algo_calc": np.array(rows["algo_calc"]), "stddev": np.array(rows["stddev"])}
