"""Multi-GPU host logic of the Throughput Anomaly Detection path: one process per GPU, key-sharded.

Why this shape: after Stage 0 everything is independent per flow key — sigma is per
key (anomaly_detection.py:674-684) and every detector runs on one key's series (:440-484) — and the
Stage-0 aggregates are associative integers.  So flow keys are hash-partitioned across the GPUs of a
node, each rank runs the whole single-GPU job (tad_run) on the rows of ITS keys, and the data path
needs no collective.  What does cross ranks (RCCL over xGMI with backend "nccl", gloo on CPU):

  1. ONE all-gather of 9 doubles per rank per job: the counters {anomalies, keys, points, rows_used,
     keys_no_result, rows_in}, summed on every rank — the reference's global `ret_plot.count() == 0`
     decision for the sentinel row (:395) needs the global anomaly count —
  2. and, in the same message, the (n, mean, M2) moments of the shard's aggregated points, Chan-merged in
     rank order on every rank — the job-wide mean / sigma telemetry BASELINE.json's north_star asks for
     (the reference itself has no global sigma);
  3. (only when rows arrive row-sharded instead of key-sharded) one all-to-all(v) of rows or partial
     aggregates to the key owners: `exchange_rows`.  Re-aggregating partial sums / maxima with the
     same operator is bit-exact (wrapping add and unsigned max are associative and commutative).

Anomaly rows stay on the rank that produced them: each rank appends its own rows to `tadetector`
(row order in that table is irrelevant, it is ORDER BY flowStartSeconds — create_table.sh:384); only
the sentinel row is written by rank 0.  Nothing here computes detector numbers: `run_local` is the
HIP engine (TadEngine.run); the tests substitute the oracle for it on CPU ranks.
"""
import numpy as np

SKIP = np.uint64(0xFFFFFFFFFFFFFFFF)  # TAD_KEY_SKIP
STAT_FIELDS = ("n_anomalies", "n_keys", "n_points", "rows_used", "keys_no_result", "rows_in")


def collectives_on(world):
    """Whether the collectives run: always with more than one rank, and with ONE rank when a process group exists — a
    one-rank RCCL group executes the same all-gather / all-to-all(v) calls on device tensors, which is how a single-GPU box
    exercises this module's RCCL path (tests/test_gpu_multirank.py::test_rccl_one_rank_group...)."""
    if world > 1:
        return True
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def owner_of(key_id, world):
    """Rank that owns a key.  Key ids are dense dictionary codes (first-appearance order on the host), so
    `id mod world` is already a uniform hash partition and keeps the local ids dense: local = id // world."""
    return np.asarray(key_id, dtype=np.uint64) % np.uint64(world)


def local_key(key_id, world):
    return np.asarray(key_id, dtype=np.uint64) // np.uint64(world)


def global_key(local_id, rank, world):
    return np.asarray(local_id, dtype=np.uint64) * np.uint64(world) + np.uint64(rank)


def num_local_keys(num_keys, rank, world):
    return (int(num_keys) - rank + world - 1) // world if num_keys > rank else 0


def shard_rows(rank, world, key_id, flow_end_s, value, key_id2=None, flow_start_s=None):
    """Rows of the batch that rank `rank` must see, with LOCAL key ids.  A pod-mode row (two keys,
    anomaly_detection.py:556-565) goes to the owner of each of its keys; the key the rank does not own is
    masked with TAD_KEY_SKIP so that it is aggregated exactly once job-wide."""
    key_id = np.asarray(key_id, dtype=np.uint64)
    live1 = key_id != SKIP
    mine1 = live1 & (owner_of(key_id, world) == rank)
    if key_id2 is not None:
        key_id2 = np.asarray(key_id2, dtype=np.uint64)
        live2 = key_id2 != SKIP
        mine2 = live2 & (owner_of(key_id2, world) == rank)
    else:
        mine2 = np.zeros(key_id.shape, dtype=bool)
    sel = np.flatnonzero(mine1 | mine2)
    out = {
        "key_id": np.where(mine1[sel], local_key(key_id[sel], world), SKIP),
        "flow_end_s": np.asarray(flow_end_s, dtype=np.int64)[sel],
        "value": np.asarray(value, dtype=np.uint64)[sel],
        "key_id2": None, "flow_start_s": None,
    }
    if key_id2 is not None:
        out["key_id2"] = np.where(mine2[sel], local_key(key_id2[sel], world), SKIP)
    if flow_start_s is not None:
        out["flow_start_s"] = np.asarray(flow_start_s, dtype=np.int64)[sel]
    return out


def chan_merge(parts):
    """[(n, mean, M2), ...] merged left to right (Chan et al.) -> (n, mean, M2).  Same formula and order as
    the engine's k_moments / host merge, so every rank computes identical bits."""
    mn, mean, m2 = 0.0, 0.0, 0.0
    for bn, bmean, bm2 in parts:
        if bn == 0:
            continue
        if mn == 0:
            mn, mean, m2 = float(bn), float(bmean), float(bm2)
            continue
        nn = mn + bn
        d = bmean - mean
        mean = mean + d * (bn / nn)
        m2 = m2 + bm2 + d * d * (mn * bn / nn)
        mn = nn
    return mn, mean, m2


class JobReducer:
    """Pre-allocated buffers for the per-job collective (so that a bench step allocates nothing).

    One all-gather of 9 doubles per rank — the six counters (exact in float64 below 2^53) and the shard's
    (n, mean, M2) — replaces an all-reduce plus an all-gather: per-job collectives are latency-bound (a C2 job
    takes under 2 ms), so one hop instead of two.  Every rank then sums the counters and Chan-merges the moments in
    rank order, which gives identical bits everywhere.

    `start(stats)` launches the all-gather asynchronously and returns a handle; `handle.result()` waits for it.  A host
    that runs jobs back to back starts job i's collective, runs job i+1 on the engine's stream, then collects job i's
    result: the collective's latency hides behind the next job (two buffer sets alternate; a handle must be collected
    before the second start after it).  `reduce(stats)` = start + result."""

    WIDTH = len(STAT_FIELDS) + 3

    class Pending:
        def __init__(self, owner, slot, work, event=None):
            self.owner, self.slot, self.work, self.event = owner, slot, work, event

        def result(self):
            o = self.owner
            if self.event is not None:      # device path: the gathered rows are already on their way to pinned host memory
                self.event.synchronize()
                rows = o._host_np[self.slot].reshape(o.world, o.WIDTH).tolist()
            else:
                if self.work is not None:
                    self.work.wait()
                if o.collective:
                    rows = o.gathered[self.slot].view(o.world, o.WIDTH).tolist()
                else:
                    rows = [o.payload[self.slot].tolist()]
            nf = len(STAT_FIELDS)
            out = {f: int(sum(int(r[i]) for r in rows)) for i, f in enumerate(STAT_FIELDS)}
            n, mean, m2 = chan_merge([tuple(r[nf:nf + 3]) for r in rows])
            out["global_mean"] = mean if n > 0 else None
            out["global_sigma"] = (m2 / (n - 1.0)) ** 0.5 if n > 1 else None
            out["write_sentinel"] = out["n_anomalies"] == 0 and o.rank == 0     # anomaly_detection.py:395-420
            return out

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = collectives_on(self.world)
        on_gpu = device is not None and str(device) != "cpu"
        self.payload = [torch.zeros(self.WIDTH, dtype=torch.float64, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(self.world * self.WIDTH, dtype=torch.float64, device=device) for _ in range(2)]
        self._stage = [torch.zeros(self.WIDTH, dtype=torch.float64).pin_memory() for _ in range(2)] if on_gpu else None
        # device path: nothing of the per-job collective waits on the host — the payload goes up from pinned memory, the gathered
        # rows come down into pinned memory right behind the all-gather (stream-ordered), result() waits for an event that has
        # normally long fired (the next job ran in between).  Measured with a one-rank RCCL group: 75 us of host work per job before.
        self._stage_np = [t.numpy() for t in self._stage] if on_gpu else None
        self._host = [torch.zeros(self.world * self.WIDTH, dtype=torch.float64).pin_memory() for _ in range(2)] if on_gpu else None
        self._host_np = [t.numpy() for t in self._host] if on_gpu else None
        self._events = [torch.cuda.Event() for _ in range(2)] if on_gpu else None
        self._next = 0

    def start(self, stats):
        """stats: tad_stats of this rank's run -> Pending (the all-gather is in flight)."""
        torch, dist = self.torch, self.dist
        slot, self._next = self._next, self._next ^ 1
        vals = [float(int(stats.get(f, 0))) for f in STAT_FIELDS] + \
               [float(stats.get("n_points", 0)), float(stats.get("pts_mean", 0.0)), float(stats.get("pts_m2", 0.0))]
        if self._stage is not None:
            self._stage_np[slot][:] = vals
            self.payload[slot].copy_(self._stage[slot], non_blocking=True)
        else:
            self.payload[slot].copy_(torch.tensor(vals, dtype=torch.float64))
        work = None
        if self.collective:
            work = dist.all_gather_into_tensor(self.gathered[slot], self.payload[slot], group=self.group, async_op=True)
        if self._host is not None:
            if work is not None:
                work.wait()              # (RCCL: orders the current stream behind the collective; the host does not block)
            self._host[slot].copy_(self.gathered[slot] if self.collective else self.payload[slot], non_blocking=True)
            self._events[slot].record()
            return JobReducer.Pending(self, slot, None, self._events[slot])
        return JobReducer.Pending(self, slot, work)

    def reduce(self, stats):
        """stats: tad_stats of this rank's run -> dict of job-wide values (identical on every rank)."""
        return self.start(stats).result()


def exchange_rows(cols, world, rank, group=None, device=None):
    """Row-sharded ingest: every rank holds an arbitrary slice of the rows; ship each row (or each locally
    pre-aggregated partial point) to the owner(s) of its key(s) with one all-to-all(v).  Returns the columns this rank
    owns, with local key ids.  Payload: 3 (or 4, 5) int64 columns per row, bucketed by destination rank."""
    import torch
    import torch.distributed as dist
    names = ["key_id", "flow_end_s", "value"] + [n for n in ("key_id2", "flow_start_s") if cols.get(n) is not None]
    send_parts, send_counts = [], []
    for dst in range(world):
        part = shard_rows(dst, world, cols["key_id"], cols["flow_end_s"], cols["value"], cols.get("key_id2"), cols.get("flow_start_s"))
        mat = np.stack([np.asarray(part[n]).view(np.int64) if part[n].dtype == np.uint64 else np.asarray(part[n], dtype=np.int64)
                        for n in names], axis=1) if part["key_id"].size else np.zeros((0, len(names)), dtype=np.int64)
        send_parts.append(mat)
        send_counts.append(mat.shape[0])
    send = torch.from_numpy(np.concatenate(send_parts, axis=0)).to(device or "cpu")
    counts = torch.tensor(send_counts, dtype=torch.int64, device=device or "cpu")
    recv_counts = torch.zeros(world, dtype=torch.int64, device=device or "cpu")
    if collectives_on(world):
        dist.all_to_all_single(recv_counts, counts, group=group)
    else:
        recv_counts.copy_(counts)
    rc = [int(c) for c in recv_counts.tolist()]
    recv = torch.zeros((sum(rc), len(names)), dtype=torch.int64, device=device or "cpu")
    if collectives_on(world):
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=send_counts, group=group)
    else:
        recv.copy_(send)
    got = recv.cpu().numpy()
    out = {"key_id2": None, "flow_start_s": None}
    for j, n in enumerate(names):
        col = np.ascontiguousarray(got[:, j])
        out[n] = col.view(np.uint64) if n in ("key_id", "key_id2", "value") else col
    return out


def exchange_points_torch(key, flow_end_s, value, world, rank, group=None):
    """The same exchange on torch tensors (CPU with gloo, HBM with RCCL): int64 tensors of global key ids, times and
    values (uint64 bit patterns) — typically the partial points a rank pre-aggregated with tad_aggregate.  Rows are
    bucketed by owner (key mod world) with a stable sort, the per-destination counts travel first, then ONE
    all-to-all(v) moves the [n, 3] payload.  Returns (local_key, flow_end_s, value) tensors of the rows this rank owns."""
    import torch
    import torch.distributed as dist
    owner = torch.remainder(key, world)
    order = torch.argsort(owner, stable=True)
    payload = torch.stack([torch.div(key, world, rounding_mode="floor"), flow_end_s, value], dim=1)[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.zeros_like(send_counts)
    if collectives_on(world):
        dist.all_to_all_single(recv_counts, send_counts, group=group)
    else:
        recv_counts.copy_(send_counts)
    sc, rc = [int(c) for c in send_counts.tolist()], [int(c) for c in recv_counts.tolist()]
    recv = torch.empty((sum(rc), 3), dtype=torch.int64, device=payload.device)
    if collectives_on(world):
        dist.all_to_all_single(recv, payload, output_split_sizes=rc, input_split_sizes=sc, group=group)
    else:
        recv.copy_(payload)
    return recv[:, 0].contiguous(), recv[:, 1].contiguous(), recv[:, 2].contiguous()


def exchange_rows_device(engine, key, flow_end_s, value, world, rank, group=None, host_collective=False):
    """Row-sharded ingest with everything resident in HBM: the engine buckets this rank's rows (or partial points) by
    owner on the GPU (tad_shard_rows: LDS histogram per workgroup, one reservation per destination, local key ids), the
    per-destination counts travel first, then the three columns move with one all-to-all(v) each (RCCL over xGMI; the
    send buffers are the engine's device arrays, viewed zero-copy).  Returns int64 CUDA tensors (local key, time, value
    bit pattern) of the rows this rank owns.  host_collective=True routes the collectives through host tensors (gloo,
    for running N ranks on fewer GPUs in tests)."""
    import torch
    import torch.distributed as dist
    (dk, dt, dv), sc = engine.shard_rows(key, flow_end_s, value, world)
    dev = key.device if hasattr(key, "device") else torch.device("cuda", engine.device)
    cdev = "cpu" if host_collective else dev
    send_counts = torch.tensor(sc, dtype=torch.int64, device=cdev)
    recv_counts = torch.zeros_like(send_counts)
    if collectives_on(world):
        dist.all_to_all_single(recv_counts, send_counts, group=group)
    else:
        recv_counts.copy_(send_counts)
    rc = [int(c) for c in recv_counts.tolist()]
    n_send, n_recv = sum(sc), sum(rc)
    out = []
    for arr in (dk, dt, dv):
        send = torch.as_tensor(DeviceColumn(arr.ptr, max(arr.n, 1)), device=dev)[:n_send]
        if host_collective:
            send = send.cpu()
        recv = torch.empty(n_recv, dtype=torch.int64, device=cdev)
        if collectives_on(world):
            dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
        else:
            recv.copy_(send)
        out.append(recv.to(dev))
    if str(dev) != "cpu":
        torch.cuda.synchronize(dev)      # the send buffers are freed below
    for arr in (dk, dt, dv):
        arr.free()
    return tuple(out)


class DeviceColumn:
    """Zero-copy torch view of an engine-owned device array (e.g. TadPoints.device_pointers()): torch.as_tensor accepts
    any object with __cuda_array_interface__ (ROCm builds of torch keep the CUDA name)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def run_sharded(run_local, algo, cols, num_keys, reducer, **job):
    """cols: this rank's rows with LOCAL key ids (shard_rows / exchange_rows).  run_local(algo, key_id, flow_end_s,
    value, num_local_keys, **job) -> object with .stats (TadEngine.run).  Returns (local result, job-wide stats)."""
    nk = max(1, num_local_keys(num_keys, reducer.rank, reducer.world))
    res = run_local(algo, cols["key_id"], cols["flow_end_s"], cols["value"], nk, key_id2=cols.get("key_id2"),
                    flow_start_s=cols.get("flow_start_s"), **job)
    return res, reducer.reduce(res.stats)
