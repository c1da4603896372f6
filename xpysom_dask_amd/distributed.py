"""Sample-sharded data parallelism: one process per GPU, one all-reduce per epoch.

Replaces the reference's Dask fan-out / gather-sum (xpysom.py:545-558): every rank holds a
contiguous shard of the rows resident on its GPU and the full codebook; per epoch each
rank accumulates its local [numerator | denominator] buffer, ONE all-reduce(sum) over the
fused float32 buffer (RCCL over xGMI when the backend is 'nccl') makes every rank hold the
global sums, and every rank applies the identical merge -- no broadcast of the codebook.
"""
import os


def dist_info():
    """(rank, world_size) of the default process group, or (0, 1) when there is none."""
    import sys
    if "torch" not in sys.modules:     # nobody imported torch, so no process group exists: do not pay for the import
        return 0, 1
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_rows, rank, world):
    """Contiguous, balanced split (the first n_rows % world ranks get one extra row)."""
    base, extra = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


SHARDS = ("contiguous", "strided")


def shard_rows(data, rank, world, shard="contiguous"):
    """This rank's rows of `data` (a host array or a torch tensor; anything sliceable along axis 0).
      contiguous  rows [lo, hi) of shard_bounds: what the reference's Dask blocks are (xpysom.py:490,546)
      strided     rows rank, rank + world, rank + 2 world, ...: every rank sees a sample of the WHOLE file, so an order in
                  the input (rows sorted by class, by time, by source) cannot become rank skew -- under block skipping a
                  rank's epoch depends on ITS rows and the all-reduce waits for the slowest.  The sums are over the same rows
                  either way: the trained map is the same to float32 summation order."""
    if shard not in SHARDS:
        raise ValueError("shard must be one of %s" % ", ".join(SHARDS))
    if world == 1:
        return data
    if shard == "strided":
        mine = data[rank::world]
        if hasattr(mine, "contiguous"):                  # torch: the engine borrows a dense block
            mine = mine.contiguous()
        return mine
    lo, hi = shard_bounds(len(data), rank, world)
    return data[lo:hi]


# bench.py sets this: every all-reduce is bracketed by events on the stream it runs on (host clocks on host-staged
# backends); allreduce_ms() returns and clears what they measured.  What is measured is what the epoch WAITS for: the
# collective itself and, ahead of it, the slowest rank.
TIME_ALLREDUCE = False
_ar_events, _ar_host_ms = [], []


def allreduce_ms():
    """(total ms, collectives) since the last call (TIME_ALLREDUCE)."""
    total, n = float(sum(_ar_host_ms)), len(_ar_host_ms)
    _ar_host_ms.clear()
    if _ar_events:
        import torch
        torch.cuda.synchronize()
        for a, b in _ar_events:
            total += a.elapsed_time(b)
            n += 1
        _ar_events.clear()
    return total, n


def allreduce_accumulator(engine):
    """Sum the engine's fused accumulator across ranks, in place."""
    rank, world = dist_info()
    if world == 1 and not os.environ.get("SOM_FORCE_ALLREDUCE"):   # (the env var lets a 1-GPU box exercise the path)
        return
    import torch.distributed as dist
    t = engine.accum_tensor()
    if t.is_cuda and dist.get_backend() == "nccl":
        # RCCL: stream-ordered, no host synchronisation.  The collective is issued with the engine's own
        # stream current, so it starts after the accumulate kernels queued there, and the merge kernel
        # queued next waits for it; the host runs ahead into the next epoch's launches.
        import torch
        try:
            ext = torch.cuda.ExternalStream(engine.stream_ptr(), device=t.device)
        except Exception:              # no ExternalStream in this torch build: the synchronising form below
            ext = None
        if ext is not None:
            with torch.cuda.stream(ext):
                if TIME_ALLREDUCE:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(ext)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                if TIME_ALLREDUCE:
                    b.record(ext)
                    _ar_events.append((a, b))
            return
    engine.sync()                      # host-staged backends (gloo): the engine runs on its own stream
    if TIME_ALLREDUCE:
        import time
        t0 = time.perf_counter()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if t.is_cuda:
        import torch
        torch.cuda.current_stream(t.device).synchronize()
    if TIME_ALLREDUCE:
        _ar_host_ms.append(1e3 * (time.perf_counter() - t0))


_comm_streams = {}


def _overlap_wanted(engine, world):
    """Blockwise all-reduce under the transform's tail: on by default under RCCL with more than one rank when the map
    has more than two 128-row blocks (configs[4]'s 512-row map: four blocks of 206 MB; at 256 rows the transform is
    0.15 ms against a 34 MB all-reduce -- nothing to hide it under, and two small collectives are slower than one);
    SOM_OVERLAP=0 / 1 forces it off / on (1 also on host-staged backends, where the blocks are reduced one
    synchronous call at a time -- the equality tests run it under gloo)."""
    force = os.environ.get("SOM_OVERLAP")
    if force is not None:
        return force != "0" and hasattr(engine, "epoch_accumulate_block")
    if world == 1 or not hasattr(engine, "epoch_accumulate_block"):
        return False
    import torch.distributed as dist
    return dist.get_backend() == "nccl" and engine.epoch_block_count() > 2


def _epoch_overlapped(engine, sigma, eta, neigh_f64):
    """accumulate -> [stage 2 of block b  ||  all-reduce of block b-1] ... -> merge.  The slices of the fused
    accumulator are disjoint, so the result equals the monolithic all-reduce element for element."""
    import torch
    import torch.distributed as dist
    engine.epoch_accumulate_begin(sigma, eta, neigh_f64)
    t = engine.accum_tensor()
    nb = engine.epoch_block_count()
    stream_ordered = t.is_cuda and dist.get_backend() == "nccl"
    ext = comm = None
    if stream_ordered:
        try:
            ext = torch.cuda.ExternalStream(engine.stream_ptr(), device=t.device)
        except Exception:
            stream_ordered = False
    if stream_ordered:
        comm = _comm_streams.get(t.device.index)
        if comm is None:
            comm = _comm_streams[t.device.index] = torch.cuda.Stream(device=t.device)
    for b in range(nb):
        off, n = engine.epoch_accumulate_block(b)
        if stream_ordered:
            ev = torch.cuda.Event()
            ev.record(ext)                             # block b's kernels are queued on the engine's stream
            comm.wait_event(ev)
            with torch.cuda.stream(comm):
                dist.all_reduce(t[off:off + n], op=dist.ReduceOp.SUM)
        else:
            engine.sync()
            dist.all_reduce(t[off:off + n], op=dist.ReduceOp.SUM)
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
    if stream_ordered:
        ext.wait_stream(comm)                          # the merge queued next waits for the last collective
    engine.epoch_merge()


def _native_comm(engine, rank, world):
    """SOM_COMM=native: the all-reduce runs inside libsomhip (RCCL bound with dlopen, include/somhip.h som_comm_*);
    torch.distributed only carries rank 0's 128-byte RCCL id to the other ranks, once per engine."""
    if getattr(engine, "has_comm", False):
        return
    ids = [engine.comm_unique_id() if rank == 0 else None]
    if world > 1:
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", engine.device) if dist.get_backend() == "nccl" else None   # this rank's GPU, not cuda:0
        dist.broadcast_object_list(ids, src=0, device=dev)
    engine.comm_init(world, rank, ids[0])


def epoch(engine, sigma, eta, neigh_f64, chunks=None):
    """One data-parallel epoch on this rank's shard (resident rows, or `chunks` streamed through)."""
    rank, world = dist_info()
    if os.environ.get("SOM_COMM") == "native" and hasattr(engine, "comm_init") and \
            (world > 1 or os.environ.get("SOM_FORCE_ALLREDUCE")):
        _native_comm(engine, rank, world)
        if chunks is None:
            engine.epoch(sigma, eta, neigh_f64)        # accumulate, RCCL all-reduce (blockwise on wide maps), merge
        else:
            engine.stream_epoch_accumulate(chunks, sigma, eta, neigh_f64)
            engine.epoch_allreduce()
            engine.epoch_merge()
        return
    if chunks is None:
        if (world > 1 or os.environ.get("SOM_FORCE_ALLREDUCE")) and _overlap_wanted(engine, world):
            return _epoch_overlapped(engine, sigma, eta, neigh_f64)
        engine.epoch_accumulate(sigma, eta, neigh_f64)
    else:
        engine.stream_epoch_accumulate(chunks, sigma, eta, neigh_f64)
    allreduce_accumulator(engine)
    engine.epoch_merge()
