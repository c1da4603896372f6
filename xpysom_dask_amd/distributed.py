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
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return
    engine.sync()                      # host-staged backends (gloo): the engine runs on its own stream
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if t.is_cuda:
        import torch
        torch.cuda.current_stream(t.device).synchronize()


def epoch(engine, sigma, eta, neigh_f64, chunks=None):
    """One data-parallel epoch on this rank's shard (resident rows, or `chunks` streamed through)."""
    if chunks is None:
        engine.epoch_accumulate(sigma, eta, neigh_f64)
    else:
        engine.stream_epoch_accumulate(chunks, sigma, eta, neigh_f64)
    allreduce_accumulator(engine)
    engine.epoch_merge()
