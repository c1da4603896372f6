"""One rank of the N > 1 GPU tests (tests/test_gpu_distributed.py): launched by torch.distributed.run,
trains through the PRODUCT path (XPySom -> HipEngine -> libsomhip) and saves what the test compares.

    dist_worker.py <backend> <out_dir> <mode>     mode: full | sharded | stream | bf16 | exact | wide | wide4
(wide: a 130-row map, i.e. two 128-row blocks of the accumulator, for the blockwise all-reduce: SOM_OVERLAP=0/1;
 wide4: a 512-row map, four blocks -- the block count from which the overlapped epoch is the default under RCCL)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    backend, out_dir, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    import torch
    import torch.distributed as dist
    from oracle import som_oracle as O                       # (seeded test data only)
    from xpysom_dask_amd import XPySom
    from xpysom_dask_amd import distributed as D
    # gloo: every rank shares GPU 0 (a one-GPU box rehearses the N > 1 path); nccl (= RCCL): one GPU per rank
    dev = local if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    try:
        assert D.dist_info() == (rank, world)
        prec = mode if mode in ("bf16", "exact") else "f32"
        sharded = mode in ("sharded", "stream")

        def feed(som, data, T, **kw):
            lo, hi = D.shard_bounds(len(data), rank, world)
            if mode in ("full", "bf16", "exact", "wide", "wide4"):
                som.train(data, T, **kw)                      # every rank passes all rows and keeps its slice
            elif mode == "sharded":
                som.train(data[lo:hi], T, **kw)
            else:                                             # streamed epochs: this rank's rows in three chunks
                mine = data[lo:hi]
                cuts = [0, len(mine) // 3, len(mine) // 2, len(mine)]
                som.train_streaming(lambda: (mine[a:b] for a, b in zip(cuts[:-1], cuts[1:])), T, **kw)

        # (a) a small well-conditioned run end to end (SURVEY 7 hard part 1: only such runs are stable over epochs)
        small = O.gaussian_blobs(601, 5, seed=11)
        som = XPySom(7, 6, 5, random_seed=3, decay_function="linear", device=dev, precision=prec, sharded_input=sharded)
        feed(som, small, 6)
        np.save(os.path.join(out_dir, "ws_%s_%d.npy" % (mode, rank)), som._weights)
        # (b) one teacher-forced epoch (iteration 2 of 5) of a mid-size map from the seeded codebook
        X, Y, Dm, n, T = (130, 6, 16, 6001, 5) if mode == "wide" else (512, 4, 16, 6001, 5) if mode == "wide4" else (24, 20, 16, 6001, 5)
        data = O.gaussian_blobs(n, Dm, seed=11)
        som = XPySom(X, Y, Dm, random_seed=3, decay_function="linear", device=dev, precision=prec, sharded_input=sharded)
        feed(som, data, T, iter_beg=2, iter_end=3)
        np.save(os.path.join(out_dir, "w_%s_%d.npy" % (mode, rank)), som._weights)
        # (c) fewer rows than ranks: the last rank's shard is EMPTY and still takes part in every collective
        one = O.gaussian_blobs(1, 5, seed=5)
        som1 = XPySom(4, 3, 5, random_seed=3, decay_function="linear", device=dev, precision=prec)
        som1.train(one, 2)
        np.save(os.path.join(out_dir, "w1_%s_%d.npy" % (mode, rank)), som1._weights)
        lo, hi = D.shard_bounds(n, rank, world)
        # the collective really summed over the ranks: one more accumulate + all-reduce, fetched raw
        eng = som._engine()
        eng.set_data(data[lo:hi])
        eng.epoch_accumulate(2.0, 0.3, False)
        D.allreduce_accumulator(eng)
        num, den, _ = eng.epoch_fetch(want_bmu=False)
        np.save(os.path.join(out_dir, "den_%s_%d.npy" % (mode, rank)), den)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
