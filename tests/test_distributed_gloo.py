"""The N > 1 path on CPU: two gloo ranks run XPySom.train through the product's host code
(shard split -> per-rank accumulate -> ONE all-reduce of the fused numerator|denominator ->
identical merge on every rank), with the oracle standing in for the device engine."""
import os
import socket

import numpy as np
import pytest

from oracle import som_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sharded_input, out_dir, overlap="0", shard="contiguous"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SOM_OVERLAP=overlap)
    import torch.distributed as dist
    from tests.oracle_engine import OracleEngine
    from xpysom_dask_amd import XPySom
    from xpysom_dask_amd import distributed as D
    from xpysom_dask_amd import engine
    engine.HipEngine = OracleEngine                     # test double in THIS worker process only (no GPU here)
    # (a file rendezvous: a TCP store on a port "asked for and given back" loses a race for it once in a few hundred runs)
    dist.init_process_group("gloo", init_method="file://%s" % os.path.join(out_dir, "rendezvous_%d" % port), rank=rank,
                            world_size=world)
    try:
        assert D.dist_info() == (rank, world)
        data = O.gaussian_blobs(601, 5, seed=11)
        som = XPySom(7, 6, 5, random_seed=3, decay_function="linear", sharded_input=sharded_input, shard=shard)
        mine = data
        if sharded_input:
            mine = D.shard_rows(data, rank, world, shard)
        som.train(mine, 6)
        np.save(os.path.join(out_dir, "w%d.npy" % rank), som._weights)
        if not sharded_input:                              # one row, two ranks: rank 1's shard is empty
            one = O.gaussian_blobs(1, 5, seed=5)
            som1 = XPySom(4, 3, 5, random_seed=3, decay_function="linear")
            som1.train(one, 2)
            np.save(os.path.join(out_dir, "one%d.npy" % rank), som1._weights)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sharded_input", [False, True])
def test_two_rank_training_equals_single_process(tmp_path, sharded_input):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), sharded_input, str(tmp_path)), nprocs=world, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    assert np.array_equal(w0, w1)                       # every rank holds the same codebook, bit for bit
    data = O.gaussian_blobs(601, 5, seed=11)
    ref = O.train(data, O.default_codebook(7, 6, 5, 3), 6, sigma0=3.0, decay="linear", n_parallel=4000)
    np.testing.assert_allclose(w0, ref, rtol=2e-5, atol=2e-6)   # sum order differs with the shard count
    if not sharded_input:
        o0, o1 = np.load(tmp_path / "one0.npy"), np.load(tmp_path / "one1.npy")
        ref1 = O.train(O.gaussian_blobs(1, 5, seed=5), O.default_codebook(4, 3, 5, 3), 2, sigma0=1.5, decay="linear", n_parallel=4000)
        assert np.array_equal(o0, o1)
        np.testing.assert_allclose(o0, ref1, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_eight_rank_training_equals_single_process(tmp_path, overlap):
    """The real world size of BASELINE configs[3] / [4]: eight gloo ranks (75 rows each, one of them with 76), the
    one-shot all-reduce and the blockwise one (three 3-row blocks of the 7-row test map)."""
    import torch.multiprocessing as mp
    world = 8
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), overlap), nprocs=world, join=True)
    ws = [np.load(tmp_path / ("w%d.npy" % r)) for r in range(world)]
    for w in ws[1:]:
        assert np.array_equal(ws[0], w)                 # every rank holds the same codebook, bit for bit
    data = O.gaussian_blobs(601, 5, seed=11)
    ref = O.train(data, O.default_codebook(7, 6, 5, 3), 6, sigma0=3.0, decay="linear", n_parallel=4000)
    np.testing.assert_allclose(ws[0], ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("world,sharded_input", [(2, False), (2, True), (8, False)])
def test_strided_shards_train_the_same_map(tmp_path, world, sharded_input):
    """shard='strided' (rows rank, rank + world, ...): every rank ends on the same codebook bit for bit, and on the
    single-process one to float32 summation order -- as the contiguous split does."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), sharded_input, str(tmp_path), "0", "strided"), nprocs=world, join=True)
    ws = [np.load(tmp_path / ("w%d.npy" % r)) for r in range(world)]
    for w in ws[1:]:
        assert np.array_equal(ws[0], w)
    data = O.gaussian_blobs(601, 5, seed=11)
    ref = O.train(data, O.default_codebook(7, 6, 5, 3), 6, sigma0=3.0, decay="linear", n_parallel=4000)
    np.testing.assert_allclose(ws[0], ref, rtol=2e-5, atol=2e-6)


def test_shard_rows_cover_every_row_once():
    from xpysom_dask_amd.distributed import shard_rows
    data = np.arange(23)
    for shard in ("contiguous", "strided"):
        for world in (1, 2, 3, 8):
            got = np.sort(np.concatenate([shard_rows(data, r, world, shard) for r in range(world)]))
            assert np.array_equal(got, data), (shard, world)
            sizes = [len(shard_rows(data, r, world, shard)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_rows(data, 0, 2, "random")


def test_blockwise_allreduce_host_logic(tmp_path):
    """distributed._epoch_overlapped (begin -> [block b, all-reduce of its slice] ... -> merge) against the one-shot
    all-reduce, two gloo ranks, the test-double engine with three-row blocks: same codebook bit for bit."""
    import torch.multiprocessing as mp
    outs = []
    for overlap in ("0", "1"):
        d = tmp_path / overlap
        d.mkdir()
        mp.spawn(_worker, args=(2, _free_port(), False, str(d), overlap), nprocs=2, join=True)
        w0, w1 = np.load(d / "w0.npy"), np.load(d / "w1.npy")
        assert np.array_equal(w0, w1)
        outs.append(w0)
    assert np.array_equal(outs[0], outs[1])


def test_shard_bounds_cover_and_balance():
    from xpysom_dask_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
