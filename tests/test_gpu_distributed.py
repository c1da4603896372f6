"""The N > 1 path on the REAL engine: two ranks of XPySom -> HipEngine -> libsomhip.

* gloo, both ranks on GPU 0: runs on the one-GPU box; the all-reduce goes through the product's host-staged branch
  of `distributed.allreduce_accumulator` on the engine's own HBM buffer.
* nccl (= RCCL), one GPU per rank: the stream-ordered branch; needs two GPUs, skipped otherwise.
* `python bench.py --gpus 2` from plain python (no launcher): the self-launching harness, weak and strong scaling.

Reference seam: the Dask fan-out / `delayed(sum)` / merge of xpysom.py:545-558,577.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(args, env, world, timeout):
    """torch.distributed.run on a free port; a port asked for and given back can be taken by somebody else before the
    launcher's store listens on it (EADDRINUSE): such a launch is repeated on another port."""
    for attempt in range(3):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
        r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
        if "EADDRINUSE" not in r.stderr and "address already in use" not in r.stderr:
            break
    return r


def _launch(backend, out_dir, mode, world=2, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **extra_env)
    r = _run_ranks([os.path.join(REPO, "tests", "dist_worker.py"), backend, str(out_dir), mode], env, world, 600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]


def _check(out_dir, mode, world=2):
    """Every rank holds the same results, bit for bit; returns rank 0's (small-run codebook, epoch codebook, den)."""
    got = []
    for stem in ("ws", "w", "den", "w1"):
        a = [np.load(os.path.join(out_dir, "%s_%s_%d.npy" % (stem, mode, r))) for r in range(world)]
        for b in a[1:]:
            assert np.array_equal(a[0], b)
        got.append(a[0])
    return got


def _references():
    small = O.gaussian_blobs(601, 5, seed=11)
    ref_small = O.train(small, O.default_codebook(7, 6, 5, 3), 6, sigma0=3.0, decay="linear", n_parallel=4000)
    data = O.gaussian_blobs(6001, 16, seed=11)
    w0 = O.default_codebook(24, 20, 16, 3).astype(np.float32)
    f = O.DECAYS["linear"]
    _, _, _, ref_epoch = O.epoch(data, w0, f(0.5, 0.01, 2, 5), f(10.0, 1, 2, 5), wide=False, n_parallel=6001)
    return data, ref_small, ref_epoch


@pytest.mark.parametrize("mode", ["full", "sharded", "stream"])
def test_two_ranks_of_the_hip_engine_under_gloo(tmp_path, mode):
    _launch("gloo", tmp_path, mode)
    ws, w, den, w1 = _check(tmp_path, mode)
    data, ref_small, ref_epoch = _references()
    np.testing.assert_allclose(ws, ref_small, rtol=2e-5, atol=2e-6)    # float32 sum order differs with the shard count
    one = O.gaussian_blobs(1, 5, seed=5)                               # one row, two ranks: rank 1's shard is empty
    ref1 = O.train(one, O.default_codebook(4, 3, 5, 3), 2, sigma0=1.5, decay="linear", n_parallel=4000)
    np.testing.assert_allclose(w1, ref1, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(w, ref_epoch, rtol=2e-5, atol=2e-6)
    # the all-reduced denominator covers ALL rows, not one shard's
    _, _, den_all = O.update(data, w.astype(np.float32), 0.3, 2.0, wide=False)
    np.testing.assert_allclose(den.reshape(-1), den_all.reshape(-1), rtol=2e-5, atol=1e-6)


def test_two_ranks_bf16_precision_under_gloo(tmp_path):
    _launch("gloo", tmp_path, "bf16")
    ws, w, _, _ = _check(tmp_path, "bf16")
    _, ref_small, ref_epoch = _references()
    # bf16 BMUs: near-tie picks may differ from float32, the epoch's result barely moves
    scale = np.abs(ref_epoch).max()
    assert np.abs(w - ref_epoch).max() < 0.1 * scale and np.abs(w - ref_epoch).mean() < 0.01 * scale
    assert np.isfinite(ws).all() and ws.shape == ref_small.shape   # (six bf16 epochs of a 42-unit map wander off f32's)


def test_blockwise_allreduce_equals_the_monolithic_one(tmp_path):
    """The overlapped epoch (all-reduce of finished 128-row blocks of the accumulator while the next block's
    transform runs; distributed._epoch_overlapped) against one all-reduce of the whole buffer: two ranks, a map of
    two blocks, results equal bit for bit."""
    a, b = tmp_path / "mono", tmp_path / "blocks"
    a.mkdir(); b.mkdir()
    _launch("gloo", a, "wide", SOM_OVERLAP="0")
    _launch("gloo", b, "wide", SOM_OVERLAP="1")
    for x, y in zip(_check(a, "wide"), _check(b, "wide")):
        assert np.array_equal(x, y)


def test_four_ranks_of_the_hip_engine_on_one_gpu(tmp_path):
    """The widest rehearsal one GPU allows (the box admits six processes on its card, the test runner and the launcher
    among them; the real run has eight ranks -- eight run under gloo on the CPU double, tests/test_distributed_gloo.py):
    four real-engine ranks under gloo, precision 'exact', rows split 4 ways -- rank-to-rank bit equality and the oracle's
    training; then the 512-row map, whose four 128-row blocks take the blockwise all-reduce (SOM_OVERLAP=1: the path
    that is the default under RCCL from three blocks on), against the one-shot all-reduce."""
    _launch("gloo", tmp_path, "exact", world=4)
    ws, w, den, w1 = _check(tmp_path, "exact", world=4)
    data, ref_small, ref_epoch = _references()
    np.testing.assert_allclose(ws, ref_small, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(w, ref_epoch, rtol=2e-5, atol=2e-6)
    _, _, den_all = O.update(data, w.astype(np.float32), 0.3, 2.0, wide=False)
    np.testing.assert_allclose(den.reshape(-1), den_all.reshape(-1), rtol=2e-5, atol=1e-6)
    a, b = tmp_path / "mono", tmp_path / "blocks"
    a.mkdir(); b.mkdir()
    _launch("gloo", a, "wide4", world=4, SOM_OVERLAP="0")
    _launch("gloo", b, "wide4", world=4, SOM_OVERLAP="1")
    # (each run's ranks agree bit for bit: _check.  Between the two runs only the ORDER in which a collective adds the
    #  ranks' partial sums may differ -- it depends on where an element sits in the buffer being reduced -- so with
    #  more than two ranks the blockwise and the one-shot result are equal to float32 summation order, not bitwise)
    for x, y in zip(_check(a, "wide4", world=4), _check(b, "wide4", world=4)):
        np.testing.assert_allclose(x, y, rtol=2e-6, atol=2e-6 * np.abs(y).max())


def test_native_collective_with_two_ranks_on_one_gpu_fails_loudly(tmp_path):
    """SOM_COMM=native (RCCL inside libsomhip) needs one GPU per rank: two ranks on ONE card must end with an error from
    ncclCommInitRank on every rank -- promptly, not in a hang -- so that a mis-launched job says what is wrong."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", SOM_COMM="native")
    r = _run_ranks([os.path.join(REPO, "tests", "dist_worker.py"), "gloo", str(tmp_path), "full"], env, 2, 180)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "ncclCommInitRank" in out or "SomHipError" in out, out[-3000:]


def test_two_ranks_under_rccl(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node)")
    for mode in ("full", "stream", "wide"):
        _launch("nccl", tmp_path, mode)
        ws, w, _, _ = _check(tmp_path, mode)
        _, ref_small, ref_epoch = _references()
        np.testing.assert_allclose(ws, ref_small, rtol=2e-5, atol=2e-6)
        if mode != "wide":                                     # (wide: another map; rank-to-rank equality above)
            np.testing.assert_allclose(w, ref_epoch, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launches_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher and no rank environment: exit 0 and one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SOM_DIST_BACKEND"] = "gloo"                           # two ranks on this box's one GPU
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--scaling", scaling]
    cmd += ["--rows", "65536"] if scaling == "weak" else ["--total-rows", "131073"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    assert out["config"]["rows_total"] == (131072 if scaling == "weak" else 131073)
    assert out["value"] > 0 and 0 < out["roofline"]["frac"] < 1
    assert out["codebooks_identical_on_all_ranks"] is True      # every rank merged the same all-reduced sums


def test_bench_reports_the_rank_spread_when_shards_differ_in_structure():
    """Under block skipping a rank's epoch depends on ITS rows, and the all-reduce waits for the slowest rank.  Four
    real-engine ranks on this box's one GPU (gloo), exact mode with skipping on, the first rank's rows WITHOUT structure
    (N(0, I): it skips nothing) beside three ranks of Gaussian blobs: the line carries every rank's own BMU search time
    and executed share, the structureless rank is the slowest and ran several times the others' blocks, and every rank still
    ends on the same codebook (xpysom.py:545-558: one sum of all partials, one merge)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SOM_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "3", "--rows", "65536",
           "--no-cpu-baseline", "--no-throughput-mode", "--no-batch65536", "--no-modes", "--unstructured-ranks", "1"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["codebooks_identical_on_all_ranks"] is True
    ms, sh = out["rank_epoch_ms"], out["rank_executed_share"]
    assert len(ms["by_rank"]) == 4 and ms["min"] <= ms["mean"] <= ms["max"]
    # (the map is trained on every rank's rows, so even the structureless shard skips -- but it runs several times the blocks
    #  of the others, and its BMU search is the one the all-reduce waits for)
    assert sh["by_rank"][0] == sh["max"] and sh["by_rank"][0] > 2.0 * max(sh["by_rank"][1:]), sh
    # (the four ranks share ONE card here, so whose kernels wait for whose is the scheduler's business: the times are
    #  reported, not compared)
    assert all(v > 0 for v in ms["by_rank"])
    # what the epoch waits for at its exchange step, and the ranks' spread
    ar = out["allreduce_exposed_ms"]
    assert len(ar["by_rank"]) == 4 and ar["collectives_per_epoch"] >= 1 and all(v > 0 for v in ar["by_rank"])
    assert out["epoch_ms_max_minus_mean_over_ranks"] >= 0.0

