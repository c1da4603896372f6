#!/usr/bin/env python3
"""Headline benchmark: samples/sec/epoch of the batch-SOM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]

`--gpus N` (N > 1) needs no launcher: when no rank environment is present this process starts its own N rank
processes (`python -m torch.distributed.run`, one per GPU) BEFORE it touches the GPU, relays their output and
exits with their code.  Under an external launcher (the driver's `torch.distributed.run ... bench.py --gpus N`)
it is a rank and reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

Workload (BASELINE.json configs[2]/[3]): 256x256 map, 128 features, synthetic Gaussian-blob rows resident in HBM,
16-bit MFMA distance GEMM, default schedule.  A step = one full epoch over the resident rows: codebook prep, fused
distance+BMU, segment sum, separable neighbourhood transform, (N > 1: one RCCL all-reduce of the fused
numerator|denominator buffer), merge.
The headline mode is precision='exact': the BMUs -- hence every epoch's accumulators and the trained codebook -- are bit
for bit those of the float32 parity mode (the reference's own arithmetic), found by an IEEE-half MFMA screen over all
units and a float32 re-score of the candidates its error bound leaves (csrc/bmu_exact.hpp).  `throughput_mode` in the
same line is the plain bf16 kernel (round 1-2's headline: faster, but its BMUs differ from float32's on up to 70 % of
the rows of a smooth mid-schedule map); `precision_modes_at_batch65536` quantifies every mode's agreement.
  --scaling weak   (default) 1,048,576 rows PER GPU: N = 1 is configs[2], N = 8 is configs[3] (8 Mi rows on 8 GPUs)
  --scaling strong configs[3]'s 8,388,608 rows IN TOTAL split over the N ranks (N = 1 holds all of them)

One JSON line on rank 0: the driver's contract fields + `roofline` (dominant kernel = the fused distance+BMU
kernel, MFMA-bound, hipEvent-timed inside the timed region; `roofline.batch65536` = the same kernel at the
north-star's batch of 65 536 rows, timed in the same process) + `cpu_baseline` (the NumPy port of the reference
path, timed on this host's cores, N = 1 only).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3          # v_mfma_f32_32x32x2_f32
STRONG_TOTAL_ROWS = 8 * (1 << 20)     # BASELINE configs[3]
NORTH_STAR_BATCH = 65536

# `c3` is the configuration BASELINE.json's metric is quoted on and the default; the other two put the
# remaining GPU configs through the same harness (python bench.py --workload c5).
WORKLOADS = {
    "c3": dict(map=(256, 256), features=128, rows=1 << 20, precision="exact", distance="euclidean",
               neighborhood="gaussian", cpu_rows=8192, label="BASELINE configs[2]/[3]"),
    "c2": dict(map=(64, 64), features=32, rows=100000, precision="f32", distance="euclidean",
               neighborhood="gaussian", cpu_rows=100000, label="BASELINE configs[1]"),
    "c5": dict(map=(512, 512), features=784, rows=250000, precision="exact", distance="cosine",
               neighborhood="mexican_hat", cpu_rows=256, label="BASELINE configs[4], one GPU's shard of 2M rows"),
    # configs[4]'s map and shard size with the ORDINARY wide pairing, euclidean + gaussian on unnormalised blobs (the G17
    # family; MNIST-shaped rows): where block skipping beyond 128 features engages (csrc/exact_skip_wide.hpp)
    "c5e": dict(map=(512, 512), features=784, rows=250000, precision="exact", distance="euclidean",
                neighborhood="gaussian", cpu_rows=256, label="configs[4]'s map and shard size, euclidean + gaussian (not a BASELINE config)"),
}


def workload_rows(name, n, seed):
    """Synthetic rows of SURVEY 8(d): Gaussian blobs; for c5 non-negative and L2-normalised (MNIST-like).
    Generated a million rows at a time so that 8 Mi rows never hold a float64 copy of themselves."""
    from xpysom_dask_amd.synthetic import gaussian_blobs
    d = WORKLOADS[name]["features"]
    out = np.empty((n, d), dtype=np.float32)
    step = 1 << 20
    for i, lo in enumerate(range(0, n, step)):
        hi = min(n, lo + step)
        # chunk 0 of seed s is exactly gaussian_blobs(n, d, seed=s) for n <= 2^20 (the round-1 workload)
        x = gaussian_blobs(hi - lo, d, seed=seed if i == 0 else seed + 7919 * i, centre_seed=1234)
        if name == "c5":
            x = np.abs(x)
            x /= np.linalg.norm(x, axis=1, keepdims=True)
        out[lo:hi] = x
    return out


def _hook(name, default):
    """A developer switch of the library: read, as the library reads it, only under SOM_TEST_HOOKS=1."""
    on = os.environ.get("SOM_TEST_HOOKS")
    try:
        on = on is not None and int(on) != 0
    except ValueError:
        on = False
    return os.environ.get(name, default) if on else default


def exact_has_screen(features, units, distance):
    """precision='exact' screens on half operands (<= 128 features: euclidean; 129..800 features on maps of >= 4096
    units: euclidean and cosine -- som_create); elsewhere the float32 kernels serve it."""
    if features <= 128:
        return distance == "euclidean"
    return features <= 800 and units >= 4096 and distance in ("euclidean", "cosine") and _hook("SOM_BF16_WIDE", "1") != "0"


def kernel_name_for(precision, features, units=1 << 16):
    """The BMU kernel som_create selects (csrc/somhip.hip)."""
    if precision == "exact":                               # (its screen kernel; no screen: served by the float32 kernels)
        if features > 128:
            return "bmu_bf16_wide_kernel" if exact_has_screen(features, units, "euclidean") else "bmu_f32_tiled_kernel"
        return "bmu_bf16_k16_kernel"
    if precision == "f32":
        return "bmu_f32_tiled_kernel" if features > 128 else "bmu_f32_res_kernel"
    if features > 128:
        wide = units >= 4096 and features <= 800 and _hook("SOM_BF16_WIDE", "1") != "0"
        return "bmu_bf16_wide_kernel" if wide else "bmu_bf16_tiled_kernel"
    return "bmu_bf16_k16_kernel"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(name="c3"):
    """The oracle (NumPy port of xpysom.py:515-577, float64 neighbourhood as with the default
    'exponential' decay) on a truncated epoch of the same workload; throughput is N-independent
    once rows >> n_parallel would be reached, so samples/s transfers."""
    from oracle import som_oracle as O
    wl = WORKLOADS[name]
    (MAP_X, MAP_Y), FEATURES, rows = wl["map"], wl["features"], wl["cpu_rows"]
    cores = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        th = [p.get("num_threads") for p in threadpool_info() if p.get("user_api") == "blas"]
        if th:
            cores = int(max(th))
    except Exception:
        pass
    data = workload_rows(name, rows, 1234)
    w = O.default_codebook(MAP_X, MAP_Y, FEATURES, 1234).astype(np.float32)
    if name == "c5":
        w = np.abs(w)
    n_par = max(1, (os.cpu_count() or 1) * 500)       # the reference's CPU rule, xpysom.py:45,246
    sig, eta = O.exponential_decay(min(MAP_X, MAP_Y) / 2, 1, 0, 10), O.exponential_decay(0.5, 0.01, 0, 10)
    kw = dict(wide=True, n_parallel=n_par, distance=wl["distance"], neighbourhood=wl["neighborhood"])
    O.epoch(data[: max(1, rows // 8)], w, eta, sig, **kw)                     # BLAS threads up, pages touched
    reps, t0 = 0, time.perf_counter()
    while reps < 8 and (reps == 0 or time.perf_counter() - t0 < 10.0):        # about 10 s of CPU work, bounded
        O.epoch(data, w, eta, sig, **kw)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": rows / dt, "unit": "samples/sec/epoch", "cores": cores, "cpu_model": cpu_model(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d rows of the same %dx%dx%d workload, mean of %d epochs after a warm-up, n_parallel=%d, "
                      "NumPy+OpenBLAS, float64 neighbourhood (exponential decay)" % (rows, MAP_X, MAP_Y, FEATURES, reps, n_par)}


def pmc_traffic(workload, rows, precision, kernel, build_hash):
    """Fabric bytes per launch of the dominant kernel from committed `rocprofv3 --pmc` passes of this same
    command (a process cannot profile itself): the newest profiles/*pmc_traffic*.json whose workload, rows per
    launch and precision match, preferring one collected on THIS build (tools/traffic.sh writes them)."""
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_traffic*.json"))):
        try:
            with open(path) as f:
                pm = json.load(f)
            if pm.get("rows_per_launch") != rows or pm.get("workload", "c3") != workload \
                    or pm.get("precision", "bf16") != precision or pm.get("library", "libsomhip.so") != "libsomhip.so":
                continue                                   # (another workload, or a timing-experiment build)
            # "<kernel><4>" too; exact: the IEEE-half screen instance -- the tile-list instance (block skipping: the launches of the
            # timed epochs) before the full-scan one
            cands = sorted(((name, v) for name, v in pm.items() if name.startswith(kernel) and isinstance(v, dict) and
                            (precision != "exact" or "F16" in name or "f16" in name)),
                           key=lambda nv: (0 if nv[0].rstrip().endswith("true, true>") else 1, nv[0]))
            k = cands[0][1]
            # (the launches of the line's timed epochs where the report has them: under block skipping a launch's traffic
            # depends on the epoch it serves)
            cand = {"bytes": k.get("timed_fabric_bytes_corrected", k["fabric_bytes_corrected"]),
                    "launches": "timed epochs" if "timed_fabric_bytes_corrected" in k else "every launch of the process",
                    "file": os.path.basename(path), "build": pm.get("build")}
        except Exception:
            continue
        if best is None or cand["build"] == build_hash or best["build"] != build_hash:
            best = cand
    return best


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes of a parent
    that never touches the GPU (no exec from a GPU-initialised process), relay their output, return their code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "4")
    rc = 1
    for attempt in range(3):
        # a free port, asked for and given back: somebody else can take it before the launcher's store listens there
        # (EADDRINUSE, seen once in ~100 launches).  A launch that dies within seconds is tried again on another port.
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.time()
        rc = subprocess.call(cmd, env=env)
        if rc == 0 or time.time() - t0 > 20.0:
            break
        print("bench.py: the launcher exited with %d after %.1f s; trying another port" % (rc, time.time() - t0), file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS), help="c3 = the metric's configuration (default)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --rows per GPU (default); strong: --total-rows split over the ranks")
    ap.add_argument("--rows", type=int, default=None, help="weak scaling: rows per GPU (default: the workload's)")
    ap.add_argument("--total-rows", type=int, default=STRONG_TOTAL_ROWS, help="strong scaling: rows of the whole job")
    ap.add_argument("--precision", default=None, choices=["exact", "bf16", "f32", "f16"])
    ap.add_argument("--no-modes", action="store_true", help="skip the per-precision-mode block at batch 65 536")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the bf16 run beside the exact headline (profiling passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch65536", action="store_true")
    ap.add_argument("--unstructured-ranks", type=int, default=0,
                    help="weak scaling: the first k ranks draw rows WITHOUT structure (N(0, I)): shards that skip nothing beside shards that skip most")
    ap.add_argument("--no-schedule", action="store_true", help="skip the epoch-by-epoch / whole-schedule / unstructured-rows block")
    ap.add_argument("--no-f32-check", action="store_true", help="skip the float32 run the headline codebook is compared with")
    ap.add_argument("--shard", default="contiguous", choices=["contiguous", "strided"],
                    help="strong scaling: which rows of the one data set a rank takes (strided: rank, rank + N, ...: no rank skew from the file's order)")
    ap.add_argument("--no-variants", action="store_true", help="skip the data_variants / survey_schedule / queries blocks")
    args = ap.parse_args()

    have_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_rank_env:
        raise SystemExit(self_launch(args))

    wl = WORKLOADS[args.workload]
    (MAP_X, MAP_Y), FEATURES = wl["map"], wl["features"]
    if args.precision is None:
        args.precision = wl["precision"]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from xpysom_dask_amd import build as B
    from xpysom_dask_amd import distributed as D
    from xpysom_dask_amd.decays import exponential_decay
    from xpysom_dask_amd.engine import HipEngine
    import torch
    dist = None
    # one process per GPU; SOM_DIST_BACKEND=gloo lets several ranks rehearse the path on ONE GPU
    backend = os.environ.get("SOM_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > max(1, ndev):
        raise SystemExit("--gpus %d but this node shows %d GPU(s) (SOM_DIST_BACKEND=gloo rehearses ranks on one GPU)"
                         % (world, ndev))
    dev = local % max(1, ndev)
    if world > 1 or os.environ.get("SOM_FORCE_ALLREDUCE"):   # (the env var: a 1-rank group, to rehearse the collective path)
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    lo = hi = 0
    if args.scaling == "strong":
        lo, hi = D.shard_bounds(args.total_rows, rank, world)
        my_rows, total_rows = hi - lo, args.total_rows
    else:
        my_rows = args.rows if args.rows is not None else wl["rows"]
        total_rows = my_rows * world

    rs = np.random.RandomState(1234)                  # default codebook init, xpysom.py:189-190
    w = rs.rand(MAP_X, MAP_Y, FEATURES) * 2 - 1
    w /= np.linalg.norm(w, axis=-1, keepdims=True)
    if args.workload == "c5":
        w = np.abs(w)
    w = w.astype(np.float32)
    # strong scaling: ONE data set, every rank holds its contiguous slice of it (an N = 1 and an N = 8 run train on
    # the same rows, so their codebooks can be compared); weak scaling: each rank draws its own rows
    rows_host = (np.ascontiguousarray(D.shard_rows(workload_rows(args.workload, args.total_rows, 1234), rank, world, args.shard))
                 if args.scaling == "strong" else workload_rows(args.workload, my_rows, 1234 + rank))
    my_rows = len(rows_host)
    if args.scaling == "weak" and rank < args.unstructured_ranks:
        rows_host = np.random.default_rng(4321 + rank).standard_normal((my_rows, FEATURES)).astype(np.float32)
        if args.workload == "c5":
            rows_host = np.abs(rows_host)
            rows_host /= np.linalg.norm(rows_host, axis=1, keepdims=True)

    total = args.warmup + args.steps
    sched = [(exponential_decay(min(MAP_X, MAP_Y) / 2, 1, t, total), exponential_decay(0.5, 0.01, t, total))
             for t in range(total)]

    def fence(eng):
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    first_epoch = {}

    def timed_run(precision):
        """W warm-up epochs, then exactly K timed epochs between fences; max over ranks.  Returns the engine too."""
        eng = HipEngine(MAP_X, MAP_Y, FEATURES, precision=precision, device=dev, distance=wl["distance"],
                        neighborhood=wl["neighborhood"])
        eng.set_weights(w)
        eng.set_data(rows_host)
        for t in range(args.warmup):
            D.epoch(eng, sched[t][0], sched[t][1], True)
            if t == 0 and precision == args.precision:
                first_epoch["w"] = eng.get_weights()      # the codebook one epoch from the seed (strong scaling: probe)
        fence(eng)
        D.TIME_ALLREDUCE = dist is not None                # (events around every collective: what the epoch waits for)
        D.allreduce_ms()
        # The timed region carries HIP events around the dominant (BMU) kernels only -- two event records per epoch
        # (precision 'exact': two more per screen pass).  Event pairs around every kernel family put a ~10 us bubble
        # on the stream at each of the four phase boundaries of an epoch (kernel trace): that breakdown
        # is taken in a separate pass below.
        eng.profile_reset()
        eng.profile_enable("bmu")
        sk0 = eng.exact_skip_stats() if precision == "exact" else (0, 0)
        t0 = time.perf_counter()
        for t in range(args.warmup, total):
            D.epoch(eng, sched[t][0], sched[t][1], True)
        fence(eng)
        dt = time.perf_counter() - t0
        eng.profile_enable(False)
        eng.allreduce_ms = D.allreduce_ms()
        D.TIME_ALLREDUCE = False
        sk1 = eng.exact_skip_stats() if precision == "exact" else (0, 0)
        # block skipping (csrc/exact_skip.hpp): the share of the distance GEMM's (256-row tile, 64-unit group) blocks the
        # screens of the timed epochs actually ran
        eng.executed_share = (sk1[0] - sk0[0]) / (sk1[1] - sk0[1]) if sk1[1] > sk0[1] else 1.0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return eng, dt

    eng, dt = timed_run(args.precision)
    head_share = getattr(eng, "executed_share", 1.0)
    # N > 1: the all-reduce waits for the slowest rank, and under block skipping a rank's BMU search depends on ITS rows:
    # every rank's own search time per epoch (hipEvents around its BMU kernels: no waiting in it) and executed share
    rank_spread = None
    if dist is not None:
        ar_ms, ar_n = getattr(eng, "allreduce_ms", (0.0, 0))
        mine = torch.tensor([eng.profile_get("bmu")[0] / max(1, args.steps), head_share, ar_ms / max(1, args.steps)], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        ms_r = [float(v[0].item()) for v in allv]
        sh_r = [float(v[1].item()) for v in allv]
        ar_r = [float(v[2].item()) for v in allv]
        rank_spread = {"rank_epoch_ms": {"min": min(ms_r), "mean": sum(ms_r) / world, "max": max(ms_r), "by_rank": ms_r,
                                         "what": "each rank's own BMU search per epoch (hipEvents on its stream; the all-reduce is not in it)"},
                       "rank_executed_share": {"min": min(sh_r), "mean": sum(sh_r) / world, "max": max(sh_r), "by_rank": sh_r},
                       "unstructured_ranks": args.unstructured_ranks,
                       # what an epoch WAITS for at its one exchange step: events on the collective's stream around the
                       # all-reduce (host clocks on host-staged backends) -- the collective itself plus, ahead of it, the
                       # slowest rank; the fastest rank's figure is the most skew, the slowest rank's the collective alone
                       "allreduce_exposed_ms": {"min": min(ar_r), "mean": sum(ar_r) / world, "max": max(ar_r), "by_rank": ar_r,
                                                "collectives_per_epoch": ar_n / max(1, args.steps), "bytes": 4 * MAP_X * MAP_Y * (FEATURES + 1 + (-(FEATURES + 1)) % 4),
                                                "what": "per epoch, per rank: from the all-reduce's issue on its stream to its end (waiting for the slowest rank included)"},
                       "epoch_ms_max_minus_mean_over_ranks": max(ms_r) - sum(ms_r) / world,
                       "shard": args.shard if args.scaling == "strong" else "own rows per rank (weak scaling)"}
    w_after_timed = eng.get_weights() if args.precision == "exact" else None
    bmu_ms, bmu_n = eng.profile_get("bmu")
    scr_ms, scr_n = eng.profile_get("screen")
    # per-kernel-family breakdown: the same epochs once more (the same schedule entries, at most 20), untimed, every
    # family under events
    nb = max(1, min(args.steps, 20))
    eng.profile_reset()
    eng.profile_enable(True)
    for t in range(args.warmup, args.warmup + nb):
        D.epoch(eng, sched[t][0], sched[t][1], True)
    fence(eng)
    eng.profile_enable(False)
    parts = {k: eng.profile_get(k)[0] / nb for k in ("prep", "bmu", "segsum", "kron", "merge")}
    if args.precision == "exact":
        parts["bmu_of_which_screen"] = eng.profile_get("screen")[0] / nb
    w_end = eng.get_weights()
    assert np.isfinite(w_end).all()
    exact_stats = eng.exact_stats() if args.precision == "exact" else None
    # N > 1: every rank merged the same all-reduced sums, so the codebooks must be the same bits on every rank
    ranks_agree = None
    import zlib
    w_crc = zlib.crc32(np.ascontiguousarray(w_end).tobytes())
    if dist is not None:
        lo_t = torch.tensor([float(w_crc)], dtype=torch.float64, device="cuda")
        hi_t = lo_t.clone()
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_t, op=dist.ReduceOp.MAX)
        ranks_agree = bool(lo_t.item() == hi_t.item())

    # the throughput mode beside the parity-grade headline: the same K epochs through the plain bf16 kernel
    thr = None
    if args.precision == "exact" and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"]) and not args.no_throughput_mode:
        eng.close()
        e_t, dt_t = timed_run("bf16")
        t_ms, t_n = e_t.profile_get("bmu")
        thr = {"precision": "bf16", "value": total_rows / (dt_t / args.steps), "unit": "samples/sec/epoch",
               "ms_per_step": 1e3 * dt_t / args.steps, "kernel": kernel_name_for("bf16", FEATURES, MAP_X * MAP_Y),
               "avg_launch_ms": t_ms / max(1, t_n),
               "roofline_frac": 2.0 * (MAP_X * MAP_Y) * FEATURES * my_rows / (t_ms / max(1, t_n) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
               "note": "BMUs NOT those of float32 (see precision_modes_at_batch65536): the parity contract holds in the headline mode"}
        e_t.close()
        eng = HipEngine(MAP_X, MAP_Y, FEATURES, precision=args.precision, device=dev, distance=wl["distance"],
                        neighborhood=wl["neighborhood"])

    # ... and the headline mode itself with every block of the distance GEMM run (SOM_EXACT_SKIP=0): what the same kernels do
    # when nothing can be skipped (the first epoch on fresh rows; data without structure)
    full_scan = None
    # (the condition is the same on every rank -- the run below holds collectives: never a rank's own measured share)
    if args.precision == "exact" and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"]) and not args.no_throughput_mode \
            and (FEATURES <= 128 or wl["distance"] == "euclidean") and MAP_X * MAP_Y >= 4096 and os.environ.get("SOM_EXACT_SKIP", "1") != "0":
        old_env = os.environ.get("SOM_EXACT_SKIP")
        os.environ["SOM_EXACT_SKIP"] = "0"
        e_f, dt_f = timed_run("exact")
        if old_env is None:
            del os.environ["SOM_EXACT_SKIP"]
        else:
            os.environ["SOM_EXACT_SKIP"] = old_env
        f_ms, f_n = e_f.profile_get("screen")
        full_scan = {"SOM_EXACT_SKIP": 0, "value": total_rows / (dt_f / args.steps), "unit": "samples/sec/epoch",
                     "ms_per_step": 1e3 * dt_f / args.steps, "avg_launch_ms": f_ms / max(1, f_n), "launches": f_n,
                     "codebook_equal_to_headline_run": bool(np.array_equal(e_f.get_weights(), w_after_timed))}
        e_f.close()
    # What `value` is and is not.  `value` = the mean of the K timed epochs BEHIND the W warm-up epochs of a (W + K)-epoch
    # schedule.  Under block skipping an epoch's cost depends on where in the schedule it sits (the first epochs scan
    # everything) and on the rows (rows without structure skip nothing), so the same line carries: every epoch of the
    # schedule from a FRESH engine, one by one (`per_epoch_ms`, `executed_share_per_epoch`, `first_epoch_ms`), what
    # train(data, W + K) costs as a whole (`whole_schedule`), the same run on rows WITHOUT structure (`unstructured_rows`), and
    # the headline codebook against precision='f32' run over the same schedule (`codebook_equal_to_f32_run`).
    schedule_block, unstructured, equal_f32 = None, None, None
    if args.precision == "exact" and not args.no_schedule and dist is None:
        def one_by_one(rows, prec="exact"):
            e = HipEngine(MAP_X, MAP_Y, FEATURES, precision=prec, device=dev, distance=wl["distance"], neighborhood=wl["neighborhood"])
            e.set_weights(w)
            e.set_data(rows)
            e.sync()
            ms, sh = [], []
            for t in range(total):
                s0 = e.exact_skip_stats() if prec == "exact" else (0, 0)
                t0 = time.perf_counter()
                D.epoch(e, sched[t][0], sched[t][1], True)
                e.sync()
                ms.append(1e3 * (time.perf_counter() - t0))
                s1 = e.exact_skip_stats() if prec == "exact" else (0, 0)
                sh.append((s1[0] - s0[0]) / (s1[1] - s0[1]) if s1[1] > s0[1] else 1.0)
            return e, ms, sh
        e_s, ms_s, sh_s = one_by_one(rows_host)
        w_sched = e_s.get_weights()
        res = e_s.exact_resident_stats()
        e_s.close()
        schedule_block = {
            "per_epoch_ms": [round(v, 4) for v in ms_s], "executed_share_per_epoch": [round(v, 5) for v in sh_s],
            "first_epoch_ms": ms_s[0],
            "whole_schedule": {"epochs": total, "ms_per_epoch": sum(ms_s) / total, "value": my_rows / (sum(ms_s) / total * 1e-3),
                               "unit": "samples/sec/epoch",
                               "note": "a fresh engine, epochs 0..%d one by one with a host sync after each (what train(data, %d) costs)" % (total - 1, total)},
            "epochs_under_a_plan": res[0], "of_which_sorted_the_rows": res[1],
            "codebook_equal_to_headline_run": bool(np.array_equal(w_sched, w_after_timed)),
        }
        if not args.no_f32_check and (FEATURES <= 128 or wl["distance"] == "euclidean"):
            e_f, ms_f, _ = one_by_one(rows_host, "f32")
            equal_f32 = bool(np.array_equal(e_f.get_weights(), w_sched))
            schedule_block["f32_run_ms_per_epoch"] = sum(ms_f) / total
            e_f.close()
            assert equal_f32, "the exact mode's trained codebook is not the float32 mode's"
        # rows without structure: N(0, I), no centres -- nothing for a centroid bound to separate
        rows_u = np.random.default_rng(4321 + rank).standard_normal((my_rows, FEATURES)).astype(np.float32)
        if args.workload == "c5":
            rows_u = np.abs(rows_u)
            rows_u /= np.linalg.norm(rows_u, axis=1, keepdims=True)
        e_u, ms_u, sh_u = one_by_one(rows_u)
        res_u = e_u.exact_resident_stats()
        e_u.close()
        del rows_u
        k_ms = ms_u[args.warmup:]
        unstructured = {"rows": "N(0, I) rows, no centres (same map, same schedule)", "value": my_rows / (sum(k_ms) / len(k_ms) * 1e-3),
                        "unit": "samples/sec/epoch", "ms_per_step": sum(k_ms) / len(k_ms),
                        "executed_share": sum(sh_u[args.warmup:]) / len(k_ms), "epochs_under_a_plan": res_u[0],
                        "plan_paused_epochs": total - 1 - res_u[0], "whole_schedule_ms_per_epoch": sum(ms_u) / total}

    # What the headline is worth elsewhere.  `survey_schedule`: SURVEY 8(d)'s own definition of the metric -- a 10-epoch schedule
    # from the seeded codebook, the mean of epochs 1..9.  `data_variants`: the same map and (W + K)-epoch schedule on rows
    # between "64 separated blobs" and "no structure" (xpysom_dask_amd/synthetic.py, variant()), each with block skipping on
    # and off and against precision='f32'.  `queries`: winner() / quantization_error() on device-resident rows of the trained map.
    survey, variants, queries = None, None, None
    if args.precision == "exact" and not args.no_schedule and not args.no_variants and dist is None and FEATURES <= 128 \
            and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"]):
        from xpysom_dask_amd import synthetic

        def run_schedule(rows, n_epochs, prec="exact", skip=None):
            old = os.environ.get("SOM_EXACT_SKIP")
            if skip is not None:
                os.environ["SOM_EXACT_SKIP"] = skip
            e = HipEngine(MAP_X, MAP_Y, FEATURES, precision=prec, device=dev, distance=wl["distance"], neighborhood=wl["neighborhood"])
            if skip is not None:
                if old is None:
                    del os.environ["SOM_EXACT_SKIP"]
                else:
                    os.environ["SOM_EXACT_SKIP"] = old
            e.set_weights(w)
            e.set_data(rows)
            e.sync()
            sc = [(exponential_decay(min(MAP_X, MAP_Y) / 2, 1, t, n_epochs), exponential_decay(0.5, 0.01, t, n_epochs)) for t in range(n_epochs)]
            ms, sh = [], []
            for t in range(n_epochs):
                s0 = e.exact_skip_stats() if prec == "exact" else (0, 0)
                t0 = time.perf_counter()
                D.epoch(e, sc[t][0], sc[t][1], True)
                e.sync()
                ms.append(1e3 * (time.perf_counter() - t0))
                s1 = e.exact_skip_stats() if prec == "exact" else (0, 0)
                sh.append((s1[0] - s0[0]) / (s1[1] - s0[1]) if s1[1] > s0[1] else 1.0)
            wv = e.get_weights()
            return e, ms, sh, wv

        e_v, ms_v, sh_v, w_v = run_schedule(rows_host, 10)
        e_v.close()
        e_o, ms_o, _, w_o = run_schedule(rows_host, 10, skip="0")
        e_o.close()
        survey = {"epochs": 10, "what": "SURVEY 8(d): 10-epoch default schedule from the seeded codebook, mean of epochs 1..9",
                  "ms_per_epoch": sum(ms_v[1:]) / 9, "value": my_rows / (sum(ms_v[1:]) / 9 * 1e-3), "unit": "samples/sec/epoch",
                  "per_epoch_ms": [round(v, 4) for v in ms_v], "executed_share_per_epoch": [round(v, 5) for v in sh_v],
                  "without_block_skipping_ms_per_epoch": sum(ms_o[1:]) / 9,
                  "without_block_skipping_per_epoch_ms": [round(v, 4) for v in ms_o],
                  "codebook_equal_with_and_without_skipping": bool(np.array_equal(w_v, w_o))}
        variants = {}
        for kind in ("overlap", "manifold", "heavy"):
            rows_k = synthetic.variant(kind, my_rows, FEATURES, seed=1234 + rank)
            e_k, ms_k, sh_k, w_k = run_schedule(rows_k, total)
            res_k, sc_k = e_k.exact_resident_stats(), e_k.exact_scout_stats()
            e_k.close()
            e_0, ms_0, _, w_0 = run_schedule(rows_k, total, skip="0")
            e_0.close()
            e_f, ms_f, _, w_f = run_schedule(rows_k, total, prec="f32")
            e_f.close()
            k_ms, k_0 = ms_k[args.warmup:], ms_0[args.warmup:]
            variants[kind] = {
                "ms_per_step": sum(k_ms) / len(k_ms), "value": my_rows / (sum(k_ms) / len(k_ms) * 1e-3), "unit": "samples/sec/epoch",
                "executed_share": sum(sh_k[args.warmup:]) / len(k_ms), "whole_schedule_ms_per_epoch": sum(ms_k) / total,
                "without_block_skipping_ms_per_step": sum(k_0) / len(k_0), "without_block_skipping_whole_schedule_ms_per_epoch": sum(ms_0) / total,
                "slowest_epoch_vs_without_skipping": max(a / b for a, b in zip(ms_k, ms_0)),
                "epochs_under_a_plan": res_k[0], "launches_scouted": sc_k[0],
                "codebook_equal_to_f32_run": bool(np.array_equal(w_k, w_f)), "codebook_equal_without_skipping": bool(np.array_equal(w_k, w_0)),
                "per_epoch_ms": [round(v, 3) for v in ms_k], "executed_share_per_epoch": [round(v, 4) for v in sh_k]}
            assert variants[kind]["codebook_equal_to_f32_run"], "data variant %s: the exact mode's trained codebook is not the float32 mode's" % kind
            del rows_k
        variants["what"] = ("the headline's map and %d-epoch schedule on other rows (synthetic.variant): overlap = 1024 centres whose noise has the centres' own "
                            "spread; manifold = a 2-D sheet embedded in %d-D; heavy = 64 blobs with power-law sizes; ms_per_step = mean of epochs %d..%d as in "
                            "`value`" % (total, FEATURES, args.warmup, total - 1))
        # queries on the trained map: rows already in HBM (a torch tensor), other rows of the same mixture
        qrows = torch.from_numpy(workload_rows(args.workload, my_rows, 99 + rank)).cuda()
        torch.cuda.synchronize()
        queries = {"rows": my_rows, "what": "winner() / quantization_error() ids for device-resident rows (som_bmu_device, som_quantization_error_device) on the "
                                            "headline run's trained map; wall ms per call incl. the ids' copy to the host, mean of 5 calls after one"}
        ids_by = {}
        for tag, skip in (("planned", None), ("without_block_skipping", "0")):
            old = os.environ.get("SOM_EXACT_SKIP")
            if skip is not None:
                os.environ["SOM_EXACT_SKIP"] = skip
            e_q = HipEngine(MAP_X, MAP_Y, FEATURES, precision="exact", device=dev, distance=wl["distance"], neighborhood=wl["neighborhood"])
            if skip is not None:
                if old is None:
                    del os.environ["SOM_EXACT_SKIP"]
                else:
                    os.environ["SOM_EXACT_SKIP"] = old
            e_q.set_weights(w_after_timed)
            ids_by[tag] = e_q.bmu_device(qrows.data_ptr(), my_rows)
            e_q.sync()
            e_q.profile_reset()
            e_q.profile_enable("bmu")
            s0, t0 = e_q.exact_skip_stats(), time.perf_counter()
            for _ in range(5):
                e_q.bmu_device(qrows.data_ptr(), my_rows)
            t_w = (time.perf_counter() - t0) / 5
            e_q.profile_enable(False)
            s1 = e_q.exact_skip_stats()
            t0 = time.perf_counter()
            for _ in range(5):
                qe_v = e_q.quantization_error_device(qrows.data_ptr(), my_rows)
            t_q = (time.perf_counter() - t0) / 5
            queries[tag] = {"winner_ms": 1e3 * t_w, "bmu_search_on_the_stream_ms": e_q.profile_get("bmu")[0] / 5,
                            "executed_share": (s1[0] - s0[0]) / max(1, s1[1] - s0[1]), "quantization_error_ms": 1e3 * t_q, "quantization_error": qe_v}
            e_q.close()
        e_q = HipEngine(MAP_X, MAP_Y, FEATURES, precision="f32", device=dev, distance=wl["distance"], neighborhood=wl["neighborhood"])
        e_q.set_weights(w_after_timed)
        ids_f = e_q.bmu_device(qrows.data_ptr(), my_rows)
        e_q.close()
        queries["ids_equal_to_f32"] = bool(np.array_equal(ids_by["planned"], ids_f) and np.array_equal(ids_by["without_block_skipping"], ids_f))
        assert queries["ids_equal_to_f32"], "the planned query path left the float32 ids"
        del qrows

    kernel_name = kernel_name_for(args.precision, FEATURES, MAP_X * MAP_Y)
    peak = MFMA_F32_PEAK_TFLOPS if args.precision == "f32" else MFMA_BF16_PEAK_TFLOPS
    KD2 = 2.0 * (MAP_X * MAP_Y) * FEATURES            # 2*K*D flop per sample

    def kernel_ms(e):
        """(average launch, launches) of the MFMA distance kernel of the epochs just timed: the screen kernel of the
        exact mode, the fused distance+BMU kernel otherwise."""
        fam = "screen" if e.precision == "exact" and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"]) else "bmu"
        ms, n = e.profile_get(fam)
        return ms / max(1, n), n

    # the north-star's launch: the same kernel over a batch of 65 536 resident rows, same process
    batch = None
    do_batch = rank == 0 and world == 1 and not args.no_batch65536 and my_rows >= NORTH_STAR_BATCH
    if do_batch:
        eng.set_weights(w)
        eng.set_data(rows_host[:NORTH_STAR_BATCH])
        reps = 30
        for t in range(3):
            D.epoch(eng, sched[0][0], sched[0][1], True)
        eng.sync()
        eng.profile_reset()
        eng.profile_enable("bmu")
        bs0 = eng.exact_skip_stats() if args.precision == "exact" else (0, 0)
        tb = time.perf_counter()
        for t in range(reps):
            D.epoch(eng, sched[t % total][0], sched[t % total][1], True)
        eng.sync()
        tb = time.perf_counter() - tb
        eng.profile_enable(False)
        bs1 = eng.exact_skip_stats() if args.precision == "exact" else (0, 0)
        b_share = (bs1[0] - bs0[0]) / (bs1[1] - bs0[1]) if bs1[1] > bs0[1] else 1.0
        b_avg, b_n = kernel_ms(eng)
        b_ach = KD2 * NORTH_STAR_BATCH * b_share / (b_avg * 1e-3) / 1e12      # EXECUTED flops
        eng.profile_reset()
        eng.profile_enable(True)                               # breakdown pass (see above)
        for t in range(10):
            D.epoch(eng, sched[t % total][0], sched[t % total][1], True)
        eng.sync()
        eng.profile_enable(False)
        by_k = {k: eng.profile_get(k)[0] / 10 for k in ("prep", "bmu", "segsum", "kron", "merge")}
        if args.precision == "exact":
            by_k["bmu_of_which_screen"] = eng.profile_get("screen")[0] / 10
        batch = {"rows": NORTH_STAR_BATCH, "avg_launch_ms": b_avg, "launches": b_n, "achieved": b_ach,
                 "frac": b_ach / peak, "executed_share": b_share, "epoch_ms": 1e3 * tb / reps, "ms_per_epoch_by_kernel": by_k}
        # the north-star's own figure: the ALGORITHMIC 2 N K D flop of the batch over the distance kernel that runs all of them
        # (the same launch with SOM_EXACT_SKIP=0; target: >= 0.40 of the dense 16-bit peak)
        if args.precision == "exact" and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"]) and os.environ.get("SOM_EXACT_SKIP", "1") != "0":
            old_env = os.environ.get("SOM_EXACT_SKIP")
            os.environ["SOM_EXACT_SKIP"] = "0"
            e_b = HipEngine(MAP_X, MAP_Y, FEATURES, precision="exact", device=dev, distance=wl["distance"], neighborhood=wl["neighborhood"])
            if old_env is None:
                del os.environ["SOM_EXACT_SKIP"]
            else:
                os.environ["SOM_EXACT_SKIP"] = old_env
            e_b.set_weights(w)
            e_b.set_data(rows_host[:NORTH_STAR_BATCH])
            for t in range(3):
                D.epoch(e_b, sched[0][0], sched[0][1], True)
            e_b.sync()
            e_b.profile_reset()
            e_b.profile_enable("bmu")
            tb0 = time.perf_counter()
            for t in range(reps):
                D.epoch(e_b, sched[t % total][0], sched[t % total][1], True)
            e_b.sync()
            tb0 = time.perf_counter() - tb0
            e_b.profile_enable(False)
            f_avg, f_n = kernel_ms(e_b)
            batch["full_scan"] = {"avg_launch_ms": f_avg, "launches": f_n, "epoch_ms": 1e3 * tb0 / reps,
                                  "achieved": KD2 * NORTH_STAR_BATCH / (f_avg * 1e-3) / 1e12,
                                  "frac": KD2 * NORTH_STAR_BATCH / (f_avg * 1e-3) / 1e12 / peak}
            e_b.close()

    # Every precision mode on the same batch: speed AND how far its BMUs / its trained codebook are from float32's --
    # on the seeded codebook (the easiest state a SOM is ever in), on the smooth maps of the early schedule (where a
    # screen is least sure) and at the end of the schedule.  The codebook states come from the float32 arithmetic
    # (trained through 'exact', which is bit for bit the float32 training: checked below against 'f32' itself).
    modes, update_forms = None, None
    if do_batch and not args.no_modes and FEATURES <= 128 and wl["distance"] == "euclidean":
        xb = rows_host[:NORTH_STAR_BATCH]
        T = 10
        sch = [(exponential_decay(min(MAP_X, MAP_Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)) for t in range(T)]
        mk = lambda prec: HipEngine(MAP_X, MAP_Y, FEATURES, precision=prec, device=dev, distance=wl["distance"],
                                    neighborhood=wl["neighborhood"])
        states = {}                                            # epochs done -> float32-trajectory codebook
        tr_e = mk("exact")
        tr_e.set_weights(w)
        tr_e.set_data(xb)
        for t in range(T):
            if t in (0, 1, 5):
                states[t] = tr_e.get_weights()
            D.epoch(tr_e, sch[t][0], sch[t][1], True)
        w_exact_end = tr_e.get_weights()
        tr_e.close()
        modes = {}
        ref = {}
        w_f32_end = None
        for prec, flop_peak in (("f32", MFMA_F32_PEAK_TFLOPS), ("exact", MFMA_BF16_PEAK_TFLOPS), ("f16", MFMA_BF16_PEAK_TFLOPS),
                                ("bf16", MFMA_BF16_PEAK_TFLOPS)):
            e2 = mk(prec)
            e2.set_data(xb)
            agree, ep_ms, k_ms = {}, {}, {}
            for t, wt in states.items():
                e2.set_weights(wt)
                ids = e2.bmu(xb)
                if prec == "f32":
                    ref[t] = ids
                agree["after_%d_epochs" % t] = float(np.mean(ids == ref[t]))
                # this state's epoch, timed (the exact mode's re-score load depends on the map's smoothness)
                e2.epoch_accumulate(sch[t][0], sch[t][1], True)
                e2.sync()
                e2.profile_reset()
                e2.profile_enable("bmu")
                e2.set_weights(wt)
                t2 = time.perf_counter()
                for _ in range(3):                             # (three epochs from this state at its sigma)
                    D.epoch(e2, sch[t][0], sch[t][1], True)
                e2.sync()
                ep_ms["after_%d_epochs" % t] = 1e3 * (time.perf_counter() - t2) / 3
                e2.profile_enable(False)
                k_ms["after_%d_epochs" % t] = kernel_ms(e2)[0]
            # the whole schedule from the seeded codebook in this mode: where does its codebook end up?
            e2.set_weights(w)
            for t in range(T):
                D.epoch(e2, sch[t][0], sch[t][1], True)
            w_end_m = e2.get_weights()
            if prec == "f32":
                w_f32_end = w_end_m
                qe_f32 = e2.quantization_error(xb[:8192])
            qe = e2.quantization_error(xb[:8192])
            k0 = k_ms["after_0_epochs"]
            ach = KD2 * NORTH_STAR_BATCH / (k0 * 1e-3) / 1e12
            modes[prec] = {"rows": NORTH_STAR_BATCH, "epoch_ms": ep_ms, "distance_kernel_launch_ms": k_ms,
                           "achieved_tflops_algorithmic": ach, "frac_of_its_pipe_peak": ach / flop_peak,
                           "bmus_equal_to_float32": agree,
                           "trained_codebook_max_rel_dev_vs_float32": float(np.abs(w_end_m - w_f32_end).max() / np.abs(w_f32_end).max()),
                           "trained_codebook_bitwise_equal_to_float32": bool(np.array_equal(w_end_m, w_f32_end)),
                           "quantization_error_rel_dev_vs_float32": float(abs(qe - qe_f32) / qe_f32),
                           "kernel": kernel_name_for(prec, FEATURES, MAP_X * MAP_Y)}
            if prec == "exact":
                modes[prec]["fallback_rows"], modes[prec]["rows_screened"] = e2.exact_stats()[1], e2.exact_stats()[0]
            e2.close()
        assert np.array_equal(w_exact_end, w_f32_end), "the exact mode's training left the float32 trajectory"

        # both forms of the update on this batch: the bucketed one the product runs, and the reference's
        # own K x N x D formulation g^T x as one float32 MFMA GEMM (som_epoch_accumulate_faithful)
        e3 = mk("exact")
        e3.set_weights(states[1])
        e3.set_data(xb)
        e3.epoch_accumulate_faithful(sch[1][0], sch[1][1], True)
        e3.sync()
        forms = {}
        for name, fn in (("bucketed", e3.epoch_accumulate), ("faithful", e3.epoch_accumulate_faithful)):
            e3.profile_reset()
            e3.profile_enable(True)
            for _ in range(3):
                fn(sch[1][0], sch[1][1], True)
            e3.sync()
            e3.profile_enable(False)
            forms[name] = (e3.profile_get("segsum")[0] + e3.profile_get("kron")[0]) / 3
        fl = 2.0 * NORTH_STAR_BATCH * (MAP_X * MAP_Y) * (FEATURES + 1)
        update_forms = {"rows": NORTH_STAR_BATCH, "bucketed_ms": forms["bucketed"], "faithful_ms": forms["faithful"],
                        "faithful_tflops": fl / (forms["faithful"] * 1e-3) / 1e12,
                        "faithful_frac": fl / (forms["faithful"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "note": "update only (segment sum + transform, or tables + g^T x GEMM), hipEvent-timed; float32 MFMA peak"}
        e3.close()

    # strong scaling: did this run train the SAME map as the one-GPU run of the same rows?  After ONE epoch from the
    # seeded codebook the answer is a number (later epochs amplify float32 summation-order noise chaotically):
    # 4096 fixed entries of the codebook against the probe a --gpus 1 --scaling strong run stored under profiles/.
    probe = None
    if rank == 0 and args.scaling == "strong" and "w" in first_epoch:
        idx = np.random.RandomState(99).randint(0, first_epoch["w"].size, 4096)
        mine = first_epoch["w"].reshape(-1)[idx].astype(np.float64)
        key = {"workload": args.workload, "total_rows": total_rows, "precision": args.precision}
        ref_path = os.path.join(REPO, "profiles", "strong_first_epoch_probe_%s.json" % args.workload)
        probe = {"entries": 4096, "after_epochs": 1}
        try:
            with open(ref_path) as f:
                ref_p = json.load(f)
            if all(ref_p.get(k) == v for k, v in key.items()):
                r = np.asarray(ref_p["probe"])
                probe["codebook_max_rel_vs_n1"] = float(np.abs(mine - r).max() / np.abs(r).max())
                probe["n1_probe"] = {"file": "profiles/" + os.path.basename(ref_path), "build": ref_p.get("build")}
        except (OSError, ValueError, KeyError):
            pass
        if world == 1:                                         # this IS the one-GPU run: (re)write the probe
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", os.path.basename(ref_path)), "w") as f:
                json.dump(dict(key, build=B.built_hash(), probe=[float(v) for v in mine]), f)
            probe["written"] = "gpurun_out/" + os.path.basename(ref_path)

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        is_exact = args.precision == "exact" and exact_has_screen(FEATURES, MAP_X * MAP_Y, wl["distance"])
        k_avg = (scr_ms / max(1, scr_n)) if is_exact else (bmu_ms / max(1, bmu_n))
        k_n = scr_n if is_exact else bmu_n
        rows_launch = my_rows * (bmu_n / max(1, k_n)) if is_exact else my_rows     # (the exact mode screens in passes)
        share = head_share if is_exact else 1.0
        flops_launch = KD2 * rows_launch * share
        achieved = flops_launch / (k_avg * 1e-3) / 1e12
        # algorithmic flops against the peak of the pipe the kernel runs on.
        # Block skipping: the flops of the blocks the screens RAN (the kernel's quality), never the full scan's: a launch
        # that proves most blocks empty is fast because it does less, not because the pipe runs faster -- the full scan's
        # algorithmic rate is reported beside it for what it is (rows per second are `value`)
        build_hash = B.built_hash()
        tr = pmc_traffic(args.workload, my_rows, args.precision, kernel_name, build_hash)
        out = {
            "metric": "samples/sec/epoch", "value": total_rows / (dt / args.steps), "unit": "samples/sec/epoch",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "codebooks_identical_on_all_ranks": ranks_agree,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16", "exact": "f16"}.get(args.precision, "bf16"), "data": "synthetic",
            "config": {"workload": "batch-SOM epoch, %dx%d map, %d features, %d Gaussian-blob rows %s resident "
                                   "in HBM (%s), one launch over all resident rows"
                                   % (MAP_X, MAP_Y, FEATURES, my_rows if args.scaling == "weak" else total_rows,
                                      "per GPU" if args.scaling == "weak" else "in total (split over the GPUs)",
                                      wl["label"] if args.scaling == "weak" else "BASELINE configs[3] rows, strong scaling"),
                       "map": [MAP_X, MAP_Y], "features": FEATURES, "rows_per_gpu": my_rows, "rows_total": total_rows,
                       "precision": args.precision, "distance": wl["distance"], "neighborhood": wl["neighborhood"],
                       "parity": ("BMUs, accumulators and codebook bit for bit those of the float32 parity mode (IEEE-half "
                                  "MFMA screen + float32 re-score of the candidate units, csrc/bmu_exact.hpp)" if is_exact else
                                  "float32 parity mode itself" if args.precision in ("f32", "exact") else
                                  "throughput mode: BMUs within the operand rounding of float32's, not equal to them"),
                       "update": "bucketed (segment sum by BMU + separable neighbourhood transform; exact algebra of "
                                 "the reference's g^T.x GEMM, xpysom.py:434-438; both forms timed: update_forms)",
                       "parallelism": "dp%d (sample shards, 1 all-reduce/epoch)" % world,
                       "epochs_per_sec": args.steps / dt, "build": build_hash,
                       "value_is": "mean of epochs %d..%d of a %d-epoch schedule (the %d warm-up epochs hold the schedule's full scans)"
                                   % (args.warmup, total - 1, total, args.warmup),
                       "codebook_crc32_after_run": w_crc},
            "roofline": {"bound": "mfma", "kernel": kernel_name + (" (IEEE-half distance GEMM of the exact mode's screen, fused "
                                                                   "argmin + per-group minima)" if is_exact else " (fused distance GEMM + argmin)"),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": tr["bytes"] if tr else None,
                         "traffic_source": ({"file": "profiles/" + tr["file"], "build": tr["build"],
                                             "same_build": tr["build"] == build_hash,
                                             "launches": tr.get("launches")} if tr else None),
                         "avg_launch_ms": k_avg, "launches": k_n, "rows_per_launch": rows_launch,
                         "flops_per_launch": flops_launch,
                         "executed_share_of_the_distance_gemm": share,
                         "full_scan_flops_per_launch": KD2 * rows_launch,
                         "full_scan_equivalent_tflops": KD2 * rows_launch / (k_avg * 1e-3) / 1e12,
                         "whole_bmu_search_ms_per_step": bmu_ms / max(1, bmu_n)},
            "ms_per_step_by_kernel": parts,
            "ms_per_step_by_kernel_pass": "separate untimed pass of %d epochs after the timed region (events around every kernel family)" % nb,
        }
        if exact_stats is not None:
            out["parity_mode"] = {"precision": "exact", "is_headline": True, "value": out["value"], "unit": out["unit"],
                                  "roofline_frac": achieved / peak, "rows_screened": exact_stats[0],
                                  "rows_through_float32_fallback_kernel": exact_stats[1], "screen_passes": exact_stats[2]}
        if thr is not None:
            out["throughput_mode"] = thr
        if schedule_block is not None:
            out.update(schedule_block)
            out["codebook_equal_to_f32_run"] = equal_f32
        if unstructured is not None:
            out["unstructured_rows"] = unstructured
        if survey is not None:
            out["survey_schedule"] = survey
        if variants is not None:
            out["data_variants"] = variants
        if queries is not None:
            out["queries"] = queries
        if full_scan is not None:
            full_scan["roofline_frac"] = KD2 * rows_launch / (full_scan["avg_launch_ms"] * 1e-3) / 1e12 / peak
            out["without_block_skipping"] = full_scan
            # SURVEY 8(d)'s figure: the ALGORITHMIC 2 N K D flop over the kernel that executes all of them
            out["roofline"]["frac_full_scan"] = full_scan["roofline_frac"]
            out["roofline"]["frac_is"] = ("executed work: the flops of the blocks the planned screens ran / their launch time; frac_full_scan: the "
                                          "algorithmic 2 N K D flop / the screen that runs every block (SOM_EXACT_SKIP=0, same epochs)")
        if rank_spread is not None:
            out.update(rank_spread)
        if probe is not None:
            out["strong_scaling_probe"] = probe
            # the same rows on N ranks train the one-GPU map to float32 summation order (one epoch from the seeded codebook)
            if world > 1 and "codebook_max_rel_vs_n1" in probe:
                assert probe["codebook_max_rel_vs_n1"] <= 2e-6, "strong scaling: the N-rank codebook left the one-GPU run's: %g" % probe["codebook_max_rel_vs_n1"]
        if batch is not None:
            out["roofline"]["batch65536"] = batch
        if modes is not None:
            out["precision_modes_at_batch65536"] = modes   # 'f32' = the parity mode itself, 'exact' = its BMUs at MFMA-half speed, 'f16' = IEEE half operands
        if update_forms is not None:
            out["update_forms"] = update_forms
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
