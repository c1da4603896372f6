#!/usr/bin/env python3
"""When does every workgroup of the exact mode's LISTED screen run its scan?  A diagnostic build (-DSOM_STAMPS=2: the first wave
of every workgroup leaves s_memrealtime at the start and at the end of the kernel's list walk) over the benchmark's schedule, one workgroup per tile (SOM_EXACT_QUEUE=0: the grid the work queue replaced);
after the chosen epochs: the walks' durations (how uneven are the tiles' lists) and the launch's occupancy over time (how much of
the launch is a tail of few long walks).
    python tools/wg_timeline.py          # on the GPU box; WT_EPOCHS=6,12,20"""
import ctypes as C, json, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child():
    from xpysom_dask_amd.engine import HipEngine
    from xpysom_dask_amd import _lib
    from xpysom_dask_amd.decays import exponential_decay
    from xpysom_dask_amd.synthetic import gaussian_blobs
    # (WT_SIDE / WT_D / WT_ROWS / WT_T: another shape -- e.g. 512 / 784 / 250000 / 12: the wide screen's lists, 250 rows to a tile)
    X = Y = int(os.environ.get("WT_SIDE", "256")); D = int(os.environ.get("WT_D", "128"))
    N = int(os.environ.get("WT_ROWS", str(1 << 20))); T = int(os.environ.get("WT_T", "25"))
    epochs = [int(v) for v in os.environ.get("WT_EPOCHS", "6,12,20,24").split(",")]
    data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1; w /= np.linalg.norm(w, axis=-1, keepdims=True)
    e = HipEngine(X, Y, D, precision="exact"); e.set_weights(w.astype(np.float32)); e.set_data(data); e.sync()
    lib = _lib.load()
    wgs = (N + 255) // 256
    for t in range(T):
        sig, eta = exponential_decay(min(X, Y) / 2, 1, t, T), exponential_decay(0.5, 0.01, t, T)
        if t in epochs:
            assert lib.som_debug_stamps(e._h, 4 * wgs, None) == 0, lib.som_last_error(e._h)
        e.epoch(sig, eta, True); e.sync()
        if t in epochs:
            buf = np.zeros((4 * wgs, 2), dtype=np.uint64)
            assert lib.som_debug_stamps(e._h, 4 * wgs, buf.ctypes.data_as(C.c_void_p)) == 0
            lib.som_debug_stamps(e._h, 0, None)
            ok = buf[:, 1] > 0
            b, en = buf[ok, 0].astype(np.int64), buf[ok, 1].astype(np.int64)
            t0 = b.min(); b = (b - t0) / 100.0; en = (en - t0) / 100.0          # us
            dur = en - b
            span = en.max()
            grid = np.linspace(0, span, 41)
            occ = [(int(((b <= g) & (en > g)).sum())) for g in grid]
            print(json.dumps({"epoch": t, "workgroups": int(ok.sum()), "launch_span_us": round(float(span), 1),
                              "walk_us": {"p10": round(float(np.percentile(dur, 10)), 2), "median": round(float(np.median(dur)), 2),
                                          "p90": round(float(np.percentile(dur, 90)), 2), "max": round(float(dur.max()), 2),
                                          "sum_over_768_slots": round(float(dur.sum() / 768), 1), "sum": round(float(dur.sum()), 1)},
                              "workgroups_in_their_walk_at_40_instants": occ}), flush=True)
    e.close()


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        from xpysom_dask_amd import build as B
        lib = os.path.join(REPO, "xpysom_dask_amd", "libsomhip_timeline.so")
        if B.built_hash(lib) != B.source_hash():           # (an in-tree build of these sources travels with the snapshot)
            B.build(force=True, verbose=False, extra=["-DSOM_STAMPS=2"], out=lib)
        # (one workgroup per tile: under the work queue a workgroup walks many items and only its last one would be stamped)
        env = dict(os.environ, SOM_LIB_PATH=lib, SOM_TEST_HOOKS="1", SOM_EXACT_QUEUE="0")
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=env))
