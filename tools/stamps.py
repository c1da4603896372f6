#!/usr/bin/env python3
"""In-kernel clock of the hot BMU kernels (MI355X_MICROARCH.md, DVFS give-back item 6): a DIAGNOSTIC build of the
library (-DSOM_STAMPS: the first wave of every workgroup stamps s_memtime and s_memrealtime around the kernel's scan loop
and leaves the differences in a buffer of their own) runs the kernel back to back for >= 2 s on random data, then the
median over workgroups of  d(s_memtime) / d(s_memrealtime) * 100 MHz  is the clock the chip holds INSIDE the kernel.

    python tools/stamps.py            # on the GPU box: builds xpysom_dask_amd/libsomhip_stamps.so, prints one JSON line
"""
import ctypes as C, json, os, subprocess, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child():
    from xpysom_dask_amd.engine import HipEngine
    from xpysom_dask_amd import _lib
    out = {}
    cases = [("bmu_bf16_k16_kernel<4, Bf16> (256x256x128, 1 Mi rows, bf16)", (256, 256, 128, 1 << 20, "bf16", "euclidean", 4096 * 3)),
             ("bmu_bf16_k16_kernel<4, F16, GM> (the exact mode's screen, passes of 262 144 rows)", (256, 256, 128, 1 << 20, "exact", "euclidean", 1024 * 3)),
             ("bmu_bf16_wide_kernel<25, Bf16> (512x512x784 shard, 250 000 rows, cosine)", (512, 512, 784, 250000, "bf16", "cosine", 8192))]
    for name, (X, Y, D, n, prec, dist, pairs) in cases:
        rs = np.random.RandomState(1)
        from xpysom_dask_amd.synthetic import gaussian_blobs
        data = gaussian_blobs(n, D)
        w = rs.rand(X, Y, D).astype(np.float32) * 2 - 1
        if dist == "cosine":
            data, w = np.abs(data), np.abs(w)
        e = HipEngine(X, Y, D, precision=prec, distance=dist)
        e.set_weights(w); e.set_data(data)
        lib = _lib.load()
        assert lib.som_debug_stamps(e._h, pairs, None) == 0, lib.som_last_error(e._h)
        t0 = time.time()
        reps = 0
        while time.time() - t0 < 2.5:                     # >= 2 s of back-to-back launches: the clock has settled
            e.epoch_accumulate(8.0, 0.3, True); reps += 1
        e.sync()
        buf = np.zeros((pairs, 2), dtype=np.uint64)
        assert lib.som_debug_stamps(e._h, pairs, buf.ctypes.data_as(C.c_void_p)) == 0
        ok = buf[:, 1] > 0
        clk = buf[ok, 0].astype(np.float64) / buf[ok, 1].astype(np.float64) * 0.1   # GHz
        out[name] = {"workgroups_stamped": int(ok.sum()), "in_kernel_clock_ghz_median": float(np.median(clk)),
                     "p10": float(np.percentile(clk, 10)), "p90": float(np.percentile(clk, 90)),
                     "scan_loop_us_median": float(np.median(buf[ok, 1]) / 100.0), "launch_reps": reps}
        lib.som_debug_stamps(e._h, 0, None)
        e.close()
    print(json.dumps({"in_kernel_clock": out, "method": "d(s_memtime)/d(s_memrealtime)*100MHz around the scan loop, first wave of every "
                      "workgroup, diagnostic build -DSOM_STAMPS, after >= 2.5 s of back-to-back launches on random data"}))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        from xpysom_dask_amd import build as B
        lib = os.path.join(REPO, "xpysom_dask_amd", "libsomhip_stamps.so")
        if B.built_hash(lib) != B.source_hash():           # (an in-tree build of these sources travels with the snapshot)
            B.build(force=True, verbose=False, extra=["-DSOM_STAMPS"], out=lib)
        env = dict(os.environ, SOM_LIB_PATH=lib)
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--child"], env=env))
