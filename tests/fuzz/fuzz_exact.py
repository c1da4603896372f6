"""precision='exact' against precision='f32' on random configurations: IDENTICAL ids, row for row, from the epoch path and
from the query path -- never "near".  FUZZ_WIDE=1: maps of 64..110 a side, 129..800 features, euclidean and cosine (the
wide screen); otherwise maps 1..100 a side, 1..128 features (and past 128, where 'exact' is served by the
float32 kernels), 1..5000 rows, magnitudes 1e-3..1e3, zero rows, duplicated units, and the three codebook shapes that
decide how hard the screen's job is: random units (one candidate group per row), a smooth sheet (hundreds of
near-ties per row: the early-schedule state) and clusters of units a few float32 ulps apart (every row re-scored)."""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine

F32 = np.float32
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
WIDE = os.environ.get("FUZZ_WIDE", "0") == "1"
bad = 0
fb_total = rows_total = 0
t0 = time.time()
for case in range(n_cases):
    side = int(os.environ.get("FUZZ_MAXSIDE", "100"))
    X, Y = int(rs.randint(1, side + 1)), int(rs.randint(1, side + 1))
    D = int(rs.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 96, 100, 127, 128, 128, 128, 130, 200]))
    n = int(rs.choice([1, 2, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 3000, 5000]))
    dist = "euclidean"
    if WIDE:                                                 # the wide screen: >= 4096 units, 129..800 features
        X, Y = int(rs.randint(64, 111)), int(rs.randint(64, 111))
        D = int(rs.choice([129, 130, 160, 161, 200, 256, 257, 300, 512, 600, 784, 800]))
        n = int(rs.choice([1, 2, 31, 32, 33, 255, 256, 257, 1000, 2500]))
        dist = str(rs.choice(["euclidean", "cosine"]))
    if rs.rand() < 0.5:                                      # map sides multiples of 8: the images in patch order
        X, Y = 8 * ((X + 7) // 8), 8 * ((Y + 7) // 8)
    kind = str(rs.choice(["random", "random", "sheet", "sheet", "clusters"]))
    data = O.gaussian_blobs(n, D, seed=case)
    if kind == "random":
        w = O.default_codebook(X, Y, D, case + 1).astype(F32) * 3
    elif kind == "sheet":                                    # a smooth, nearly flat sheet through the data: many near-ties
        a, b, c = rs.randn(D), rs.randn(D), rs.randn(D)
        ii, jj = np.meshgrid(np.arange(X) / max(X, 1), np.arange(Y) / max(Y, 1), indexing="ij")
        amp = float(rs.choice([1e-3, 1e-2, 0.3]))
        w = (data.mean(0) + amp * (ii[..., None] * b + jj[..., None] * c) + 1e-6 * rs.randn(X, Y, D) + 0 * a).astype(F32)
    else:                                                    # units a few ulps apart
        base = data[rs.randint(0, n)]
        w = np.repeat(base[None, None, :], X, 0).repeat(Y, 1).astype(F32)
        w = (w * (1.0 + rs.randint(-6, 7, size=(X, Y, 1)) * 2.0 ** -22)).astype(F32)
    sx, sw = F32(rs.choice([1e-3, 1.0, 1.0, 1e3])), F32(rs.choice([1e-2, 1.0, 1.0, 1e2]))
    data, w = data * sx, w * sw
    if n > 4 and rs.rand() < 0.3:
        data[rs.randint(0, n, size=max(1, n // 50))] = 0
    if X * Y > 3 and rs.rand() < 0.3:
        wf0 = w.reshape(-1, D)
        wf0[rs.randint(0, X * Y)] = wf0[rs.randint(0, X * Y)]
    if rs.rand() < 0.1 and n > 2:
        data[rs.randint(0, n)] = np.nan
    sig = float(rs.choice([max(min(X, Y) / 2, 1.0), 1.0, 2.5]))
    w2 = (w[::-1, ::-1] * F32(rs.choice([1.0, 0.5, 3.0]))).copy()   # (another codebook of the same kind: the units change places)
    rs_cut = float(rs.rand())
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):   # (replay one case of a seed)
        continue
    if os.environ.get("FUZZ_DUMP"):
        np.savez(os.environ["FUZZ_DUMP"], data=data, w=w, w2=w2, sig=sig, dims=np.array([X, Y, D, n]))
    try:
        out = {}
        for p in ("f32", "exact"):
            e = HipEngine(X, Y, D, precision=p, distance=dist)
            e.set_weights(w); e.set_data(data)
            e.epoch_accumulate(sig, 0.5, True)
            num, den, bmu = e.epoch_fetch()
            q = e.bmu(data[: min(n, 700)])
            # a second and third epoch on the same resident rows: the exact mode now seeds its thresholds from the previous
            # epoch's BMUs -- after a merge (the usual case: most BMUs stay) and after a codebook the old BMUs say nothing about
            e.epoch_merge()
            e.epoch_accumulate(sig * 0.7, 0.4, True)
            bmu2 = e.epoch_fetch()[2]
            e.set_weights(w2)
            e.epoch_accumulate(sig, 0.5, True)
            bmu3 = e.epoch_fetch()[2]
            # ... and a streamed epoch (chunks have no last BMU: under SOM_EXACT_SKIP=2 every chunk goes through the scout)
            cut = int(rs_cut * n)
            e.stream_epoch_accumulate([data[:cut], data[cut:]] if 0 < cut < n else [data], sig, 0.5, True)
            snum, sden = e.epoch_fetch(want_bmu=False)[:2]
            out[p] = (bmu, q, num, den, bmu2, bmu3, snum, sden)
            if p == "exact":
                r, fb, _ = e.exact_stats()
                rows_total += r; fb_total += fb
            e.close()
        a, b = out["f32"], out["exact"]
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True) \
            and np.array_equal(a[3], b[3], equal_nan=True) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) \
            and np.array_equal(a[6], b[6], equal_nan=True) and np.array_equal(a[7], b[7], equal_nan=True)
        detail = "%d epoch rows, %d query rows, %d / %d rows of the seeded epochs differ; streamed sums equal %s" % (
            (a[0] != b[0]).sum(), (a[1] != b[1]).sum(), (a[4] != b[4]).sum(), (a[5] != b[5]).sum(),
            np.array_equal(a[6], b[6], equal_nan=True) and np.array_equal(a[7], b[7], equal_nan=True))
    except Exception as ex:                      # noqa: BLE001
        ok, detail = False, "EXC " + repr(ex)[:200]
    if not ok:
        bad += 1
        print(f"FAIL case {case}: {X}x{Y}x{D} {dist} n={n} {kind} scale {sx}/{sw}: {detail}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t0:.1f} s; float32 fallback kernel: {fb_total} of {rows_total} rows")
