"""Path-equivalence fuzz (f32 and exact): for a random configuration the same epoch must come out of
(1) resident rows, (2) rows streamed in random chunks from pageable memory, (3) rows streamed from two pinned
buffers, (4) contiguous shards accumulated separately and added (what the all-reduce relies on), and
(5) the teacher-forced call with the engine's own BMUs."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from oracle import som_oracle as O
from xpysom_dask_amd.engine import HipEngine

F32 = np.float32
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


for case in range(n_cases):
    X, Y = int(rs.randint(1, 30)), int(rs.randint(1, 30))
    D = int(rs.choice([1, 3, 16, 33, 128, 130, 257]))
    n = int(rs.choice([7, 64, 300, 1025, 4000]))
    prec = str(rs.choice(["f32", "exact"]))
    neigh = str(rs.choice(["gaussian", "mexican_hat", "bubble", "triangle"]))
    topo = "rectangular" if neigh == "triangle" else str(rs.choice(["rectangular", "hexagonal"]))
    data = O.gaussian_blobs(n, D, seed=case + 77)
    w = O.default_codebook(X, Y, D, case).astype(F32) * 2
    sig, eta, wide = float(rs.choice([1.0, 2.5, 6.0])), 0.3, bool(rs.randint(2))
    msgs = []
    try:
        e = HipEngine(X, Y, D, precision=prec, neighborhood=neigh, topology=topo)
        e.set_weights(w); e.set_data(data)
        e.epoch_accumulate(sig, eta, wide)
        num, den, bmu = e.epoch_fetch()
        tol = 3e-6                                   # denominators (positive sums)
        ntol = 5e-5                                  # numerators: signed sums cancel, the summation order differs by path
        stream_tol = tol                             # (both modes return float32's BMUs whatever the launch's row set)
        # (2) pageable chunks
        cuts = sorted(set([0, n] + [int(c) for c in rs.randint(0, n + 1, size=int(rs.randint(0, 5)))]))
        e2 = HipEngine(X, Y, D, precision=prec, neighborhood=neigh, topology=topo); e2.set_weights(w)
        e2.stream_epoch_accumulate((data[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a), sig, eta, wide)
        n2, d2, _ = e2.epoch_fetch()
        if rel(n2, num) > max(stream_tol, ntol) or rel(d2, den) > stream_tol: msgs.append("pageable stream %.1e %.1e" % (rel(n2, num), rel(d2, den)))
        # (3) pinned double buffer
        step = max(1, n // 3)
        bufs = [e2.pinned_empty((step, D)), e2.pinned_empty((step, D))]
        def chunks():
            for i, a in enumerate(range(0, n, step)):
                b = min(n, a + step); buf = bufs[i & 1]; buf[:b - a] = data[a:b]; yield buf[:b - a]
        e2.stream_epoch_accumulate(chunks(), sig, eta, wide)
        n3, d3, _ = e2.epoch_fetch()
        if rel(n3, num) > max(stream_tol, ntol) or rel(d3, den) > stream_tol: msgs.append("pinned stream %.1e %.1e" % (rel(n3, num), rel(d3, den)))
        # (4) shards
        parts = int(rs.choice([2, 3, 5]))
        tn, td = np.zeros_like(num, dtype=np.float64), np.zeros_like(den, dtype=np.float64)
        moved = 0
        for r in range(parts):
            lo, hi = (n * r) // parts, (n * (r + 1)) // parts
            if hi <= lo: continue
            e2.set_data(data[lo:hi]); e2.epoch_accumulate(sig, eta, wide)
            pn, pd, pb = e2.epoch_fetch(); tn += pn; td += pd; moved += int((pb != bmu[lo:hi]).sum())
        if moved or rel(tn, num) > ntol or rel(td, den) > tol:
            msgs.append("shards moved=%d %.1e %.1e" % (moved, rel(tn, num), rel(td, den)))
        # (5) teacher-forced
        e.epoch_accumulate_forced(bmu, sig, eta, wide)
        n5, d5, b5 = e.epoch_fetch()
        if rel(n5, num) > ntol or rel(d5, den) > tol or not np.array_equal(b5, bmu): msgs.append("forced")
    except Exception as ex:                      # noqa: BLE001
        msgs.append("EXC " + repr(ex)[:200])
    if msgs:
        bad += 1
        try:                                         # diagnostics: which engine disagrees with the oracle's BMUs?
            ref = O.bmu_ids(data, w.reshape(-1, D))
            e.epoch_accumulate(sig, eta, wide); again = e.epoch_fetch()[2]
            msgs.append("[diag: first-epoch BMUs != oracle in %d rows, re-run != oracle in %d, blocks of 128 hit: %s]"
                        % (int((bmu != ref).sum()), int((again != ref).sum()), sorted(set(np.flatnonzero(bmu != ref) // 128))[:16]))
        except Exception as ex:                      # noqa: BLE001
            msgs.append("[diag failed: %r]" % (ex,))
        print(f"FAIL case {case}: {X}x{Y}x{D} n={n} {prec} {neigh} {topo} sig={sig} wide={wide}: {'; '.join(msgs)}", flush=True)
print(f"{n_cases} cases, {bad} failures, {time.time()-t0:.1f} s")
