#!/usr/bin/env python3
"""Fresh-process stress of the FIRST epoch of a fresh engine.

Round 1 recorded ONE run (of ~55 000 fuzz cases, on a freshly started box) in which the first epoch of a fresh
engine had whole 128-row blocks of BMUs wrong; the response (blocking host<->device copies) was a guess.  This
tool tries to reproduce that signature where it would live: the first calls of a NEW process.

    python tools/fresh_process_stress.py [--procs 300] [--parallel 4] [--async-copies 0|1] [--log FILE]

The parent never touches the GPU: it computes the expected BMUs of each case once with the oracle (test
infrastructure, CPU), then starts `--procs` child processes, `--parallel` at a time.  Each child: import the
engine -> som_create -> som_set_weights -> som_set_data -> FIRST som_epoch_accumulate -> fetch BMUs and the
accumulators -> compare with the oracle's BMUs (float32 precision: bit-exact; bf16: near-best) and with a SECOND
epoch of the same process (must be bit-identical BMUs).  `--async-copies 1` sets SOM_ASYNC_COPIES=1, round 1's
original copy path, so the hypothesis can be confirmed or killed.
"""
import argparse
import json
import os
os.environ.setdefault("SOM_TEST_HOOKS", "1")   # (the library reads its developer switches only under this one)
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# (X, Y, D, rows, precision): a 256-row-block shape like the failing family, plus the other kernel families
CASES = [
    (16, 16, 12, 3000, "f32"), (24, 20, 32, 2561, "f32"), (40, 33, 7, 1000, "f32"), (13, 9, 130, 700, "f32"),
    (16, 16, 12, 3000, "bf16"), (24, 20, 32, 2561, "bf16"), (32, 32, 128, 4096, "bf16"), (13, 9, 130, 700, "bf16"),
    (64, 64, 200, 1500, "bf16"), (24, 20, 32, 2561, "f16"), (64, 66, 160, 1200, "f16"),   # wide kernel, IEEE half
    (24, 20, 32, 2561, "exact"), (64, 64, 128, 4096, "exact"),
    (64, 64, 200, 1500, "exact"), (72, 64, 784, 900, "exact"),                     # the wide screen + two-round re-score
]


def child(case_file, idx):
    t0 = time.time()
    from xpysom_dask_amd.engine import HipEngine
    z = np.load(case_file)
    X, Y, D, n = (int(v) for v in z["shape"])
    prec = str(z["precision"])
    data, w, ref = z["data"], z["w"], z["bmu"]
    e = HipEngine(X, Y, D, precision=prec)
    e.set_verify(256)                                  # the canary rides along: an independent float32 re-score of 256 rows per launch
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(3.0, 0.5, True)                 # the FIRST epoch of this process
    num1, den1, bmu1 = e.epoch_fetch()
    e.epoch_accumulate(3.0, 0.5, True)
    num2, den2, bmu2 = e.epoch_fetch()
    q = e.bmu(data)
    out = {"idx": idx, "case": [X, Y, D, n, prec], "sec": round(time.time() - t0, 2)}
    wrong = np.flatnonzero(bmu1 != ref)
    if len(wrong):                                     # picks may differ from the host BLAS's only by near-ties
        x64, w64 = data[wrong].astype(np.float64), w.reshape(-1, D).astype(np.float64)
        dd = (x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :]
        scale = (np.linalg.norm(x64, axis=1) + np.linalg.norm(w64, axis=1).max()) ** 2
        tol = {"f32": 2.0 ** -18, "exact": 2.0 ** -18, "bf16": 2.0 ** -6, "f16": 2.0 ** -9}[prec]
        wrong = wrong[dd[np.arange(len(wrong)), bmu1[wrong]] > dd.min(1) + tol * scale]
    out["wrong_vs_oracle"] = int(len(wrong))
    out["first_vs_second_epoch"] = int((bmu1 != bmu2).sum())
    out["first_vs_query"] = int((bmu1 != q).sum()) if prec in ("f32", "exact") else int((bmu1 != q).sum() > n // 50)
    out["den_sum_rel"] = float(abs(den1.sum() - den2.sum()) / max(abs(den2.sum()), 1e-30))
    if len(wrong):
        out["wrong_rows_head"] = [int(v) for v in wrong[:16]]
        out["wrong_128_blocks"] = sorted({int(v) // 128 for v in wrong})[:32]
    out["ok"] = bool(len(wrong) == 0 and out["first_vs_second_epoch"] == 0 and out["first_vs_query"] == 0
                     and out["den_sum_rel"] < 1e-5)
    print(json.dumps(out), flush=True)
    return 0 if out["ok"] else 3


def parent(args):
    from oracle import som_oracle as O
    tmp = tempfile.mkdtemp(prefix="somstress_")
    files = []
    for c, (X, Y, D, n, prec) in enumerate(CASES):
        data = O.gaussian_blobs(n, D, seed=700 + c)
        w = O.default_codebook(X, Y, D, 800 + c).astype(np.float32) * 3
        ref = O.bmu_ids(data, w.reshape(-1, D), "euclidean", O.row_sq(w.reshape(-1, D))).astype(np.int32)
        f = os.path.join(tmp, "case%d.npz" % c)
        np.savez(f, shape=np.array([X, Y, D, n]), precision=np.array(prec), data=data, w=w, bmu=ref)
        files.append(f)
    env = dict(os.environ)
    if args.async_copies:
        env["SOM_ASYNC_COPIES"] = "1"
    else:
        env.pop("SOM_ASYNC_COPIES", None)
    log = open(args.log, "w") if args.log else None

    def emit(line):
        print(line, flush=True)
        if log:
            log.write(line + "\n")
            log.flush()

    emit("# fresh_process_stress: procs=%d parallel=%d SOM_ASYNC_COPIES=%d" % (args.procs, args.parallel, args.async_copies))
    running, done, bad, t0, nxt = [], 0, 0, time.time(), 0
    while done < args.procs:
        while nxt < args.procs and len(running) < args.parallel:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", files[nxt % len(files)], "--idx", str(nxt)]
            running.append((nxt, subprocess.Popen(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
            nxt += 1
        still = []
        for i, p in running:
            if p.poll() is None:
                still.append((i, p))
                continue
            so, se = p.communicate()
            done += 1
            line = [ln for ln in so.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                bad += 1
                emit("FAIL proc %d rc=%d %s %s" % (i, p.returncode, line[-1] if line else so[-300:], se[-300:].replace("\n", " | ")))
            elif done % 25 == 0 or args.verbose:
                emit("ok   proc %d %s" % (i, line[-1]))
        running = still
        time.sleep(0.02)
    emit("# %d fresh processes, %d failures, %.0f s, SOM_ASYNC_COPIES=%d" % (args.procs, bad, time.time() - t0, args.async_copies))
    return 1 if bad else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=300)
    ap.add_argument("--parallel", type=int, default=4)       # the GPU box allows at most 6 processes on the card
    ap.add_argument("--async-copies", type=int, default=0)
    ap.add_argument("--log", default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--child", default=None)
    ap.add_argument("--idx", type=int, default=0)
    a = ap.parse_args()
    sys.exit(child(a.child, a.idx) if a.child else parent(a))
