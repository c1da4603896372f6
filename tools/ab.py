#!/usr/bin/env python3
"""A/B of library builds on one box: `tools/ab.py [--reps 3] libA.so libB.so ... -- [bench.py args]`.
Runs bench.py once per build, alternating, `--reps` rounds; prints the per-kernel milliseconds of every run and
the per-build medians ("default" = the in-tree library)."""
import json
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
reps = 3
if args and args[0] == "--reps":
    reps = int(args[1]); args = args[2:]
split = args.index("--") if "--" in args else len(args)
libs, bench_args = args[:split], args[split + 1:]
res = {}
for r in range(reps):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["SOM_LIB_PATH"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline"] + bench_args,
                             env=env, capture_output=True, text=True, cwd=REPO)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-500:]); continue
        d = json.loads(line[-1])
        k = dict(d["ms_per_step_by_kernel"]); k["epoch"] = d["ms_per_step"]
        b = d["roofline"].get("batch65536")
        if b:
            k["b64k_epoch"] = b["epoch_ms"]; k["b64k_bmu"] = b["avg_launch_ms"]
            k["b64k_kron"] = b["ms_per_epoch_by_kernel"]["kron"]; k["b64k_seg"] = b["ms_per_epoch_by_kernel"]["segsum"]
            k["b64k_merge"] = b["ms_per_epoch_by_kernel"]["merge"]; k["b64k_prep"] = b["ms_per_epoch_by_kernel"]["prep"]
        res.setdefault(lib, []).append(k)
        print("%-40s %s" % (os.path.basename(lib), " ".join("%s=%.4f" % kv for kv in k.items())), flush=True)
print("--- medians")
for lib, runs in res.items():
    print("%-40s %s" % (os.path.basename(lib), " ".join("%s=%.4f" % (k, statistics.median(r[k] for r in runs)) for k in runs[0])))
