"""Build libsomhip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m xpysom_dask_amd.build
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "somhip.hip")
OUT = os.path.join(HERE, "libsomhip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in
        ("somhip.hip", "som_common.hpp", "bmu_f32.hpp", "bmu_bf16.hpp", "update.hpp")]
DEPS.append(os.path.join(os.path.dirname(HERE), "include", "somhip.h"))


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=True, extra=(), out=None):
    """`extra`/`out` build experiment variants (e.g. -DSOM_K16_SB=8) next to the product library."""
    if not force and not extra and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-munsafe-fp-atomics", "-Wall", "-Wno-unused-command-line-argument",
           SRC, "-o", out or OUT, "-Wl,-rpath,/opt/rocm/lib"] + list(extra)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
