"""Build libsomhip.so in-tree with hipcc for gfx950 (no JIT cache, no torch extension).

    python -m xpysom_dask_amd.build

The library carries a hash of every source it was built from (`som_version()` ends in `src:<hash>`);
`up_to_date()` and `_lib.load()` compare it with the sources in the tree, so an edit to any kernel
header can neither be skipped by `build(force=False)` nor be benchmarked through a stale binary.
"""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "somhip.hip")
OUT = os.path.join(HERE, "libsomhip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "somhip.h")


def sources():
    """Every file the library is compiled from: csrc/* and the public header."""
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.hpp"))) + [HEADER]


def source_hash():
    h = hashlib.sha256()
    for p in sources():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash(path=None):
    """The source hash baked into a built library (read from the file, no dlopen), or None."""
    path = path or OUT
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    tag = b"somhip-src:"
    i = blob.find(tag)
    if i < 0:
        return None
    return blob[i + len(tag):i + len(tag) + 16].decode("ascii", "replace")


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def up_to_date():
    return os.path.exists(OUT) and built_hash(OUT) == source_hash()


def build(force=False, verbose=True, extra=(), out=None):
    """`extra`/`out` build experiment variants (e.g. -DSOM_STAMPS for tools/stamps.py) next to the product library."""
    if not force and not extra and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-command-line-argument",
           '-DSOM_SRC_HASH="%s"' % source_hash(),
           SRC, "-o", out or OUT, "-Wl,-rpath,/opt/rocm/lib"] + list(extra)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
