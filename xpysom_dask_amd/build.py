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
TEST_HEADER = os.path.join(os.path.dirname(HERE), "include", "somhip_test.h")


def sources():
    """Every file the library is compiled from: csrc/* and the two headers."""
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.hpp"))) + [HEADER, TEST_HEADER]


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


# The wide BMU kernel (csrc/bmu_bf16_wide.hpp) places its LDS fragment reads and their counted waits by hand (inline
# assembly); that is only as good as the register allocation around it.  It was validated -- every `-m gpu` test, the
# fuzzers, SOM_VERIFY -- with this compiler; any other gets the plain C++ reads (-DSOM_WIDE_PLAIN_READS: the compiler's own
# waits, ~3 % slower at 512 x 512 x 784) unless SOM_WIDE_ASM=1 insists.
VALIDATED_HIPCC = "roc-7.2.0"


def hipcc_version(cc=None):
    try:
        return subprocess.run([cc or hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return ""


def wide_reads_flags(cc=None):
    if os.environ.get("SOM_WIDE_ASM") == "1" or VALIDATED_HIPCC in hipcc_version(cc):
        return []
    print("xpysom_dask_amd.build: hipcc is not the validated %s: building the wide kernel with plain LDS reads" % VALIDATED_HIPCC, file=sys.stderr)
    return ["-DSOM_WIDE_PLAIN_READS"]


def up_to_date():
    return os.path.exists(OUT) and built_hash(OUT) == source_hash()


def build(force=False, verbose=True, extra=(), out=None):
    """`extra`/`out` build experiment variants (e.g. -DSOM_STAMPS for tools/stamps.py) next to the product library."""
    if not force and not extra and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-command-line-argument",
           '-DSOM_SRC_HASH="%s"' % source_hash(),
           SRC, "-o", out or OUT, "-Wl,-rpath,/opt/rocm/lib"] + wide_reads_flags() + list(extra)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out or OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
