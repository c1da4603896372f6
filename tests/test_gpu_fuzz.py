"""Randomised sweeps of the C ABI against the oracle (tests/fuzz/*.py), a fixed seed and a modest case count
each; the scripts take `[seed] [cases]` for longer runs."""
import os
import subprocess
import sys

import pytest

from tests.conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,cases", [("fuzz_shapes.py", 250), ("fuzz_paths.py", 150), ("fuzz_train.py", 150), ("fuzz_infer.py", 150),
                                          ("fuzz_exact.py", 300)])
def test_fuzz(script, cases):
    r = subprocess.run([sys.executable, os.path.join("tests", "fuzz", script), "12345", str(cases)], cwd=REPO,
                       capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert "%d cases, 0 failures" % cases in r.stdout, tail


def test_fuzz_exact_wide():
    """the exact mode's wide screen (129..800 features, >= 4096 units, euclidean and cosine)"""
    env = dict(os.environ, FUZZ_WIDE="1")
    r = subprocess.run([sys.executable, os.path.join("tests", "fuzz", "fuzz_exact.py"), "777", "120"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert "120 cases, 0 failures" in r.stdout, tail
