"""The exact mode's block-skipping POLICY (csrc/exact_policy.hpp) without a GPU: pure functions of measured costs, reached through
the library's test hook som_policy_eval (include/somhip_test.h).  What is pinned: each decision's direction in its cost terms, its
behaviour before anything is measured, and the benchmark's own recorded numbers (profiles/r05_*: epoch 1 of the schedule is
committed at a forecast of 0.58, a random codebook's 1.0 is declined; level 2 pays late in the schedule and not on the smooth map
after the first merge)."""
import ctypes as C

import pytest

from xpysom_dask_amd import _lib

NAMES = ("full_total", "full_screen", "plan_total", "plan_over", "plan_over_scout", "blk_ms", "l2_ms_group", "l2_ratio", "sort_ms")
BPR = 1024 * 4 / 256.0            # 16-unit blocks per row at 256 x 256 units: 1024 groups x 4 / 256-row tiles


def ev(which, costs, *args):
    lib = _lib.load()
    c = dict(full_total=0.0, full_screen=0.0, plan_total=0.0, plan_over=0.0, plan_over_scout=0.0, blk_ms=0.0, l2_ms_group=0.0, l2_ratio=1.0, sort_ms=0.0)
    c.update(costs)
    ca = (C.c_double * 9)(*[c[k] for k in NAMES])
    aa = (C.c_double * 4)(*(list(args) + [0.0] * (4 - len(args))))
    out = C.c_int32(-1)
    assert lib.som_policy_eval(which, ca, aa, C.byref(out)) == 0
    return bool(out.value)


# the benchmark's own measurements (per row of a 1 Mi-row launch, ms): profiles/r05_* schedule traces
FULL = dict(full_total=14.6 / 2 ** 20, full_screen=13.2 / 2 ** 20)
PLANNED = dict(FULL, plan_total=1.3 / 2 ** 20, plan_over=1.05 / 2 ** 20, plan_over_scout=3.0 / 2 ** 20, blk_ms=1.0e-6, l2_ms_group=0.8e-6, l2_ratio=0.5, sort_ms=0.6 / 2 ** 20)


def test_commit_a_scouted_plan():
    assert not ev(0, {}, 1.0, BPR) and not ev(0, {}, 0.81, BPR) and ev(0, {}, 0.8, BPR) and ev(0, {}, 0.02, BPR)   # nothing priced: 0.8
    assert ev(0, FULL, 0.58, BPR)                            # epoch 1 of the schedule: committed (it costs what the scan costs)
    assert not ev(0, FULL, 1.0, BPR) and not ev(0, FULL, 0.9, BPR)
    assert ev(0, PLANNED, 0.04, BPR) and not ev(0, PLANNED, 0.95, BPR)
    # monotone in the forecast, in the block time and in the overhead
    shares = [s / 100.0 for s in range(0, 101)]
    got = [ev(0, PLANNED, s, BPR) for s in shares]
    assert got == sorted(got, reverse=True) and got[0] and not got[-1]
    assert ev(0, PLANNED, 0.5, BPR) and not ev(0, dict(PLANNED, blk_ms=2.0e-6), 0.5, BPR)
    assert not ev(0, dict(PLANNED, plan_over_scout=14.0 / 2 ** 20), 0.04, BPR)


def test_level_two():
    # from a sample: the smooth map after the first merge (0.7491 -> 0.7238 of the blocks) does not pay, the late map (0.25 -> 0.07) does
    assert not ev(1, FULL, 0.7238, 0.7491, BPR) and ev(1, FULL, 0.07, 0.25, BPR)
    assert ev(1, {}, 0.5, 0.7, BPR) and not ev(1, {}, 0.69, 0.7, BPR)            # nothing priced: a ratio below 0.85
    assert ev(1, FULL, 0.0, 0.0, BPR)                                            # no sample of level 1: on
    # measured: (1 - ratio) * 4 * block time against its own time per kept group
    assert ev(2, PLANNED, 0.01, 0.02) and not ev(2, dict(PLANNED, l2_ratio=0.9), 0.01, 0.02)
    assert not ev(2, dict(PLANNED, l2_ms_group=3.0e-6), 0.01, 0.02)
    assert ev(2, {}, 0.02, 0.2) and not ev(2, {}, 0.19, 0.2)                     # before both are measured: round 4's rule


def test_sort_paid_idle_plans_and_the_scout():
    assert ev(3, PLANNED, 0.04, 0.02, BPR, 8) and not ev(3, PLANNED, 0.0201, 0.02, BPR, 8)
    assert ev(3, PLANNED, 0.0215, 0.02, BPR, 64) and not ev(3, PLANNED, 0.0215, 0.02, BPR, 1)   # the epochs the order will serve count
    assert ev(3, {}, 0.1, 0.09, BPR, 8) and not ev(3, {}, 0.1, 0.095, BPR, 8)                   # unmeasured: the share fell by 7 %
    assert not ev(4, PLANNED, 0.02) and not ev(4, dict(PLANNED, plan_total=20.0 / 2 ** 20), 0.3)   # few blocks run: never idle
    assert ev(4, dict(PLANNED, plan_total=14.5 / 2 ** 20), 0.9) and not ev(4, dict(PLANNED, plan_total=10.0 / 2 ** 20), 0.9)
    assert ev(4, {}, 0.98) and not ev(4, {}, 0.96)
    assert not ev(5, PLANNED, 0.2, 0.9, BPR)                                      # too few wins
    assert ev(5, PLANNED, 0.6, 0.5, BPR) and not ev(5, PLANNED, 0.6, 0.03, BPR)   # ... and halving the screen must pay for it
    assert not ev(6, {}, 30000, 65536, 128) and ev(6, {}, 40000, 65536, 128) and not ev(6, {}, 1e6, 4096, 32)


def test_unknown_decision_is_refused():
    lib = _lib.load()
    out = C.c_int32(0)
    ca, aa = (C.c_double * 9)(), (C.c_double * 4)()
    assert lib.som_policy_eval(99, ca, aa, C.byref(out)) != 0
    assert lib.som_policy_eval(0, None, aa, C.byref(out)) != 0
