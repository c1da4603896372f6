"""sigma / learning-rate schedules of the host class (reference xpysom_dask/decays.py:4-65).

The return TYPE matters as much as the value: 'exponential' goes through numpy's exp/log
and hands back numpy.float64, which under NumPy >= 2 promotion makes the reference evaluate
the whole neighbourhood in float64; the other two stay Python floats (float32 neighbourhood).
XPySom.train passes that distinction to the engine as `neigh_f64`."""
import numpy as np


def asymptotic_decay(val0, valN, curr_iter, max_iter):
    return val0 / (1 + 2 * curr_iter / max_iter)


def exponential_decay(val0, valN, curr_iter, max_iter):
    target = 0.1 if valN == 0 else valN / val0
    return val0 * np.exp(curr_iter * (np.log(target) / max_iter))


def linear_decay(val0, valN, curr_iter, max_iter):
    if max_iter == 1:
        return val0
    return val0 + (valN - val0) * curr_iter / (max_iter - 1)


DECAY_FUNCTIONS = {
    "exponential": exponential_decay,
    "asymptotic": asymptotic_decay,
    "linear": linear_decay,
}
