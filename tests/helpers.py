"""Shared helpers for the test-suite (oracle = checker, lpbox_hip = product)."""
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def lp_instances(name):
    return O.load_lp_batch(os.path.join(GOLDEN, name))


def make_oracle(I, order=O.ORDER_EIGEN, T=512, positions=None, npos=0, row_split=None):
    s = O.LpOracle(0, order=order, T=T, positions=positions, npos=npos, row_split=row_split)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"))
    s.solve_init()
    return s


def scripted_fix_vec(xiters, lo=0.02, hi=0.98, last=20):
    """Deterministic stand-in for the trained policy (LP/trainer.py:101-135,216-252): fix a live variable to 1 (0) when
    its last `last` iterates are all > hi (< lo); everything else stays free (-1).  Returns (vec, num)."""
    tail = xiters[:, -last:]
    vec = -np.ones(xiters.shape[0])
    vec[np.all(tail > hi, axis=1)] = 1.0
    vec[np.all(tail < lo, axis=1)] = 0.0
    num = int(np.sum(vec != -1))
    return vec, num


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def oracle_like(g, I):
    """CPU oracle configured with the reduction tree of the HIP solver `g` (threads + storage positions)."""
    return oracle_for(g.batch if hasattr(g, "batch") else g, 0, I)


def oracle_for(batch, idx, I):
    cfg = batch.config()
    return make_oracle(I, O.ORDER_GPU, cfg["threads"], batch.layout(idx), cfg["threads"] * cfg["elems_per_thread"],
                       batch.row_split(idx))
