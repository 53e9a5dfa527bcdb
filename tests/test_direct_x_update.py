"""The opt-in DIRECT x-update (lpbox_set_x_update) has no reference counterpart: the reference solves its
x-update by Jacobi-PCG to 1e-3 (LPcpp:251-335, :894).  What can be pinned on the CPU: the C oracle's mirror of the kernel arithmetic
(closed-form inverse over the column-disjoint rows + Woodbury over the rest) solves the SAME linear system as a plain dense solve
(oracle/lpbox_numpy.py, written without that algebra), for any admissible row split, and the ADMM around it is unchanged."""
import numpy as np
import pytest

from helpers import lp_instances
from oracle import oracle as O
from oracle.lpbox_numpy import NumpyLpBox


def greedy_split(I, ascending=True):
    """Rows with pairwise disjoint columns -> -1, the rest get dense indices (what lpbox_set_x_update does on the host)."""
    n, l, cp, ri = I["n"], I["l"], I["colptr"], I["rowidx"]
    rows = [[] for _ in range(l)]
    for j in range(n):
        for e in range(cp[j], cp[j + 1]):
            rows[ri[e]].append(j)
    order = sorted(range(l), key=lambda r: len(rows[r]) if ascending else -len(rows[r]))
    used, is_d = np.zeros(n, bool), np.zeros(l, bool)
    for r in order:
        if not used[rows[r]].any():
            is_d[r] = True
            used[rows[r]] = True
    g = -np.ones(l, np.int32)
    g[~is_d] = np.arange((~is_d).sum())
    return g


def direct_oracle(I, rows=None):
    s = O.LpOracle(0, order=O.ORDER_EIGEN, x_update="direct", direct_rows=rows)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"))
    s.solve_init()
    return s


@pytest.mark.parametrize("name,idx", [("lp_20_60_seed0.npz", 0), ("lp_20_60_seed0.npz", 3), ("lp_100_500_seed0.npz", 1)])
def test_mirror_solves_the_same_system_as_a_dense_solve(name, idx):
    I = lp_instances(name)[idx]
    ref = NumpyLpBox(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"), x_update="direct")
    ref.solve_init()
    splits = [None, greedy_split(I, True), greedy_split(I, False)]
    assert (splits[1] < 0).any()                      # the auction's XOR rows are found
    orcs = [direct_oracle(I, g) for g in splits]
    for w in range(3):
        ref.solve_iter(w * 20, (w + 1) * 20)
        for o in orcs:
            o.solve_iter(w * 20, (w + 1) * 20)
            assert np.abs(o.vec("x") - ref.x).max() < 1e-9, w
            assert o.total_pcg_iters == 0
    assert orcs[1].total_outer_iters == 60


def test_direct_mode_converges_like_the_pcg_mode_on_the_small_batch():
    """Different x-update, same ADMM: on the 20/60 fixtures both modes stop by a reference stop rule with feasible roundings and
    objectives of the same size (the direct mode is NOT expected to reproduce the PCG iterates)."""
    insts = lp_instances("lp_20_60_seed0.npz")
    obj = {"pcg": [], "direct": []}
    for I in insts:
        for mode in obj:
            s = O.LpOracle(0, order=O.ORDER_EIGEN, x_update=mode, direct_rows=greedy_split(I) if mode == "direct" else None)
            s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"))
            s.solve_init()
            s.solve_iter(0, 20000)
            assert s.last_stop_reason in (1, 2) and s.check_infeasible_l2f() == 0, mode
            obj[mode].append(-s.cal_Obj())
    p, d = np.array(obj["pcg"]), np.array(obj["direct"])
    assert abs(d.mean() - p.mean()) < 0.05 * abs(p.mean())


def test_fix_rebuilds_the_inverse():
    """After an early fix E loses columns: the mirror rebuilds H and W and still matches the dense solve on the reduced problem."""
    I = lp_instances("lp_100_500_seed0.npz")[2]
    o = direct_oracle(I, greedy_split(I))
    ref = NumpyLpBox(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"), x_update="direct")
    ref.solve_init()
    z = -np.ones(I["n"])
    assert o.solve_iter_l2f(0, 30, z, 0) == ref.solve_iter_l2f(0, 30, z, 0)
    x = o.vec("x")
    vec = -np.ones(I["n"])
    vec[np.argsort(x)[:40]] = 0.0
    assert o.solve_iter_l2f(30, 60, vec, 40) == ref.solve_iter_l2f(30, 60, vec, 40)
    assert o.get_n() == I["n"] - 40
    assert np.abs(o.vec("x") - ref.x).max() < 1e-9
