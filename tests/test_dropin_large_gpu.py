"""The drop-in class on instances that do not fit the on-chip kernel (max(n, l) > 2048): PyLPboxADMMsolver hands them to the
large-instance path at solve_init and keeps the whole pyx surface -- plain solve, early-fixing windows, x_iters, solutions, objective,
infeasibility counts, the instance-file route -- bit-exact against the oracle in that path's (two-level) summation order.  The
reference has no size limit (LPcpp:2446-2545 reads whatever the files hold)."""
import os

import numpy as np
import pytest

from helpers import bits_equal, scripted_fix_vec
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle(P, g):
    cfg = g.batch.config()
    o = O.LpOracle(0, order=O.ORDER_GPU, T=cfg["threads"], chunk=cfg["chunk"])
    o.set_problem(P["n"], P["l"], P["colptr"], P["rowidx"], P["b"])
    o.solve_init()
    return o


def test_oversize_instance_runs_through_the_dropin_class():
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like
    P = make_auction_like(2600, 4)
    assert max(P["n"], P["l"]) > 2048
    g = PyLPboxADMMsolver(0)
    g.set_problem(P["n"], P["l"], P["colptr"], P["rowidx"], P["b"])
    assert not g.large
    assert g.solve_init() == 1
    assert g.large and g.get_n() == P["n"]
    o = _oracle(P, g)
    vec, num = np.zeros(P["n"]), 0
    fixed = 0
    for w in range(6):
        rg, ro = g.solve_iter_l2f(100 * w, 100 * (w + 1), vec, num), o.solve_iter_l2f(100 * w, 100 * (w + 1), vec, num)
        assert rg == ro
        assert (g.get_n(), g.get_iter()) == (o.get_n(), o.get_iter())
        xg, xo = g.get_x_iters_2d(100), o.get_x_iters_2d(100)
        assert bits_equal(xg, xo), f"window {w}"
        assert bits_equal(g.get_x_iters_1d(100).ravel(), xo.ravel()[: xo.shape[0] * 20])      # LP pyx:35-41: the first n*20 entries
        assert g.cal_Obj() == o.cal_Obj() and g.get_curBinObj() == o.get_curBinObj()
        assert np.array_equal(g.get_x_sol().ravel(), o.get_x_sol().ravel())
        assert bits_equal(g.get_final_x_sol().ravel(), o.get_final_x_sol().ravel())
        assert g.check_infeasible_l2f() == o.check_infeasible_l2f()
        assert g.check_infeasible_lpbox() == o.check_infeasible_lpbox()
        if rg:
            break
        vec, num = scripted_fix_vec(xg, lo=0.05, hi=0.95, last=20)
        fixed += num
    assert fixed > 0


def test_oversize_plain_solve_and_file_route(tmp_path):
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like, write_instance_files
    P = make_auction_like(2300, 7)
    d = tmp_path / "instance" / "1000_2300"
    os.makedirs(d)
    l = write_instance_files(P, str(d / "instance_1_C.txt"), str(d / "instance_1_b.txt"))
    assert l == P["l"]
    g = PyLPboxADMMsolver(0)
    g.data_root = str(tmp_path)
    g.write_files = False
    g.read_File(1, 1000, 2300)
    g.solve_init()
    assert g.large
    o = _oracle(P, g)
    for (a, b) in ((0, 60), (60, 20000)):
        assert g.solve_iter(a, b) == o.solve_iter(a, b)
        assert g.batch.counters() == (o.total_outer_iters, o.total_pcg_iters)
    assert g.cal_Obj() == o.cal_Obj()
    assert np.array_equal(g.get_x_sol().ravel(), o.get_x_sol().ravel())
    assert g.check_infeasible_l2f() == o.check_infeasible_l2f()
    # the same object takes a small instance afterwards and goes back to the on-chip kernel
    S = make_auction_like(300, 1)
    g.set_problem(S["n"], S["l"], S["colptr"], S["rowidx"], S["b"])
    g.solve_init()
    assert not g.large and g.get_n() == 300


@pytest.mark.parametrize("print_info", [2, 3])
def test_oversize_instance_writes_the_iterate_dump(print_info, tmp_path):
    """print_info 2 (every iterate, LPcpp:777-780, :903-909) and 3 (the iterate of the stop, :940-946) behind the size hand-over: the file
    route of the drop-in class writes <root>/xiter/<k>_<j>_xiters_<i>.csv for an instance the on-chip kernel does not hold, and every
    printed row is the oracle's iterate to the printed digits."""
    from lpbox_hip import files
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like, write_instance_files
    P = make_auction_like(2300, 7)
    d = tmp_path / "instance" / "1000_2300"
    os.makedirs(d)
    write_instance_files(P, str(d / "instance_1_C.txt"), str(d / "instance_1_b.txt"))
    g = PyLPboxADMMsolver(print_info)
    g.data_root = str(tmp_path)
    g.write_files = True
    g.read_File(1, 1000, 2300)
    g.solve_init()
    assert g.large
    o = _oracle(P, g)
    iters = 40 if print_info == 2 else 20000
    assert g.solve_iter(0, iters) == o.solve_iter(0, iters)
    path = tmp_path / "xiter" / "1000_2300_xiters_1.csv"
    if print_info == 2:
        X = files.read_xiters_csv(str(path))                              # (n, iterations), as LP/trainer.py:32-48 reads it
        assert X.shape == (P["n"], 40)
        for k in (0, 1, 39):                  # (a fresh oracle per column: a resumed plain call overwrites z4 on its first iteration, LPcpp:920-923)
            o2 = _oracle(P, g)
            o2.solve_iter(0, k + 1)
            assert np.array_equal(X[:, k], np.array([float("%f" % v) for v in o2.vec("x")])), k
    else:
        reason, p1 = g.batch.stop(0)
        assert reason in (1, 2)
        lines = open(path).read().splitlines()
        assert len(lines) == 1 and lines[0] == "Iter%d," % p1 + ",".join("%f" % v for v in o.vec("x"))
    rec = files.read_results_csv(str(tmp_path / "xiter" / "allres.csv"))
    assert len(rec) == 1 and rec[0][0] == 1 and rec[0][2] == g.batch.stop(0)[1]
