"""Rule-based early fixing, ADMM_lp_iters_fix (LPcpp:1689-2286; in the C++ class, not in the pxd): the repaired semantics of
lpbox_hip.lp.PyLPboxADMMsolver.solve_iter_fix against a restatement in the reference's own statement order on the oracle
(helpers.oracle_iters_fix: fix at the END of the iteration that decided it; the product applies it at the start of the next
single-iteration window -- same state evolution, different decomposition, so the zero-length fix window is exercised too)."""
import numpy as np
import pytest

from helpers import bits_equal, lp_instances, oracle_iters_fix, oracle_like

pytestmark = pytest.mark.gpu


def _solver(I):
    from lpbox_hip.lp import PyLPboxADMMsolver
    g = PyLPboxADMMsolver(0)
    g.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    g.solve_init()
    return g


@pytest.mark.parametrize("fixture,idx,calls,min_fix", [("lp_100_500_seed0.npz", 0, [(0, 60), (60, 500)], 3),
                                                       ("lp_100_500_seed0.npz", 1, [(0, 150), (150, 420)], 10),
                                                       ("lp_100_500_seed0.npz", 7, [(0, 2500)], 10)])
def test_rule_based_fixing_matches_reference_ordered_restatement(fixture, idx, calls, min_fix):
    I = lp_instances(fixture)[idx]
    g = _solver(I)
    o = oracle_like(g, I)
    prev = None
    for (a, b) in calls:
        rg = g.solve_iter_fix(a, b, min_fix=min_fix)
        ro, prev = oracle_iters_fix(o, a, b, prev, min_fix=min_fix)
        assert rg == ro and g.get_n() == o.get_n()
        assert bits_equal(g.get_final_x_sol().ravel(), o.get_final_x_sol().ravel())
        assert g.cal_Obj() == o.cal_Obj()
        if rg:
            break
    assert g.get_n() < I["n"], "the rule never fixed anything: the test does not exercise the fix path"
    assert np.array_equal(g.get_x_sol().ravel(), o.get_x_sol().ravel())
    assert g.check_infeasible_l2f() == o.check_infeasible_l2f()


def test_rule_with_unreachable_consistency_is_the_l2f_loop():
    """No variable can be flagged -> nothing is fixed -> the iterates are those of plain single-iteration l2f windows."""
    I = lp_instances("lp_20_60_seed0.npz")[1]
    a, b = _solver(I), _solver(I)
    ra = a.solve_iter_fix(0, 120, consistency=10 ** 9)
    rb = 0
    z = np.zeros(I["n"])
    for it in range(120):
        rb = b.solve_iter_l2f(it, it + 1, z, 0)
        if rb:
            break
    assert a.get_n() == b.get_n() == I["n"]
    assert bits_equal(a.get_final_x_sol(), b.get_final_x_sol()) and a.cal_Obj() == b.cal_Obj()


def test_reinit_resets_the_persistence_memory():
    """ADMM_lp_iters_init sets x_prev = Zero(n) (LPcpp:572): a second solve on the same object must take the decisions of a fresh object,
    not compare its first iterates with the last iterate of the solve before."""
    I = lp_instances("lp_100_500_seed0.npz")[1]
    a = _solver(I)
    a.solve_iter_fix(0, 300, min_fix=10)
    assert a.get_n() < I["n"]
    a.solve_init()                                                       # same size: the stale vector would have fitted
    b = _solver(I)
    ra, rb = a.solve_iter_fix(0, 300, min_fix=10), b.solve_iter_fix(0, 300, min_fix=10)
    assert ra == rb and a.get_n() == b.get_n()
    assert bits_equal(a.get_final_x_sol(), b.get_final_x_sol()) and a.cal_Obj() == b.cal_Obj()
