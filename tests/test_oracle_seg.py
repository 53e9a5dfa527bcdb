"""CPU tests of the segmentation oracle and of the product's host-side cost construction (no GPU needed)."""
import os

import numpy as np

from helpers import GOLDEN
from oracle import oracle as O


def gray(name="0.jpg"):
    from lpbox_hip.seg import load_gray
    return load_gray(os.path.join(GOLDEN, "seg", name))


def test_cost_builder_product_equals_oracle():
    """lpbox_seg_set_image (C++ in the C-ABI library) vs oracle/seg_oracle.c on the reference's own sample image:
    resize, unary / pairwise costs (SEGcpp:46-248), A, b, c -- exact, including the explicit zeros and their signs."""
    from lpbox_hip.seg import PyLPboxADMMsolver
    g = gray()
    for nodes in (10000, 2500):
        s = PyLPboxADMMsolver(0, nodes, 0)
        s.set_image(g)
        P = s.get_problem()
        small = O.seg_resize_u8(g, np.sqrt(nodes / g.size))
        Q = O.seg_build_costs(small.astype(float))
        assert (P["n"], P["rows"], P["cols"], P["c"]) == (Q["n"], Q["rows"], Q["cols"], Q["c"])
        for k in ("rowptr", "colidx", "vals", "b"):
            assert np.array_equal(P[k], Q[k]), k
        assert np.array_equal(np.signbit(P["vals"]), np.signbit(Q["vals"]))


def test_problem_structure_matches_survey_q3():
    """7-diagonal matrix with offsets {0, +-1, +-(ncols-1), +-ncols}, symmetric, zero row sums, integer data."""
    g = gray()
    small = O.seg_resize_u8(g, np.sqrt(10000 / g.size))
    P = O.seg_build_costs(small.astype(float))
    n, rows, cols = P["n"], P["rows"], P["cols"]
    assert n == rows * cols == 10005
    import scipy.sparse as sp
    A = sp.csr_matrix((P["vals"], P["colidx"], P["rowptr"]), shape=(n, n))
    offs = set(np.unique(P["colidx"] - np.repeat(np.arange(n), np.diff(P["rowptr"]))))
    assert offs <= {0, 1, -1, cols - 1, -(cols - 1), cols, -cols}
    assert abs(A - A.T).max() == 0
    assert np.abs(np.asarray(A.sum(axis=1))).max() == 0          # A = D - W
    assert np.all(P["vals"] == np.round(P["vals"])) and np.all(P["b"] == np.round(P["b"]))
    assert np.all(np.diff(P["rowptr"]) <= 7) and np.all(A.diagonal() >= 0)


def test_seg_oracle_regression_and_energy():
    g = gray()
    small = O.seg_resize_u8(g, np.sqrt(10000 / g.size))
    P = O.seg_build_costs(small.astype(float))
    s = O.SegOracle(0, 10000, 0)
    s.set_problem(P)
    s.solve_init()
    e = s.solve_iter()
    # numbers of this oracle (Eigen order) on 0.jpg @ 1e4 nodes; they pin the oracle against accidental change
    assert (e, s.legacy_iter_plus1, s.last_stop) == (9575, 457, 1)
    x = s.get_x_sol().ravel()
    import scipy.sparse as sp
    A = sp.csr_matrix((P["vals"], P["colidx"], P["rowptr"]), shape=(P["n"], P["n"]))
    assert s.get_obj() == x @ (A @ x) + P["b"] @ x + P["c"] == 9575.0
    # the ADMM labelling beats both trivial labellings
    assert s.get_obj() < min(P["c"], P["c"] + P["b"].sum())


def test_seg_l2f_equals_legacy_without_fixing_and_gpu_order_close():
    g = gray("7.jpg")
    small = O.seg_resize_u8(g, np.sqrt(2500 / g.size))
    P = O.seg_build_costs(small.astype(float))
    a = O.SegOracle(0, 2500, 0); a.set_problem(P); a.solve_init()
    b = O.SegOracle(0, 2500, 0); b.set_problem(P); b.solve_init()
    c = O.SegOracle(0, 2500, 0, order=O.ORDER_GPU); c.set_problem(P); c.solve_init()
    a.solve_iter()
    z = np.zeros(P["n"])
    for w in range(200):
        r = b.solve_iter_l2f(10 * w, 10 * w + 10, z, 0)
        if w == 0:
            c.solve_iter_l2f(0, 10, z, 0)
            assert np.abs(b.get_x_iters_2d(10) - c.get_x_iters_2d(10)).max() < 1e-6
        if r:
            break
    assert np.array_equal(a.get_x_sol(), b.get_x_sol()) and a.get_obj() == b.get_obj()
    assert a.total_outer_iters == b.total_outer_iters
