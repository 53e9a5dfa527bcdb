"""Generic constrained binary QP (the reference's ADMM_bqp, SEGcpp:1384-1832): the HIP path through the
C-ABI against the CPU oracle (oracle/bqp_oracle.c) in the kernels' reduction order, bit-exact, for all four problem types."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, bits_equal, lp_instances
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _pair(P, **kw):
    from lpbox_hip.bqp import BqpSolver
    g = BqpSolver(P["n"], P["A"], P["b"], P["x0"], P.get("C"), P.get("d"), P.get("E"), P.get("f"), **kw)
    it_g = g.solve()
    o = O.BqpOracle(P, order=O.ORDER_GPU, T=int(g.scalar("threads")), chunk=int(g.scalar("chunk")),
                    preset=kw.get("preset"), params=kw.get("params"))
    it_o = o.solve()
    return g, o, it_g, it_o


def _compare(g, o, it_g, it_o, names):
    assert it_g == it_o and g.scalar("stop") == o.scalar("stop")
    assert g.scalar("total_pcg") == o.scalar("total_pcg")
    for name in names:
        assert bits_equal(g.vec(name), o.vec(name)), f"{name}: max diff {np.abs(g.vec(name) - o.vec(name)).max():.3e}"
    for name in ("rho1", "gamma", "std_obj", "cvg1", "cvg2", "best_bin_obj", "obj_val"):
        assert np.isfinite(o.scalar(name)) and g.scalar(name) == o.scalar(name), name


def _seg_problem():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(GOLDEN, "seg", "0.jpg")).convert("L"))
    small = O.seg_resize_u8(img, (1e4 / img.size) ** 0.5)
    S = O.seg_build_costs(small.astype(np.float64))
    return dict(n=S["n"], A=(S["rowptr"], S["colidx"], S["vals"]), b=S["b"], x0=np.zeros(S["n"])), S


def test_unconstrained_equals_oracle_and_the_segmentation_legacy_loop():
    P, S = _seg_problem()
    g, o, it_g, it_o = _pair(P)
    _compare(g, o, it_g, it_o, ("x", "y1", "y2", "z1", "z2", "best_sol"))
    # the same arithmetic as ADMM_bqp_unconstrained_legacy (SEGcpp:1200-1380) from x0 = 0: the segmentation flavour's result
    from lpbox_hip.seg import PyLPboxADMMsolver
    s = PyLPboxADMMsolver(0, 10000, 0)
    s.set_problem(S)
    s.solve_init()
    s.solve_iter()
    assert np.array_equal((g.vec("x") >= 0.5).astype(float), np.asarray(s.get_x_sol()).ravel())


def _lp_as_bqp(I):
    """A combinatorial-auction LP instance as a BQP with linear inequalities: A = 0 (explicit zero diagonal), b, E x <= 1."""
    n, l = I["n"], I["l"]
    rows = np.repeat(np.arange(n), np.diff(I["colptr"]))          # entry k sits in column rows[k], row rowidx[k]
    order = np.lexsort((rows, I["rowidx"]))
    Er = np.concatenate([[0], np.cumsum(np.bincount(I["rowidx"], minlength=l))]).astype(np.int32)
    A = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.zeros(n))
    return dict(n=n, A=A, b=I["b"], x0=np.ones(n), E=(Er, rows[order].astype(np.int32), np.ones(len(order))), f=np.ones(l))


def test_inequality_type_on_an_auction_instance():
    P = _lp_as_bqp(lp_instances("lp_20_60_seed0.npz")[0])
    prm = [1e-4, 1e-6, 1.6, 0.95, 5, 1500, 25, 3, 1.01, 1e-4, 1000]          # the ineq preset with a shorter max_iters
    g, o, it_g, it_o = _pair(P, params=prm)
    _compare(g, o, it_g, it_o, ("x", "y1", "y2", "z1", "z2", "z4", "y3", "best_sol"))


def _random_problem(n, groups, l, seed):
    rs = np.random.RandomState(seed)
    # A: sparse symmetric with full diagonal
    dense = np.zeros((n, n))
    for _ in range(3 * n):
        i, j = rs.randint(n), rs.randint(n)
        v = rs.uniform(-1, 1)
        dense[i, j] += v; dense[j, i] += v
    dense[np.arange(n), np.arange(n)] = 0.0
    dense[np.arange(n), np.arange(n)] = np.abs(dense).sum(axis=1) + rs.uniform(0.5, 2.0, n)      # diagonally dominant: A is positive definite

    def csr(M, keep_diag=False):
        rp, ci, va = [0], [], []
        for i in range(M.shape[0]):
            for j in range(M.shape[1]):
                if M[i, j] != 0 or (keep_diag and i == j):
                    ci.append(j); va.append(M[i, j])
            rp.append(len(ci))
        return np.array(rp, np.int32), np.array(ci, np.int32), np.array(va, np.float64)
    Cm = np.zeros((groups, n))
    for j in range(n):
        Cm[j % groups, j] = 1.0                       # every variable in exactly one group, one of each group must be chosen
    Em = (rs.rand(l, n) < 0.15) * rs.uniform(0.5, 1.5, (l, n))
    return dict(n=n, A=csr(dense, True), b=rs.uniform(-2, 1, n), x0=rs.uniform(0.2, 0.8, n), C=csr(Cm), d=np.ones(groups),
                E=csr(Em), f=np.full(l, 2.0))


@pytest.mark.parametrize("kind", ["eq", "both"])
def test_equality_and_mixed_types(kind):
    P = _random_problem(600, 40, 50, 3)
    if kind == "eq":
        P.pop("E"); P.pop("f")
    prm = [1e-4, 1e-6, 1.6, 0.95, 5, 800, 1 if kind == "eq" else 25, 3, 1.05 if kind == "eq" else 1.01, 1e-4, 1000]
    g, o, it_g, it_o = _pair(P, params=prm)
    names = ("x", "y1", "y2", "z1", "z2", "z3", "best_sol") + (("z4", "y3") if kind == "both" else ())
    _compare(g, o, it_g, it_o, names)
    # sanity of the answer: the equality residual of the continuous iterate shrinks well below its start
    Cx = np.zeros(40)
    np.add.at(Cx, np.arange(600) % 40, g.vec("x"))
    assert np.abs(Cx - 1).max() < 0.5


def test_named_entry_points_and_scipy_input():
    import scipy.sparse as sp
    from lpbox_hip import bqp
    rs = np.random.RandomState(5)
    n = 300
    a, b = rs.uniform(0.5, 2.0, n), rs.uniform(-4, 2, n)
    sol = bqp.ADMM_bqp_unconstrained(n, sp.diags(a).tocsr(), b, np.zeros(n))
    clear = np.abs(a + b) > 0.3
    assert np.array_equal((sol["x_sol"] >= 0.5)[clear], (a + b < 0)[clear]) and sol["stop"] in (1, 2)
    # one-of-each-group: a scipy matrix WITHOUT stored diagonal is completed with explicit zeros
    groups, per = 10, 4
    n = groups * per
    cost = rs.uniform(1, 2, n); pick = np.arange(groups) * per + rs.randint(per, size=groups); cost[pick] = 0.1
    Cm = sp.csr_matrix((np.ones(n), (np.arange(n) // per, np.arange(n))), shape=(groups, n))
    sol = bqp.ADMM_bqp_linear_eq(n, sp.csr_matrix((n, n)), cost, np.full(n, 1.0 / per), groups, Cm, np.ones(groups))
    want = np.zeros(n); want[pick] = 1
    assert np.array_equal((sol["x_sol"] >= 0.5).astype(float), want)


def test_best_sol_when_the_loop_ends_on_an_improving_iteration():
    """max_iters chosen so that the last iteration run still improves best_bin_obj: best_sol must be that iteration's x_sol."""
    P = _random_problem(300, 20, 30, 7)
    P.pop("E"); P.pop("f")
    hit = 0
    for mi in (1, 2, 3, 4, 6, 9):
        prm = [1e-4, 1e-6, 1.6, 0.95, 5, mi, 1, 3, 1.05, 1e-4, 1000]
        g, o, it_g, it_o = _pair(P, params=prm)
        assert it_g == it_o and bits_equal(g.vec("best_sol"), o.vec("best_sol")) and bits_equal(g.vec("x"), o.vec("x"))
        hit += bits_equal(o.vec("best_sol"), o.vec("x"))
    assert hit > 0            # at least one of the runs ended on an improving iteration (best_sol == x_sol)
