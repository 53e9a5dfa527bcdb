"""CPU checks of oracle/bqp_oracle.c (the reference's generic ADMM_bqp, SEGcpp:1384-1832).  The reference holds no fixture for this
function (nothing calls it), so the oracle is pinned by cross-checks: its unconstrained type against the segmentation oracle's
legacy loop (same arithmetic, separately restated from SEGcpp:1200-1380), analytic answers of separable problems, and the
agreement of its two summation orders."""
import os

import numpy as np

from helpers import GOLDEN, bits_equal
from oracle import oracle as O


def _diag_csr(vals):
    n = len(vals)
    return np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.asarray(vals, np.float64)


def test_unconstrained_type_is_the_segmentation_legacy_loop():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(GOLDEN, "seg", "0.jpg")).convert("L"))
    small = O.seg_resize_u8(img, (1e4 / img.size) ** 0.5)
    S = O.seg_build_costs(small.astype(np.float64))
    s = O.SegOracle(0, 10000, 0)
    s.set_problem(S)
    s.solve_init()
    s.solve_iter()
    b = O.BqpOracle(dict(n=S["n"], A=(S["rowptr"], S["colidx"], S["vals"]), b=S["b"], x0=np.zeros(S["n"])))
    it = b.solve()
    assert it + 1 == s.legacy_iter_plus1 and b.scalar("total_pcg") == s.total_pcg_iters and b.scalar("stop") == s.last_stop
    assert bits_equal(b.vec("x"), s.vec("x")) and bits_equal(b.vec("z1"), s.vec("z1"))
    assert np.array_equal(b.pcg_trace(), s.pcg_trace())


def test_separable_problem_has_the_analytic_answer():
    # min sum a_i x_i^2 + b_i x_i over {0,1}: x_i = 1 iff a_i + b_i < 0
    rs = np.random.RandomState(0)
    n = 200
    a, b = rs.uniform(0.5, 2.0, n), rs.uniform(-4, 2, n)
    o = O.BqpOracle(dict(n=n, A=_diag_csr(a), b=b, x0=np.zeros(n)))
    o.solve()
    clear = np.abs(a + b) > 0.3
    assert np.array_equal((o.vec("x") >= 0.5)[clear], (a + b < 0)[clear])
    assert o.scalar("stop") in (1.0, 2.0)


def test_one_of_each_group_equality_constraints():
    # linear costs, every variable in exactly one group, sum over a group = 1: the optimum takes the cheapest member of each group
    rs = np.random.RandomState(1)
    groups, per = 12, 5
    n = groups * per
    cost = rs.uniform(1, 2, n)
    best = np.zeros(n)
    for g in range(groups):
        k = g * per + rs.randint(per)
        cost[k] = 0.1                                   # clearly cheapest
        best[k] = 1
    Cp = np.arange(0, n + 1, per, dtype=np.int32)
    P = dict(n=n, A=_diag_csr(np.zeros(n)), b=cost, x0=np.full(n, 1.0 / per), C=(Cp, np.arange(n, dtype=np.int32), np.ones(n)), d=np.ones(groups))
    o = O.BqpOracle(P)
    o.solve()
    assert np.array_equal((o.vec("x") >= 0.5).astype(float), best)
    Cx = (o.vec("x") >= 0.5).reshape(groups, per).sum(axis=1)
    assert np.array_equal(Cx, np.ones(groups))


def test_inequality_constraints_are_respected_and_orders_agree_early():
    # knapsack-like rows: at most one of each overlapping pair; negative costs pull everything to 1
    rs = np.random.RandomState(2)
    n, l = 80, 60
    rows = [sorted(rs.choice(n, 2, replace=False)) for _ in range(l)]
    Ep = np.arange(0, 2 * l + 1, 2, dtype=np.int32)
    Ei = np.array([c for r in rows for c in r], np.int32)
    P = dict(n=n, A=_diag_csr(np.zeros(n)), b=-rs.uniform(1, 2, n), x0=np.ones(n), E=(Ep, Ei, np.ones(2 * l)), f=np.ones(l))
    prm = [1e-4, 1e-6, 1.6, 0.95, 5, 4000, 25, 3, 1.01, 1e-4, 1000]
    a = O.BqpOracle(P, params=prm)
    a.solve()
    xb = (a.vec("x") >= 0.5).astype(float)
    assert all(xb[r].sum() <= 1 for r in rows) and xb.sum() > 0
    prm[5] = 3                                            # three iterations in both summation orders: agreement to rounding
    e = O.BqpOracle(P, params=prm); e.solve()
    g = O.BqpOracle(P, params=prm, order=O.ORDER_GPU, T=256, chunk=512); g.solve()
    assert np.abs(e.vec("x") - g.vec("x")).max() < 1e-9
