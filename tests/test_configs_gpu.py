"""BASELINE configs 4 and 5 at their real sizes (-m gpu).

Config 4 (one GPU's share of it): the 256 DISTINCT j=500/k=2000 instances of the reference generator
(tests/golden/lp_500_2000_seed0.npz) solved to convergence in one launch -- size-independent properties for all of them,
oracle parity to convergence for three of them.
Config 5 (world = 1): the n = 10^6 instance -- determinism, objective and feasibility recomputed on the host, and bit-exact
iterates against the oracle at the largest size the oracle manages in under a minute.
"""
import os

import numpy as np
import pytest

from helpers import bits_equal, lp_instances, oracle_for, oracle_full_solve

pytestmark = pytest.mark.gpu


def test_config4_full_batch_256_distinct_instances():
    from lpbox_hip.lp import LpBatch
    insts = lp_instances("lp_500_2000_seed0.npz")
    assert len(insts) == 256 and all(I["n"] == 2000 for I in insts)
    assert len({(int(I["l"]), len(I["rowidx"]), float(I["b"][:8].sum())) for I in insts}) == 256       # distinct problems
    B = LpBatch(insts)
    assert B.config()["threads"] * B.config()["elems_per_thread"] >= 2000

    def solve(batch):
        batch.solve_init()
        rets = batch.solve_iter(0, 20000)
        return [(batch.get_x_sol(i).copy(), batch.cal_obj(i), batch.counters(i), batch.debug_vec("x", i), int(rets[i])) for i in range(batch.B)]
    r1 = solve(B)
    r2 = solve(B)                                                    # (1) determinism
    for i in range(256):
        assert r1[i][1] == r2[i][1] and r1[i][2] == r2[i][2] and bits_equal(r1[i][3], r2[i][3]), i
    sub = [200, 3, 77, 255, 128, 9]                                   # (2) slot / neighbour independence on a re-ordered subset
    rs = solve(LpBatch([insts[k] for k in sub]))
    for slot, k in enumerate(sub):
        assert rs[slot][1] == r1[k][1] and rs[slot][2] == r1[k][2] and bits_equal(rs[slot][3], r1[k][3]), (slot, k)
    for i, I in enumerate(insts):                                    # (3) reported objective / feasibility recomputed on the host
        x = r1[i][0].ravel()
        assert set(np.unique(x)) <= {0.0, 1.0}
        assert abs(float(I["b"] @ x) - r1[i][1]) <= 1e-9 * max(1.0, abs(r1[i][1]))
        rows = np.zeros(I["l"])
        np.add.at(rows, I["rowidx"], np.repeat(x, np.diff(I["colptr"])))
        assert int((rows > 1.0).sum()) == B.check_infeasible_l2f(i)
        assert 0 < r1[i][2][0] <= 20000
    assert max(r[2][0] for r in r1) < 20000                          # every instance met a reference stop test
    for i in (0, 131, 255):                                          # (4) oracle parity TO CONVERGENCE
        o = oracle_for(B, i, insts[i])
        ro = o.solve_iter(0, 20000)
        assert r1[i][4] == ro and r1[i][2] == (o.total_outer_iters, o.total_pcg_iters)
        assert bits_equal(r1[i][3], o.vec("x")) and r1[i][1] == o.cal_Obj()
        assert np.array_equal(r1[i][0].ravel(), o.get_x_sol().ravel())


def _big_oracle(P, big):
    from oracle import oracle as O
    o = O.LpOracle(0, order=O.ORDER_GPU, T=int(big.scalar("threads")), chunk=int(big.scalar("chunk")))
    o.set_problem(P["n"], P["l"], P["colptr"], P["rowidx"], P["b"])
    o.solve_init()
    return o


def test_config5_million_variables_single_rank():
    from lpbox_hip.big import BigLp
    from lpbox_hip.synth import make_auction_like
    P = make_auction_like(1000000, 0)
    runs = []
    for _ in range(2):
        g = BigLp(P)
        g.solve_init()
        g.solve_iter(0, 60)
        runs.append((g.local_x(), g.vec("z4"), g.scalar("cur_obj"), g.scalar("pcg_total"), g.cal_Obj(), g.local_x_sol()))
        g.close()
    a, b = runs
    assert bits_equal(a[0], b[0]) and bits_equal(a[1], b[1]) and a[2:5] == b[2:5]                     # determinism at full size
    x = a[0]
    assert np.all(np.isfinite(x)) and x.min() > -0.5 and x.max() < 1.5
    xb = (x >= 0.5).astype(np.float64)                               # binarised iterate: objective and rows recomputed on the host
    assert np.array_equal(xb, a[5])
    assert abs(float(P["b"] @ xb) - a[2]) <= 1e-9 * abs(a[2])        # cur_obj = b . round(x) of the last iteration (LPcpp:1001-1003)
    assert a[4] == a[2]                                              # cal_obj = sum_fix_obj (0) + cur_obj (LPcpp:1630-1642)


@pytest.mark.parametrize("n,seed", [(200000, 2), (1000000, 0)])
def test_config5_windows_bit_exact_at_full_size(n, seed):
    """Config 5's own size (n = 10^6: the oracle needs about a second per iteration) and a 200k instance: every iterate of the
    first windows bit for bit."""
    from lpbox_hip.big import BigLp
    from lpbox_hip.synth import make_auction_like
    P = make_auction_like(n, seed)
    g = BigLp(P)
    g.solve_init()
    o = _big_oracle(P, g)
    for (s, e) in ((0, 3), (3, 8)):
        assert g.solve_iter(s, e) == o.solve_iter(s, e)
        for name in ("x", "z1", "z2", "z4"):
            assert bits_equal(g.vec(name), o.vec(name)), (s, e, name)
        assert (g.scalar("outer_total"), g.scalar("pcg_total")) == (o.total_outer_iters, o.total_pcg_iters)
        assert g.scalar("cur_obj") == o.scalar("cur_obj")


@pytest.mark.parametrize("fixture,count", [("lp_100_500_seed0.npz", 256), ("lp_500_2000_seed0.npz", 48)])
def test_every_benchmark_instance_bit_exact_to_convergence(fixture, count):
    """Not three spot checks: EVERY instance of the headline batch (and the first 48 of the config-4 batch) solved to convergence by the
    HIP kernel equals the oracle bit for bit -- return code, outer and PCG iteration counts, final iterate, objective, binary solution.
    The oracle solves run on the host cores in parallel (about 1.5 million oracle iterations for the headline batch)."""
    from concurrent.futures import ProcessPoolExecutor
    from lpbox_hip.lp import LpBatch
    insts = lp_instances(fixture)[:count]
    B = LpBatch(insts)
    B.solve_init()
    rets = B.solve_iter(0, 20000)
    cfg = B.config()
    jobs = [(I, cfg["threads"], cfg["threads"] * cfg["elems_per_thread"], B.layout(i), B.row_split(i), B.col_split(i)) for i, I in enumerate(insts)]
    # spawn, not fork: this process has a live HIP runtime, and a forked copy of one is a hazard
    import multiprocessing
    with ProcessPoolExecutor(min(16, os.cpu_count() or 1), mp_context=multiprocessing.get_context("spawn")) as ex:
        res = list(ex.map(oracle_full_solve, jobs, chunksize=2))
    for i, (ret, outer, pcg, obj, x, xs) in enumerate(res):
        assert int(rets[i]) == ret and B.counters(i) == (outer, pcg), i
        assert bits_equal(B.debug_vec("x", i), x) and B.cal_obj(i) == obj, i
        assert np.array_equal(B.get_x_sol(i).ravel(), xs), i
