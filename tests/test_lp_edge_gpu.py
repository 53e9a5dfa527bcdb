"""Edge cases of the batched LP path: ragged batches (instances of different n, l, nnz in one launch), an instance with an empty
column and an empty row, a single-variable-per-row instance; each against the oracle in the kernels' order, bit-exact."""
import numpy as np
import pytest

from helpers import bits_equal, lp_instances, oracle_for, oracle_like, scripted_fix_vec
from lpbox_hip.lp import LpBatch

pytestmark = pytest.mark.gpu


def _csc(cols, l):
    colptr, rowidx = [0], []
    for c in cols:
        rowidx += sorted(c)
        colptr.append(len(rowidx))
    return dict(n=len(cols), l=l, nnz=len(rowidx), colptr=np.array(colptr, np.int32), rowidx=np.array(rowidx, np.int32))


def _check(insts, windows=((0, 40), (40, 400))):
    b = LpBatch(insts)
    b.solve_init()
    os_ = [oracle_for(b, i, I) for i, I in enumerate(insts)]
    for (a, e) in windows:
        rets = b.solve_iter(a, e)
        for i, o in enumerate(os_):
            ro = o.solve_iter(a, e)
            assert rets[i] == ro, (i, a, e)
            for name in ("x", "z1", "z2", "z4"):
                assert bits_equal(b.debug_vec(name, i), o.vec(name)), (i, name, a, e)
            assert b.counters(i) == (o.total_outer_iters, o.total_pcg_iters), i
    return b, os_


def test_ragged_batch_of_three_sizes():
    small = lp_instances("lp_20_60_seed0.npz")[:3]
    mid = lp_instances("lp_100_500_seed0.npz")[:2]
    _check([small[0], mid[0], small[1], mid[1], small[2]])


def test_empty_column_and_empty_row():
    rs = np.random.RandomState(7)
    cols = [list(rs.choice(12, size=rs.randint(1, 4), replace=False)) for _ in range(30)]
    cols[5] = []                                     # a bid on nothing: column without entries (pd = rho1 + rho2 only)
    cols = [[r for r in c if r != 9] for c in cols]   # item 9 is in no bid: empty row (y3 = f, contributes nothing)
    I = _csc(cols, 12)
    I["b"] = -rs.uniform(1, 100, size=I["n"])
    _check([I])


def test_identity_like_constraints():
    n = 40
    I = _csc([[j] for j in range(n)], n + 1)         # every variable alone in its row (n == l is refused: reference quirk Q5, LPcpp:103-107)
    I["b"] = -np.linspace(1, 50, n)
    _check([I])


def test_square_instance_is_refused_loudly():
    from lpbox_hip.lp import LpboxError
    I = _csc([[j] for j in range(8)], 8)
    I["b"] = -np.ones(8)
    with pytest.raises(LpboxError, match="n == l"):
        LpBatch([I])


def test_handles_driven_from_different_threads():
    """One handle = one stream; different handles may be driven concurrently from different threads (ctypes releases the GIL): the
    results are those of the sequential runs, bit for bit."""
    import threading
    insts = lp_instances("lp_100_500_seed0.npz")
    groups = [insts[0:16], insts[16:32], insts[32:48], insts[48:64]]

    def solve(g):
        b = LpBatch(g)
        b.solve_init()
        b.solve_iter(0, 3000)
        return [b.debug_vec("x", i) for i in range(b.B)], [b.counters(i) for i in range(b.B)]
    seq = [solve(g) for g in groups]
    out = [None] * len(groups)

    def run(k):
        out[k] = solve(groups[k])
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(groups))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(len(groups)):
        assert out[k][1] == seq[k][1]
        assert all(bits_equal(a, b) for a, b in zip(out[k][0], seq[k][0]))


@pytest.mark.parametrize("n,seed", [(800, 1), (1500, 2)])
def test_two_and_four_slot_variants_bit_exact(n, seed):
    """Instances between the benchmark sizes: n = 800 runs on the 512 x 2 kernel, n = 1500 on the register-lean 512 x 4 one, both with
    long columns shared inside quads of lanes of a slot; iterates, state and an early-fixing window bit for bit against the oracle."""
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like
    P = make_auction_like(n, seed)
    I = dict(n=P["n"], l=P["l"], colptr=P["colptr"], rowidx=P["rowidx"], b=P["b"])
    g = PyLPboxADMMsolver(0)
    g.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    g.solve_init()
    cfg = g.batch.config()
    assert (cfg["threads"], cfg["elems_per_thread"]) == ((512, 2) if n <= 1024 else (512, 4))
    own, help4 = g.batch.col_split(0)
    assert (help4.sum(axis=1) > 0).any(), "no column was split: the helper-list path is not exercised"
    o = oracle_like(g, I)
    vec, num = np.zeros(I["n"]), 0
    for w in range(3):
        assert g.solve_iter_l2f(100 * w, 100 * (w + 1), vec, num) == o.solve_iter_l2f(100 * w, 100 * (w + 1), vec, num)
        xg = g.get_x_iters_2d(100)
        assert bits_equal(xg, o.get_x_iters_2d(100)), w
        vec, num = scripted_fix_vec(xg, lo=0.05, hi=0.95, last=10)
        if num <= 10:
            num = 0
    assert g.cal_Obj() == o.cal_Obj() and g.get_n() == o.get_n()


def test_mixed_size_batch_bit_exact():
    """One batch holding instances of very different sizes (n = 60 next to n = 500): the common layout has many empty positions for the
    small ones (quads of holes, helper lanes without a column); each instance still equals its oracle bit for bit."""
    from lpbox_hip.lp import LpBatch
    small, big = lp_instances("lp_20_60_seed0.npz"), lp_instances("lp_100_500_seed0.npz")
    insts = [small[0], big[3], small[5], big[200], small[2]]
    B = LpBatch(insts)
    B.solve_init()
    rets = B.solve_iter(0, 3000)
    for i, I in enumerate(insts):
        o = oracle_for(B, i, I)
        assert int(rets[i]) == o.solve_iter(0, 3000), i
        assert B.counters(i) == (o.total_outer_iters, o.total_pcg_iters), i
        assert bits_equal(B.debug_vec("x", i), o.vec("x")) and B.cal_obj(i) == o.cal_Obj(), i
