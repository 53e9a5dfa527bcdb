"""GPU parity tests of the SEGMENTATION flavour: HIP kernels (through the C-ABI) vs the CPU oracle in the kernels' reduction
order.  Bar: bit-exact iterates / state / counters / energy."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, bits_equal, scripted_fix_vec
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def make_pair(num_nodes, image="0.jpg"):
    from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
    gray = load_gray(os.path.join(GOLDEN, "seg", image))
    g = PyLPboxADMMsolver(0, num_nodes, 0)
    g.set_image(gray)
    P = g.get_problem()
    g.solve_init()
    cfg = g.config()
    o = O.SegOracle(0, num_nodes, 0, order=O.ORDER_GPU, T=cfg["threads"], chunk=cfg["threads"] * cfg["elems_per_thread"])
    o.set_problem(P)
    o.solve_init()
    return g, o


def compare_state(g, o, tag):
    left = o.vec("left_idx").astype(int)
    for name in ("x", "z1", "z2", "b"):
        assert bits_equal(g.debug_vec(name)[left], o.vec(name)), f"{tag}: {name}"
    for name in ("rho1", "gamma", "cur_obj", "std_obj", "cvg1", "cvg2", "obj_val", "best_bin_obj"):
        assert g.debug_scalar(name) == o.scalar(name), f"{tag}: {name}"


def test_l2f_windows_bit_exact():
    g, o = make_pair(10000)
    vec = np.zeros(g.get_org_n())
    for w in range(4):
        rg = g.solve_iter_l2f(w * 10, (w + 1) * 10, vec, 0)
        ro = o.solve_iter_l2f(w * 10, (w + 1) * 10, vec, 0)
        xg, xo = g.get_x_iters_2d(10), o.get_x_iters_2d(10)
        if not bits_equal(xg, xo):
            bad = np.where((xg != xo).any(axis=0))[0]
            raise AssertionError(f"window {w}: first differing iteration {bad[0]}, max diff {np.abs(xg - xo)[:, bad[0]].max():.3e}")
        assert rg == ro
        assert g.counters() == (o.total_outer_iters, o.total_pcg_iters)
        compare_state(g, o, f"window {w}")


def test_legacy_full_solve_bit_exact():
    g, o = make_pair(10000)
    eg, eo = g.solve_iter(), o.solve_iter()
    assert eg == eo
    assert g.stop() == (o.last_stop, o.legacy_iter_plus1)
    assert g.counters() == (o.total_outer_iters, o.total_pcg_iters)
    assert np.array_equal(g.get_x_sol(), o.get_x_sol())
    assert g.get_obj() == o.get_obj()
    compare_state(g, o, "final")


def test_early_fix_windows_bit_exact():
    """SEG/trainer.py:699-745 with a scripted policy: windows of 10 iterations, fix what has settled."""
    g, o = make_pair(10000, "7.jpg")
    vec, num = np.zeros(g.get_org_n()), 0
    fixed_any = False
    for w in range(40):
        rg = g.solve_iter_l2f(w * 10, (w + 1) * 10, vec, num)
        ro = o.solve_iter_l2f(w * 10, (w + 1) * 10, vec, num)
        assert rg == ro and g.get_n() == o.get_n(), f"window {w}"
        if rg:
            break
        xg, xo = g.get_x_iters_2d(10), o.get_x_iters_2d(10)
        assert bits_equal(xg, xo), f"window {w}"
        compare_state(g, o, f"window {w}")
        vec, num = scripted_fix_vec(xo, last=5)
        if num <= 10:
            num = 0
        fixed_any |= num > 0
    assert fixed_any
    assert np.array_equal(g.get_x_sol(), o.get_x_sol())
    assert g.get_obj() == o.get_obj()


def test_full_resolution_image_windows_bit_exact():
    """BASELINE config 3 size: 500 x 375 = 187 500 variables."""
    from lpbox_hip.seg import load_gray
    n = load_gray(os.path.join(GOLDEN, "seg", "0.jpg")).size
    g, o = make_pair(n)
    assert g.get_org_n() == 187500
    vec = np.zeros(n)
    for w in range(2):
        assert g.solve_iter_l2f(w * 10, (w + 1) * 10, vec, 0) == o.solve_iter_l2f(w * 10, (w + 1) * 10, vec, 0)
        assert bits_equal(g.get_x_iters_2d(10), o.get_x_iters_2d(10)), f"window {w}"
    compare_state(g, o, "full-res")


@pytest.mark.parametrize("image", ["0.jpg", "7.jpg"])
def test_full_resolution_legacy_solve_bit_exact_to_convergence(image):
    """BASELINE config 3 as bench.py runs it: ADMM_bqp_unconstrained_legacy (SEGcpp:1200-1380) on the full-resolution problem of both sample
    images (187 500 variables for 0.jpg), all the way to its stop test -- energy, stop reason, outer and PCG iteration counts, final iterate
    and duals, binary solution."""
    from lpbox_hip.seg import load_gray
    n = load_gray(os.path.join(GOLDEN, "seg", image)).size
    g, o = make_pair(n, image)
    eg, eo = g.solve_iter(), o.solve_iter()
    assert eg == eo
    assert g.stop() == (o.last_stop, o.legacy_iter_plus1)
    assert g.counters() == (o.total_outer_iters, o.total_pcg_iters) and o.total_outer_iters > 300
    assert np.array_equal(g.get_x_sol(), o.get_x_sol())
    assert g.get_obj() == o.get_obj()
    compare_state(g, o, "full-res final")


def test_batched_legacy_solves_equal_individual_solves():
    """lpbox_seg_legacy_batch: several problems of DIFFERENT sizes advanced in lockstep by one launch chain; each must come out
    bit-identical to its own solve_init() + solve_iter() (own control state, own PCG / outer iteration counts, own stop)."""
    from lpbox_hip.seg import PyLPboxADMMsolver, load_gray, solve_batch

    def gray(name):
        return load_gray(os.path.join(GOLDEN, "seg", name))
    imgs = [(gray("0.jpg"), 10000), (gray("7.jpg"), 2500), (gray("0.jpg")[40:300, 60:420], 6000), (gray("7.jpg"), 10000),
            (gray("0.jpg")[:, ::-1].copy(), 4000)]

    def make(k):
        g, nodes = imgs[k]
        s = PyLPboxADMMsolver(0, nodes, k)
        s.write_files = False
        s.set_image(g)
        return s
    single = [make(k) for k in range(len(imgs))]
    ref = []
    for s in single:
        s.solve_init()
        ref.append((s.solve_iter(), s.get_obj(), s.counters(), s.stop(), s.get_x_sol().copy(), s.debug_vec("x")))
    batch = [make(k) for k in range(len(imgs))]
    en = solve_batch(batch)
    assert len({r[2] for r in ref}) > 1                       # the problems really take different iteration counts
    for k, s in enumerate(batch):
        assert en[k] == ref[k][0] and s.get_obj() == ref[k][1] and s.counters() == ref[k][2] and s.stop() == ref[k][3], k
        assert np.array_equal(s.get_x_sol(), ref[k][4]) and bits_equal(s.debug_vec("x"), ref[k][5]), k
    # a solver of the batch goes on working on its own afterwards
    batch[1].solve_init()
    assert batch[1].solve_iter() == ref[1][0]


def test_diagonal_and_ell_storage_give_the_same_bits(monkeypatch):
    """Image problems are held as 7 diagonals with u8 weights (SegDev::dia); LPBOX_SEG_NODIA keeps the ELL form with f64 values and i32
    columns.  Same rows, same ascending-column order: a full legacy solve and an early-fixing run must agree bit for bit -- and a generic
    matrix that is not of that shape (a value that is no small negative integer) must stay in ELL and still match the oracle."""
    def run(image, nodes):
        from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
        g = PyLPboxADMMsolver(0, nodes, 0)
        g.write_files = False
        g.set_image(load_gray(os.path.join(GOLDEN, "seg", image)))
        g.solve_init()
        vec, num, xs = np.zeros(g.get_org_n()), 0, []
        for w in range(12):
            if g.solve_iter_l2f(w * 10, (w + 1) * 10, vec, num):
                break
            xs.append(g.get_x_iters_2d(10).copy())
            vec, num = scripted_fix_vec(xs[-1], last=5)
            num = num if num > 10 else 0
        l2f = (xs, g.get_obj(), g.counters(), g.get_x_sol().copy())
        g.solve_init()
        return g.debug_scalar("matrix_as_diagonals"), l2f, (g.solve_iter(), g.get_obj(), g.counters(), g.debug_vec("x"))
    da, l2f_a, leg_a = run("7.jpg", 10000)
    monkeypatch.setenv("LPBOX_SEG_NODIA", "1")
    db, l2f_b, leg_b = run("7.jpg", 10000)
    monkeypatch.delenv("LPBOX_SEG_NODIA")
    assert (da, db) == (1.0, 0.0)
    assert len(l2f_a[0]) == len(l2f_b[0]) and all(bits_equal(p, q) for p, q in zip(l2f_a[0], l2f_b[0]))
    assert l2f_a[1:3] == l2f_b[1:3] and np.array_equal(l2f_a[3], l2f_b[3])
    assert leg_a[:3] == leg_b[:3] and bits_equal(leg_a[3], leg_b[3])
    # not diagonal-shaped: one off-diagonal value made fractional (symmetrically)
    from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
    g0 = PyLPboxADMMsolver(0, 2500, 0)
    g0.set_image(load_gray(os.path.join(GOLDEN, "seg", "7.jpg")))
    P = g0.get_problem()
    vals = P["vals"].copy()
    r = 100
    k = [q for q in range(P["rowptr"][r], P["rowptr"][r + 1]) if P["colidx"][q] == r + 1][0]
    k2 = [q for q in range(P["rowptr"][r + 1], P["rowptr"][r + 2]) if P["colidx"][q] == r][0]
    vals[k] = vals[k2] = -0.5
    P2 = dict(P, vals=vals)
    g = PyLPboxADMMsolver(0, 2500, 0)
    g.write_files = False
    g.set_problem(P2)
    g.solve_init()
    assert g.debug_scalar("matrix_as_diagonals") == 0.0
    cfg = g.config()
    o = O.SegOracle(0, 2500, 0, order=O.ORDER_GPU, T=cfg["threads"], chunk=cfg["threads"] * cfg["elems_per_thread"])
    o.set_problem(P2)
    o.solve_init()
    assert g.solve_iter() == o.solve_iter() and g.counters() == (o.total_outer_iters, o.total_pcg_iters)
    compare_state(g, o, "ELL, generic matrix")


def test_batch_refuses_a_handle_listed_twice_and_leaves_the_solvers_usable():
    from lpbox_hip.lp import LpboxError
    from lpbox_hip.seg import PyLPboxADMMsolver, load_gray, solve_batch
    gray = load_gray(os.path.join(GOLDEN, "seg", "7.jpg"))
    a, b = PyLPboxADMMsolver(0, 2500, 0), PyLPboxADMMsolver(0, 2500, 1)
    for s in (a, b):
        s.write_files = False
        s.set_image(gray)
    with pytest.raises(LpboxError, match="same handle"):
        solve_batch([a, b, a])
    with pytest.raises(LpboxError, match="solve_init has not been called"):     # nobody was flagged initialised by the refused call
        a.solve_iter_l2f(0, 10, np.zeros(a.get_org_n()), 0)
    ea, eb = solve_batch([a, b])
    assert ea == eb
    a.solve_init()
    assert a.solve_iter() == ea


def test_alternating_window_lengths_leave_no_stale_columns():
    """x_iters = Zero(n, 10) on every l2f call (SEGcpp:924): a 3-iteration window after a 10-iteration one shows zeros in columns 3..9."""
    g, o = make_pair(2500, "7.jpg")
    z = np.zeros(g.get_n())
    for (a, b) in ((0, 10), (10, 13)):
        assert g.solve_iter_l2f(a, b, z, 0) == o.solve_iter_l2f(a, b, z, 0)
        assert bits_equal(g.get_x_iters_2d(10), o.get_x_iters_2d(10))
    assert not g.get_x_iters_2d(10)[:, 3:].any()
