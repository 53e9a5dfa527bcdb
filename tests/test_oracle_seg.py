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


def test_resize_restatement_against_an_independent_bilinear():
    """The restated cv::resize(src, dst, Size(), s, s) INTER_LINEAR on 8-bit data (oracle = product, previous test) against torch's bilinear
    interpolation driven with the SAME explicit scale factor -- both sample at src = (dst + 0.5) / s - 0.5 with replicated edges and no
    antialiasing; torch in float32, OpenCV's 8-bit path through 11-bit fixed-point weights and a rounded result.  Agreement within one grey
    level everywhere (measured: max 0.75, mean 0.26 = rounding to integers) pins the geometry of the resize -- centre convention, scale,
    orientation -- to an independent implementation; a half-pixel shift or the size-ratio scale torch uses by default is off by up to 187
    levels on these images.  (Output size: OpenCV rounds s * size, SEGcpp:713-714; torch floors, so the common region is compared.)  Not
    a reference-generated vector: there is no OpenCV here, the last bit of the fixed-point path stays unpinned."""
    import torch
    for name in ("0.jpg", "7.jpg"):
        g = gray(name)
        for nodes in (10000, 2500, 40000):
            scale = float(np.sqrt(nodes / g.size))
            small = O.seg_resize_u8(g, scale)
            assert small.shape == (int(round(g.shape[0] * scale)), int(round(g.shape[1] * scale)))
            t = torch.from_numpy(g.astype(np.float32))[None, None]
            ref = torch.nn.functional.interpolate(t, scale_factor=scale, mode="bilinear", align_corners=False, antialias=False,
                                                  recompute_scale_factor=False)[0, 0].numpy()
            r, c = min(ref.shape[0], small.shape[0]), min(ref.shape[1], small.shape[1])
            assert r >= small.shape[0] - 1 and c >= small.shape[1] - 1
            d = np.abs(small[:r, :c].astype(np.float64) - ref[:r, :c])
            assert d.max() < 1.0 and d.mean() < 0.35, (name, nodes, d.max(), d.mean())


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
