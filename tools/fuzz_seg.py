"""Differential fuzz of the segmentation flavour against the SEG oracle in the kernels' order: random gray images of awkward shapes
(one row, one column, two columns, primes, wide, tall), several node budgets for the resize; an early-fixing window pair and the legacy
solve to convergence, bit for bit.  usage: python tools/fuzz_seg.py [count=24] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tests')]
import numpy as np

SHAPES = [(1, 37), (41, 1), (2, 2), (3, 2), (2, 9), (7, 13), (16, 16), (33, 65), (5, 200), (120, 7), (64, 48), (97, 101)]


def random_image(rs):
    h, w = SHAPES[rs.randint(0, len(SHAPES))]
    kind = int(rs.randint(0, 4))
    if kind == 0:
        img = rs.randint(0, 256, (h, w))
    elif kind == 1:                                      # two flat regions + noise: a real segmentation
        img = np.where(np.add.outer(np.arange(h), np.arange(w)) < (h + w) // 2, 40, 200) + rs.randint(-20, 21, (h, w))
    elif kind == 2:
        img = np.full((h, w), int(rs.randint(0, 256)))   # constant image: all weights maximal
    else:
        img = (rs.rand(h, w) < 0.5) * 255                # salt and pepper
    nodes = int(rs.choice([h * w, max(4, h * w // 3), 2500, 10000]))
    return np.clip(img, 0, 255).astype(np.uint8), nodes, kind


def check(gray, nodes, legacy=True):
    from helpers import bits_equal, scripted_fix_vec
    from oracle import oracle as O
    from lpbox_hip.seg import PyLPboxADMMsolver

    def pair():
        g = PyLPboxADMMsolver(0, nodes, 0)
        g.set_image(gray)
        P = g.get_problem()
        g.solve_init()
        cfg = g.config()
        o = O.SegOracle(0, nodes, 0, order=O.ORDER_GPU, T=cfg["threads"], chunk=cfg["threads"] * cfg["elems_per_thread"])
        o.set_problem(P); o.solve_init()
        return g, o
    g, o = pair()
    ok = True
    vec, num = np.zeros(g.get_org_n()), 0
    for w in range(2):
        rg, ro = g.solve_iter_l2f(w * 10, (w + 1) * 10, vec, num), o.solve_iter_l2f(w * 10, (w + 1) * 10, vec, num)
        xg, xo = g.get_x_iters_2d(10), o.get_x_iters_2d(10)
        ok = ok and rg == ro and xg.shape == xo.shape and bits_equal(xg, xo)
        if not ok or rg:
            break
        vec, num = scripted_fix_vec(xg, lo=0.05, hi=0.95, last=5)
    n = g.get_org_n()
    g.close()
    if legacy:
        g, o = pair()
        og, oo = g.get_obj(), o.get_obj()
        ok = ok and g.solve_iter() == o.solve_iter() and np.array_equal(g.get_x_sol(), o.get_x_sol())
        og, oo = g.get_obj(), o.get_obj()
        ok = ok and (og == oo or (np.isnan(og) and np.isnan(oo))) and g.counters() == (o.total_outer_iters, o.total_pcg_iters)
        g.close()
    return ok, n


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for t in range(count):
        gray, nodes, kind = random_image(rs)
        try:
            # a constant image gives the reference's cost construction a zero variance: every weight is NaN (SEGcpp:46-248), the solver runs
            # to its caps on NaNs (10^4 iterations x 10^3 PCG steps: minutes) -- windows only for those
            ok, n = check(gray, nodes, legacy=kind != 2)
            msg = "ok" if ok else "MISMATCH"
        except Exception as e:
            refused = "too small" in str(e)               # an image that scales to one pixel in a direction is refused loudly: not a solve
            ok, n, msg = refused, -1, ("refused: %s" if refused else "ERROR %s") % str(e)[:120]
        print("case %3d kind %d image %3dx%-3d nodes %5d -> n %6d: %s" % (t, kind, gray.shape[0], gray.shape[1], nodes, n, msg), flush=True)
        bad += not ok
    print("fuzz_seg: %d of %d cases differ" % (bad, count))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
