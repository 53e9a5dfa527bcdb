"""Differential fuzz of the large-instance LP path (lpbox_big_*, one rank) against the oracle in that path's order: random sparse patterns
with empty rows, very long rows (thousands of entries: runs longer than the unrolled six inside a column slice), one-entry columns,
duplicate columns; plain windows and an early-fixing window.  usage: python tools/fuzz_big.py [count=24] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tests')]
import numpy as np


def random_instance(rs):
    kind = int(rs.randint(0, 5))
    n = int(rs.choice([50, 257, 1000, 2500, 4097, 9000, 20000, 33000]))
    l = max(2, int(n * rs.uniform(0.05, 0.6)))
    k = rs.randint(1, 7, n) if kind != 2 else np.ones(n, int)
    rows = [np.unique(rs.randint(0, l, kk)) for kk in k]
    if kind in (1, 4):                                   # a few very long rows
        for r in rs.choice(l, size=3, replace=False):
            for j in np.nonzero(rs.rand(n) < rs.uniform(0.2, 0.9))[0]:
                rows[j] = np.union1d(rows[j], [r])
    if kind == 3:
        for j in range(1, n):
            if rs.rand() < 0.2:
                rows[j] = rows[j - 1]
    if kind in (0, 4):                                   # empty rows in the middle
        dead = np.setdiff1d(rs.choice(l, size=max(1, l // 4), replace=False), [l - 1])
        rows = [np.setdiff1d(r, dead) if len(np.setdiff1d(r, dead)) else np.array([l - 1]) for r in rows]
    l = int(max(r.max() for r in rows)) + 1
    colptr = np.zeros(n + 1, np.int32); colptr[1:] = np.cumsum([len(r) for r in rows])
    rowidx = np.concatenate(rows).astype(np.int32)
    return dict(n=n, l=l, colptr=colptr, rowidx=rowidx, b=-rs.uniform(1, 500, n)), kind


def check(I, slice_kb=None):
    from helpers import bits_equal, scripted_fix_vec
    from oracle import oracle as O
    from lpbox_hip.big import BigLp
    if slice_kb is not None:
        os.environ["LPBOX_BIG_SLICE_KB"] = str(slice_kb)
    g = BigLp(I)
    os.environ.pop("LPBOX_BIG_SLICE_KB", None)
    g.solve_init()
    o = O.LpOracle(0, order=O.ORDER_GPU, T=int(g.scalar("threads")), chunk=int(g.scalar("chunk")))
    o.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"]); o.solve_init()
    ok = g.solve_iter(0, 25) == o.solve_iter(0, 25)
    for name in ("x", "z1", "z2", "z4"):
        ok = ok and bits_equal(g.vec(name), o.vec(name))
    vec, num = np.zeros(I["n"]), 0
    for w in range(2):
        ok = ok and g.solve_iter_l2f(25 + 40 * w, 65 + 40 * w, vec, num) == o.solve_iter_l2f(25 + 40 * w, 65 + 40 * w, vec, num)
        xg, xo = g.get_x_iters_2d(40), o.get_x_iters_2d(40)
        ok = ok and xg.shape == xo.shape and bits_equal(xg, xo)
        if not ok:
            break
        vec, num = scripted_fix_vec(xg, lo=0.05, hi=0.95, last=15)
    ok = ok and (g.scalar("outer_total"), g.scalar("pcg_total")) == (o.total_outer_iters, o.total_pcg_iters)
    g.close()
    return ok


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for t in range(count):
        I, kind = random_instance(rs)
        skb = [None, 8, 64][t % 3]
        ok = check(I, skb)
        print("case %3d kind %d n %5d l %5d nnz %7d slice_kb %s: %s" % (t, kind, I["n"], I["l"], len(I["rowidx"]), skb, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("fuzz_big: %d of %d cases differ" % (bad, count))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
