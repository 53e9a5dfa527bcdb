"""Bit-exactness of the LP window kernel against the oracle for the workgroup geometry selected by LPBOX_LP_THREADS (tuning aid):
first instance of the j=100/k=500 (or, with argument 4, j=500/k=2000) fixture, a 100-iteration l2f window plus 400 plain iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.lp import PyLPboxADMMsolver
from oracle import oracle as O
fx = "lp_500_2000_seed0.npz" if len(sys.argv) > 1 and sys.argv[1] == "4" else "lp_100_500_seed0.npz"
I = O.load_lp_batch(os.path.join(ROOT, "tests", "golden", fx))[0]
g = PyLPboxADMMsolver(0)
g.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"]); g.solve_init()
cfg = g.batch.config()
o = O.LpOracle(0, order=O.ORDER_GPU, T=cfg["threads"], positions=g.batch.layout(0), npos=cfg["threads"] * cfg["elems_per_thread"],
               row_split=g.batch.row_split(0), col_split=g.batch.col_split(0))
o.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"]); o.solve_init()
vec = np.zeros(I["n"])
rg, ro = g.solve_iter_l2f(0, 100, vec, 0), o.solve_iter_l2f(0, 100, vec, 0)
xg, xo = g.get_x_iters_2d(100), o.get_x_iters_2d(100)
ok1 = rg == ro and np.array_equal(xg.view(np.uint64), xo.view(np.uint64))
rg, ro = g.solve_iter(100, 500), o.solve_iter(100, 500)
ok2 = rg == ro and np.array_equal(g.get_x_sol().ravel(), o.get_x_sol().ravel()) and g.cal_Obj() == o.cal_Obj()
print(f"geometry {cfg['threads']}x{cfg['elems_per_thread']}: l2f window {'bit-exact' if ok1 else 'DIFFERS'}, plain window {'bit-exact' if ok2 else 'DIFFERS'}")
sys.exit(0 if ok1 and ok2 else 1)
