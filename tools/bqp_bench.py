"""The generic constrained-BQP path on larger problems: (a) the full-resolution segmentation problem as `unconstrained` beside the
specialised segmentation path; (b) an auction LP with 1e5 variables as `linear_ineq`; (c) a clustering-style problem with one-of-k
equality constraints."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from lpbox_hip.bqp import BqpSolver
from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
from lpbox_hip.synth import make_auction_like

def report(tag, g, t):
    it = g.scalar("iters"); print("%s: %d iterations (stop %d) in %.1f ms wall, %.1f ms stream, %.1f us/iteration, %.0f launches/iteration, PCG/iter %.2f" % (
        tag, it, g.scalar("stop"), t * 1e3, g.scalar("kernel_ms"), g.scalar("kernel_ms") * 1e3 / max(it, 1), g.scalar("launches") / max(g.scalar("outer_total"), 1), g.scalar("total_pcg") / max(g.scalar("outer_total"), 1)))

gray = load_gray(os.path.join(ROOT, 'tests', 'golden', 'seg', '0.jpg'))
s = PyLPboxADMMsolver(0, gray.size, 0); s.set_image(gray); S = s.get_problem(); s.solve_init()
t = time.perf_counter(); s.solve_iter(); ts = time.perf_counter() - t
g = BqpSolver(S["n"], (S["rowptr"], S["colidx"], S["vals"]), S["b"], np.zeros(S["n"]))
for rep in range(2):
    t = time.perf_counter(); g.solve(); tg = time.perf_counter() - t
report("(a) segmentation n=%d as unconstrained BQP" % S["n"], g, tg)
print("    specialised segmentation path: %.1f ms; same binary solution: %s" % (ts * 1e3, np.array_equal((g.vec("x") >= 0.5).astype(float), np.asarray(s.get_x_sol()).ravel())))

P = make_auction_like(100000, 0)
n, l = P["n"], P["l"]
cols = np.repeat(np.arange(n), np.diff(P["colptr"])); order = np.lexsort((cols, P["rowidx"]))
Er = np.concatenate([[0], np.cumsum(np.bincount(P["rowidx"], minlength=l))]).astype(np.int32)
A = (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.zeros(n))
g = BqpSolver(n, A, P["b"], np.ones(n), E=(Er, cols[order].astype(np.int32), np.ones(len(order))), f=np.ones(l),
              params=[1e-4, 1e-6, 1.6, 0.95, 5, 300, 25, 3, 1.01, 1e-4, 1000])
t = time.perf_counter(); g.solve(); tg = time.perf_counter() - t
report("(b) auction LP n=%d l=%d as linear_ineq BQP, 300 iterations" % (n, l), g, tg)

rs = np.random.RandomState(0)
pts, k = 20000, 8                                    # assign 20000 points to 8 clusters: x[p*k + c], one cluster per point, quadratic within-cluster cost
n = pts * k
feat = rs.randn(pts, 2) + rs.randint(0, k, pts)[:, None] * 3.0
nbr = [rs.choice(pts, 6, replace=False) for _ in range(pts)]
rows, colsA, vals = [], [], []
for p in range(pts):
    for q in nbr[p]:
        if q == p: continue
        w = float(np.sum((feat[p] - feat[q]) ** 2)) * 0.05
        for c in range(k):
            rows += [p * k + c, q * k + c]; colsA += [q * k + c, p * k + c]; vals += [w, w]
import scipy.sparse as sp
Am = sp.coo_matrix((vals, (rows, colsA)), shape=(n, n)).tocsr()
Am = Am + sp.diags(np.asarray(np.abs(Am).sum(axis=1)).ravel() + 0.1)
Cp = np.arange(0, n + 1, k, dtype=np.int32)
g = BqpSolver(n, Am, rs.uniform(-0.1, 0.1, n), np.full(n, 1.0 / k), C_=(Cp, np.arange(n, dtype=np.int32), np.ones(n)), d=np.ones(pts),
              params=[1e-4, 1e-6, 1.6, 0.95, 5, 1000, 1, 3, 1.05, 1e-4, 1000])
t = time.perf_counter(); g.solve(); tg = time.perf_counter() - t
xb = (g.vec("x") >= 0.5).reshape(pts, k)
report("(c) one-of-%d assignment, n=%d m=%d as linear_eq BQP" % (k, n, pts), g, tg)
print("    points with exactly one cluster: %.3f" % np.mean(xb.sum(axis=1) == 1))
