"""Config 5 shape on ONE GPU: a single auction-like LP with n variables solved by the large-instance path (world = 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.big import BigLp
from lpbox_hip.synth import make_auction_like
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
P = make_auction_like(n, 0)
g = BigLp(P); g.solve_init()
g.solve_iter(0, 10)
t = time.perf_counter(); g.solve_iter(10, 10 + iters); dt = time.perf_counter() - t
o, p = g.scalar("outer_total") - 10, g.scalar("pcg_total")
nnz, l = len(P['rowidx']), P['l']
mE = 12 * nnz + 4 * (l + 1); mEt = 12 * nnz + 4 * (n + 1)
K = p / g.scalar("outer_total")
b_iter = 3 * mE + 2 * mEt + 8 * (30 * n + 12 * l) + K * (mE + mEt + 8 * (13 * n + 2 * l))
print(f"n={n} l={l} nnz={nnz}: {iters} iterations in {dt*1e3:.1f} ms -> {iters/dt:.1f} iters/s, {dt/iters*1e3:.2f} ms/iter, K={K:.1f}, "
      f"algorithmic {b_iter*iters/dt/1e9:.0f} GB/s, launches/iter {g.scalar('launches')/g.scalar('outer_total'):.0f}, kmax {g.scalar('kmax'):.0f}")
if len(sys.argv) > 3:
    from oracle import oracle as O
    s = O.LpOracle(0); s.set_problem(P['n'], P['l'], P['colptr'], P['rowidx'], P['b']); s.solve_init()
    t = time.perf_counter(); s.solve_iter(0, 5); dt = time.perf_counter() - t
    print(f"cpu oracle: 5 iterations in {dt:.1f} s -> {5/dt:.2f} iters/s")
