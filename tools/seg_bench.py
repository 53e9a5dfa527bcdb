"""Config 3 of BASELINE.json: full-resolution segmentation solve (n = 187 500) on one GPU, with the CPU oracle beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
gray = load_gray(os.path.join(ROOT, 'tests', 'golden', 'seg', sys.argv[1] if len(sys.argv) > 1 else '0.jpg'))
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else gray.size
cpu = len(sys.argv) > 3 and sys.argv[3] == 'cpu'
for rep in range(3):
    s = PyLPboxADMMsolver(0, nodes, 0); s.set_image(gray); P = s.get_problem(); s.solve_init(); s.kernel_time(reset=True)
    t = time.perf_counter(); e = s.solve_iter(); dt = time.perf_counter() - t
    o, p = s.counters(); ms, nl = s.kernel_time()
    n, nnz = P['n'], len(P['colidx'])
    mA = 12 * nnz + 4 * (n + 1)
    bytes_iter = 3 * mA + 256 * n + (p / o) * (mA + 104 * n)
    print(f"n={n} nnz={nnz} energy={e} outer={o} pcg/outer={p/o:.2f} wall={dt*1e3:.1f} ms stream={ms:.1f} ms launches={nl} "
          f"-> {o/dt:.0f} iters/s, {1e3*dt/o*1e3:.1f} us/iter, algorithmic {bytes_iter*o/dt/1e9:.0f} GB/s, {ms*1e3/max(nl,1):.2f} us/launch")
if cpu:
    from oracle import oracle as O
    so = O.SegOracle(0, nodes, 0); so.set_problem(P); so.solve_init()
    t = time.perf_counter(); so.solve_iter(); dt = time.perf_counter() - t
    print(f"cpu oracle: {so.total_outer_iters} iters in {dt:.1f} s -> {so.total_outer_iters/dt:.1f} iters/s")
