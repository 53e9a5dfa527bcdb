"""One LP of the config-5 family solved TO CONVERGENCE by the large-instance path and by the oracle in that path's order: stop reason,
iteration counts, final iterate bit for bit, objective.  usage: python tools/big_full_parity.py [n=100000] [seed=0]
(the oracle needs about 0.3 us per variable and iteration: n = 10^5 takes three to four minutes, silently)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
from helpers import bits_equal
from oracle import oracle as O
from lpbox_hip.big import BigLp
from lpbox_hip.synth import make_auction_like

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
P = make_auction_like(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = BigLp(P); g.solve_init()
t = time.time(); rg = g.solve_iter(0, 20000); tg = time.time() - t
o = O.LpOracle(0, order=O.ORDER_GPU, T=int(g.scalar("threads")), chunk=int(g.scalar("chunk")))
o.set_problem(P["n"], P["l"], P["colptr"], P["rowidx"], P["b"]); o.solve_init()
t = time.time()
import threading
stop_beat = threading.Event()
def beat():                                    # the C call below is silent for minutes; ctypes releases the GIL, so this keeps reporting
    while not stop_beat.wait(60):
        print("  oracle running, %.0f s" % (time.time() - t), flush=True)
threading.Thread(target=beat, daemon=True).start()
ro = o.solve_iter(0, 20000)                   # ONE call, like the HIP side: a resumed plain loop overwrites z4 in its first iteration (LPcpp:920-923)
to = time.time() - t
stop_beat.set()
ok = (int(g.scalar("stop")), int(g.scalar("outer_total")), int(g.scalar("pcg_total"))) == (o.last_stop_reason, o.total_outer_iters, o.total_pcg_iters)
ok = ok and bits_equal(g.local_x(), o.vec("x")) and g.cal_Obj() == o.cal_Obj()
print("n = %d, l = %d, nnz = %d: HIP %d outer / %d PCG iterations in %.2f s, oracle %d / %d in %.0f s; stop %d / %d; objective %.6f / %.6f -> %s"
      % (P["n"], P["l"], len(P["rowidx"]), g.scalar("outer_total"), g.scalar("pcg_total"), tg, o.total_outer_iters, o.total_pcg_iters, to,
         g.scalar("stop"), o.last_stop_reason, g.cal_Obj(), o.cal_Obj(), "IDENTICAL" if ok else "DIFFERENT"))
sys.exit(0 if ok else 1)
