import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "accelerated-lpbox-admm_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import lp_instances, oracle_for, bits_equal
from lpbox_hip.lp import LpBatch
insts = lp_instances("lp_100_500_seed0.npz")
for idx in (57, 3, 117):
    b = LpBatch([insts[idx]]); b.set_x_update("direct"); b.solve_init()
    o = oracle_for(b, 0, insts[idx], x_update="direct", direct_rows=b.direct_rows(0))
    b.solve_iter(0, 20000); o.solve_iter(0, 20000)
    print(idx, "gpu", b.counters(0), b.stop(0), "oracle", o.total_outer_iters, o.last_stop_reason, "x equal", bits_equal(b.debug_vec("x"), o.vec("x")),
          "cvg1 %.3e cvg2 %.3e std %.3e" % (b.debug_scalar("cvg1"), b.debug_scalar("cvg2"), b.debug_scalar("std_obj")))
