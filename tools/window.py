"""Tuning helper: time a fixed window of iterations (no instance converges that early) on the 256-instance batch."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np
from bench import load_instances, FIXTURE, byte_model
from lpbox_hip.lp import LpBatch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
insts=load_instances(FIXTURE)[:256]
b=LpBatch(insts)
for r in range(reps):
    b.solve_init(); b.kernel_time(reset=True); b.solve_iter(0,N)
    ms,_=b.kernel_time()
    o=np.array([b.counters(i) for i in range(256)],float)
print("window %d: %.2f ms  -> %.2f us/outer-iter, %.3f us/pcg-iter, %.2f M inst-iters/s, pcg/outer %.2f" % (N, ms, 1e3*ms/N, 1e3*ms/(o[:,1].mean()), o[:,0].sum()/ms/1e3, o[:,1].sum()/o[:,0].sum()))
