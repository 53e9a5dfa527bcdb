"""Lottery-free timing of the LP window kernel: the first N ADMM iterations of the whole batch (no instance has stopped yet, so every
workgroup does the same number of outer iterations), repeated R times.  usage: python tools/window.py [N=2000] [R=2] [config=2|4]
Environment: LPBOX_LIB_VARIANT (experiment builds, csrc/Makefile `variant`), LPBOX_LP_THREADS (workgroup geometry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.lp import LpBatch
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fx = "lp_500_2000_seed0.npz" if cfg == 4 else "lp_100_500_seed0.npz"
insts = O.load_lp_batch(os.path.join(ROOT, "tests", "golden", fx))
insts = [insts[i % len(insts)] for i in range(256)]
b = LpBatch(insts)
c = b.config()
best = 1e9
for r in range(R + 1):
    b.solve_init()
    b.kernel_time(reset=True)
    b.solve_iter(0, N)
    ms, _ = b.kernel_time()
    if r: best = min(best, ms)
pcg = sum(b.counters(i)[1] for i in range(len(insts)))
outer = sum(b.counters(i)[0] for i in range(len(insts)))
print(f"window({N}) config {cfg} {c['threads']}x{c['elems_per_thread']}: {best:.2f} ms -> {1e3*best/N:.2f} us per outer iteration, K = {pcg/outer:.2f}")
