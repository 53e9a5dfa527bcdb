"""Throughput vs batch size on one GPU (instances cycle through the 256-instance fixture)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np
from bench import load_instances, FIXTURE
from lpbox_hip.lp import LpBatch
insts = load_instances(FIXTURE)
for B in (64, 128, 256, 512, 1024, 2048):
    b = LpBatch([insts[i % 256] for i in range(B)])
    for rep in range(2):
        b.solve_init(); t = time.perf_counter(); b.solve_iter(0, 20000); dt = time.perf_counter() - t
    it = sum(b.counters(i)[0] for i in range(B))
    print("B=%5d: %.1f ms, %.2f M inst-iters/s" % (B, dt * 1e3, it / dt / 1e6))
    b.close()
