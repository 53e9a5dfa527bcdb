"""Segmentation throughput with several solver handles driven from Python threads (one HIP stream per handle; ctypes releases the GIL)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.seg import PyLPboxADMMsolver, load_gray
imgs = [load_gray(os.path.join(ROOT, 'tests', 'golden', 'seg', f)) for f in ('0.jpg', '7.jpg')]
def make(k):
    g = imgs[k % 2]
    s = PyLPboxADMMsolver(0, g.size, 0); s.set_image(g); s.solve_init(); return s
for nthreads in (1, 2, 4, 8):
    solvers = [make(k) for k in range(nthreads)]
    for s in solvers: s.solve_iter()                      # warm-up (graph instantiation)
    solvers = [make(k) for k in range(nthreads)]
    res = [None] * nthreads
    def run(k): res[k] = solvers[k].solve_iter()
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(k,)) for k in range(nthreads)]
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    print("%d concurrent full-resolution solves: %.1f ms total, %.1f ms per image, energies %s" % (nthreads, dt * 1e3, dt * 1e3 / nthreads, res[:2]))
