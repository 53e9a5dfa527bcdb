#!/usr/bin/env python3
"""Scratch check of the direct x-update on the GPU box: bit-exactness vs the oracle's mirror + timing of the 256-batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "accelerated-lpbox-admm_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import lp_instances, oracle_for, bits_equal
from lpbox_hip.lp import LpBatch

insts = lp_instances("lp_100_500_seed0.npz")
b = LpBatch(insts[:4]); b.set_x_update("direct"); b.solve_init()
os_ = [oracle_for(b, i, insts[i], x_update="direct", direct_rows=b.direct_rows(i)) for i in range(4)]
print("G rows:", [int((b.direct_rows(i) >= 0).sum()) for i in range(4)], b.config())
for w in range(3):
    r = b.solve_iter(w * 50, (w + 1) * 50)
    for i, o in enumerate(os_):
        ro = o.solve_iter(w * 50, (w + 1) * 50)
        xg, xo = b.debug_vec("x", i), o.vec("x")
        print("win", w, "inst", i, "ret", r[i], ro, "x bits", bits_equal(xg[: len(xo)] if len(xg) != len(xo) else xg, xo), "maxdiff", np.abs(xg[:len(xo)] - xo).max(), b.counters(i), (o.total_outer_iters, o.total_pcg_iters))
B = LpBatch(insts); B.set_x_update("direct"); B.solve_init()
t = time.perf_counter(); B.solve_iter(0, 20000); dt = time.perf_counter() - t
its = np.array([B.counters(i)[0] for i in range(len(insts))])
obj = np.array([-B.cal_obj(i) for i in range(len(insts))])
inf = np.array([B.check_infeasible_l2f(i) for i in range(len(insts))])
print("direct full solve: %.1f ms, iters mean %.0f max %d, mean obj %.3f, infeasible %d, kernel ms %s" % (1e3 * dt, its.mean(), its.max(), obj.mean(), (inf != 0).sum(), B.kernel_time()))
B2 = LpBatch(insts); B2.solve_init()
t = time.perf_counter(); B2.solve_iter(0, 20000); dt2 = time.perf_counter() - t
its2 = np.array([B2.counters(i)[0] for i in range(len(insts))]); obj2 = np.array([-B2.cal_obj(i) for i in range(len(insts))])
print("pcg    full solve: %.1f ms, iters mean %.0f max %d, mean obj %.3f" % (1e3 * dt2, its2.mean(), its2.max(), obj2.mean()))
d = obj - obj2
print("paired obj diff direct - pcg: mean %.3f sem %.3f median %.3f" % (d.mean(), d.std() / np.sqrt(len(d)), np.median(d)))
