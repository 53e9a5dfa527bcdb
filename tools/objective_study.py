#!/usr/bin/env python3
"""GPU order vs Eigen order on the whole benchmark batch (run on the GPU box): every instance of a fixture solved to convergence by
the HIP solver (its own summation order) and by the CPU oracle in Eigen's order (the reference's arithmetic as far as it can be
restated).  Writes gpurun_out/objective_study_<name>.npz: per instance objective (maximisation sign),
outer iterations, stop reason, infeasible rows -- the data behind tests/test_objective_gap.py and north_star's "objective gap <= reference".

With a trailing "direct" both sides use the opt-in direct x-update (the oracle in Eigen's order with the row
split the library chose) -> gpurun_out/objective_study_direct_<name>.npz, the data behind tests/test_objective_gap.py's direct-mode tests.

usage: python tools/objective_study.py [fixture.npz] [count] [workers] [direct]
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "accelerated-lpbox-admm_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from bench import load_instances  # noqa: E402


def eigen_solve(job):
    from oracle import oracle as O
    I, rows = job
    s = O.LpOracle(0, order=O.ORDER_EIGEN, x_update="direct" if rows is not None else "pcg", direct_rows=rows)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    s.solve_init()
    s.solve_iter(0, 20000)
    return -s.cal_Obj(), s.total_outer_iters, s.last_stop_reason, s.check_infeasible_l2f(), s.total_pcg_iters


def main():
    direct = sys.argv[-1] == "direct"
    argv = sys.argv[:-1] if direct else sys.argv
    fixture = argv[1] if len(argv) > 1 else os.path.join(ROOT, "tests", "golden", "lp_100_500_seed0.npz")
    insts = load_instances(fixture)
    count = int(argv[2]) if len(argv) > 2 else len(insts)
    workers = int(argv[3]) if len(argv) > 3 else min(16, os.cpu_count() or 1)
    insts = insts[:count]
    from lpbox_hip.lp import LpBatch
    from oracle import oracle as O
    O.build()
    b = LpBatch(insts)
    if direct:
        b.set_x_update("direct")
    b.solve_init()
    t0 = time.perf_counter()
    b.solve_iter(0, 20000)
    t_gpu = time.perf_counter() - t0
    g_obj = np.array([-b.cal_obj(i) for i in range(count)])
    g_it = np.array([b.counters(i)[0] for i in range(count)])
    g_pcg = np.array([b.counters(i)[1] for i in range(count)])
    g_stop = np.array([b.stop(i)[0] for i in range(count)])
    g_inf = np.array([b.check_infeasible_l2f(i) for i in range(count)])
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as ex:
        res = list(ex.map(eigen_solve, [(I, b.direct_rows(i) if direct else None) for i, I in enumerate(insts)], chunksize=2))
    t_cpu = time.perf_counter() - t0
    e_obj, e_it, e_stop, e_inf, e_pcg = (np.array(v) for v in zip(*res))
    name = os.path.splitext(os.path.basename(fixture))[0]
    out = os.path.join(ROOT, "gpurun_out", "objective_study_%s%s.npz" % ("direct_" if direct else "", name))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, gpu_obj=g_obj, gpu_iters=g_it, gpu_pcg=g_pcg, gpu_stop=g_stop, gpu_infeasible=g_inf,
                        eigen_obj=e_obj, eigen_iters=e_it, eigen_pcg=e_pcg, eigen_stop=e_stop, eigen_infeasible=e_inf,
                        cfg=np.array(list(b.config().values())))
    gap = (g_obj - e_obj) / e_obj
    print("instances %d  gpu %.3f s  cpu(%d procs) %.1f s" % (count, t_gpu, workers, t_cpu))
    print("mean objective  gpu %.4f  eigen %.4f   mean paired gap %+.5f%% (s.e. %.5f%%)  gpu better/equal/worse %d/%d/%d" % (
        g_obj.mean(), e_obj.mean(), 100 * gap.mean(), 100 * gap.std(ddof=1) / np.sqrt(count),
        int((gap > 0).sum()), int((gap == 0).sum()), int((gap < 0).sum())))
    print("outer iterations gpu mean %.0f max %d (p90 %d, p99 %d) | eigen mean %.0f max %d (p90 %d, p99 %d)" % (
        g_it.mean(), g_it.max(), np.percentile(g_it, 90), np.percentile(g_it, 99), e_it.mean(), e_it.max(),
        np.percentile(e_it, 90), np.percentile(e_it, 99)))
    print("infeasible instances gpu %d eigen %d; stop reasons gpu %s eigen %s" % (
        int((g_inf > 0).sum()), int((e_inf > 0).sum()), np.bincount(g_stop, minlength=5).tolist(), np.bincount(e_stop, minlength=5).tolist()))
    print("wrote", out)


if __name__ == "__main__":
    main()
