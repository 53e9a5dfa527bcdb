#!/usr/bin/env python3
"""The opt-in direct x-update against the default PCG mode on the 256-instance benchmark batch (GPU box):
wall clock, iteration counts, objectives (paired), instances at the iteration cap; optionally the worst instance of the direct mode
checked bit for bit against the oracle's mirror over its whole solve.

usage: python tools/direct_study.py [fixture.npz] [--check-worst]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "accelerated-lpbox-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from bench import FIXTURE, load_instances  # noqa: E402
from lpbox_hip.lp import LpBatch  # noqa: E402


def solve(insts, mode):
    b = LpBatch(insts)
    b.set_x_update(mode)
    b.solve_init(); b.solve_iter(0, 20000)                    # warm
    b.kernel_time(reset=True)
    b.solve_init()
    t0 = time.perf_counter(); b.solve_iter(0, 20000); dt = time.perf_counter() - t0
    n = len(insts)
    return dict(batch=b, ms=1e3 * dt, iters=np.array([b.counters(i)[0] for i in range(n)]),
                obj=np.array([-b.cal_obj(i) for i in range(n)]), stop=np.array([b.stop(i)[0] for i in range(n)]),
                infeasible=np.array([b.check_infeasible_l2f(i) for i in range(n)]))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    insts = load_instances(args[0] if args else FIXTURE)
    r = {m: solve(insts, m) for m in ("pcg", "direct")}
    for m, v in r.items():
        print("%-6s %.1f ms  iterations mean %.0f max %d (instance %d)  %.2f M instance-iterations/s  %.2f us per iteration of the slowest  "
              "mean objective %.3f  at the cap %d  infeasible %d" % (m, v["ms"], v["iters"].mean(), v["iters"].max(), v["iters"].argmax(),
              v["iters"].sum() / v["ms"] / 1e3, 1e3 * v["ms"] / v["iters"].max(), v["obj"].mean(), int((v["stop"] == 0).sum()), int((v["infeasible"] > 0).sum())))
    gap = (r["direct"]["obj"] - r["pcg"]["obj"]) / r["pcg"]["obj"]
    print("paired objective gap direct vs pcg: mean %+.4f  stderr %.4f  better/equal/worse %d/%d/%d" % (
        gap.mean(), gap.std(ddof=1) / np.sqrt(len(gap)), (gap > 0).sum(), (gap == 0).sum(), (gap < 0).sum()))
    if "--check-worst" in sys.argv:
        from helpers import bits_equal, oracle_for
        w = int(r["direct"]["iters"].argmax())
        b = LpBatch([insts[w]]); b.set_x_update("direct"); b.solve_init()
        o = oracle_for(b, 0, insts[w], x_update="direct", direct_rows=b.direct_rows(0))
        b.solve_iter(0, 20000); o.solve_iter(0, 20000)
        print("worst instance %d: gpu %s oracle mirror (%d, %d), x bit-identical: %s" % (w, b.counters(0), o.total_outer_iters, o.total_pcg_iters,
              bits_equal(b.debug_vec("x"), o.vec("x"))))


if __name__ == "__main__":
    main()
