"""The WHOLE 2048-instance stream of the headline size (BASELINE's 8 x 256 j=100/k=500 draws; lpbox_hip/auction.py, digest-checked against the
reference generator) solved to convergence by the HIP kernel, every instance compared with the oracle in the kernel's order: return code,
iteration counts, final iterate bit for bit, objective, binary solution.  The oracle solves run in a process pool on the host cores.
usage: python tools/stream_parity.py [first_rank=1] [last_rank=7] [items=100] [bids=500]"""
import multiprocessing, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tests')]
import numpy as np


def main():
    from concurrent.futures import ProcessPoolExecutor
    from helpers import bits_equal, oracle_full_solve
    from lpbox_hip import auction
    from lpbox_hip.lp import LpBatch
    r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    r1 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    items = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    bids = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    bad = total = 0
    with ProcessPoolExecutor(min(16, os.cpu_count() or 1), mp_context=multiprocessing.get_context("spawn")) as ex:
        for rank in range(r0, r1 + 1):
            t0 = time.time()
            insts = auction.stream_instances(items, bids, 256 * rank, 256, workers=16)
            B = LpBatch(insts)
            B.solve_init()
            rets = B.solve_iter(0, 20000)
            cfg = B.config()
            jobs = [(I, cfg["threads"], cfg["threads"] * cfg["elems_per_thread"], B.layout(i), B.row_split(i), B.col_split(i)) for i, I in enumerate(insts)]
            res = list(ex.map(oracle_full_solve, jobs, chunksize=2))
            nb = 0
            for i, (ret, outer, pcg, obj, x, xs) in enumerate(res):
                ok = (int(rets[i]) == ret and B.counters(i) == (outer, pcg) and bits_equal(B.debug_vec("x", i), x) and B.cal_obj(i) == obj
                      and np.array_equal(B.get_x_sol(i).ravel(), xs))
                nb += not ok
            bad += nb; total += len(insts)
            print("rank %d (draws %d..%d of %d/%d): %d of %d instances differ from the oracle, %d oracle iterations, %.0f s"
                  % (rank, 256 * rank, 256 * rank + 255, items, bids, nb, len(insts), sum(r[1] for r in res), time.time() - t0), flush=True)
    print("stream parity: %d of %d instances differ" % (bad, total))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
