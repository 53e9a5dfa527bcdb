"""the CPU oracle on every host core (independent single-thread processes, one instance each at a time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from multiprocessing import Pool

def solve(k):
    from bench import load_instances, FIXTURE
    from oracle import oracle as O
    I = load_instances(FIXTURE)[k]
    s = O.LpOracle(0); s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f")); s.solve_init(); s.solve_iter(0, 20000)
    return s.total_outer_iters

if __name__ == "__main__":
    cores = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * cores
    from oracle import oracle as O
    O.build()
    with Pool(cores) as p:
        p.map(solve, range(cores))                  # warm-up (library load, fixture parse)
        t = time.perf_counter(); its = p.map(solve, range(n)); dt = time.perf_counter() - t
    print("CPU oracle on %d cores (%s): %d instances, %d iterations in %.1f s -> %.1f k instance-iterations/s" % (
        cores, open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t'), n, sum(its), dt, sum(its) / dt / 1e3))
