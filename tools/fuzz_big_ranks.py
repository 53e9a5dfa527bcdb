"""Differential fuzz of the VARIABLE-SHARDED large-instance path: W = 2, 3 ranks on the one GPU (contributions exchanged over gloo through
the callback transport), odd-structured instances from tools/fuzz_big.py -- few variables per rank, row blocks of a handful of rows, empty
rows, very long rows -- reference arithmetic and the opt-in comm-lean PCG, against the oracle's rank model bit for bit.
usage: python tools/fuzz_big_ranks.py [cases=6] [seed=0] [first_case=0] [iterations=10]   (iterations = 20000: to convergence)"""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tools')]
import numpy as np


def _rank(rank, world, port, q, seed, case, mode, iters):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'), os.path.join(ROOT, 'tools')]
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fuzz_big import random_instance
    from lpbox_hip.big import BigLp
    rs = np.random.RandomState(seed)
    for _ in range(case + 1):
        I, kind = random_instance(rs)
    g = BigLp(I, rank, world, device=0, pcg_mode=mode)
    g.solve_init()
    ret = g.solve_iter(0, iters)
    q.put((rank, g.local_x(), g.vec("z1"), g.vec("z4"), g.scalar("cur_obj"), g.scalar("outer_total"), g.scalar("pcg_total"),
           g.scalar("threads"), g.scalar("chunk"), ret))
    dist.barrier(); dist.destroy_process_group()


def run_case(seed, case, world, mode, iters=10):
    import torch.multiprocessing as mp
    from helpers import bits_equal
    from oracle import oracle as O
    from fuzz_big import random_instance
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, seed, case, mode, iters)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    rs = np.random.RandomState(seed)
    for _ in range(case + 1):
        I, kind = random_instance(rs)
    o = O.LpOracle(0, order=O.ORDER_GPU, T=int(res[0][7]), chunk=int(res[0][8]), ranks=world)
    if mode == "lean":
        o.set_pcg_lean(True, 256)
    o.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"]); o.solve_init()
    ro = o.solve_iter(0, iters)
    ok = bits_equal(np.concatenate([r[1] for r in res]), o.vec("x")) and bits_equal(np.concatenate([r[2] for r in res]), o.vec("z1"))
    for r in res:
        ok = ok and bits_equal(r[3], o.vec("z4")) and r[4] == o.scalar("cur_obj") and (r[5], r[6]) == (o.total_outer_iters, o.total_pcg_iters) and r[9] == ro
    return ok, I, kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    bad = 0
    for c in range(first, first + cases):
        for world, mode in ((2, "reference"), (3, "lean")) if c % 2 == 0 else ((3, "reference"), (2, "lean")):
            ok, I, kind = run_case(seed, c, world, mode, iters)
            print("case %2d kind %d n %5d l %5d nnz %7d W %d %-9s: %s" % (c, kind, I["n"], I["l"], len(I["rowidx"]), world, mode, "ok" if ok else "MISMATCH"), flush=True)
            bad += not ok
    print("fuzz_big_ranks: %d runs differ" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
