"""world_size-2 test of the N>1 path on CPU (gloo): instance sharding with no data-path collective; only the tiny result
gather uses the process group.  The per-rank solver is the CPU oracle standing in for the HIP batch (same interface)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import lp_instances, make_oracle


class OracleBatch:
    """LpBatch-shaped wrapper around independent oracle solvers (test stand-in only)."""

    def __init__(self, insts):
        self.s = [make_oracle(I) for I in insts]

    def solve_init(self):
        return 1

    def solve_iter(self, i, j):
        return np.array([s.solve_iter(i, j) for s in self.s])

    def counters(self, k):
        return self.s[k].total_outer_iters, self.s[k].total_pcg_iters

    def cal_obj(self, k):
        return self.s[k].cal_Obj()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lpbox_hip.dist import ShardedLpBatch
    insts = lp_instances("lp_20_60_seed0.npz")[:5]
    sb = ShardedLpBatch(insts, rank, world, solver_factory=OracleBatch)
    res = sb.gather(*sb.solve(20000))
    if rank == 0:
        q.put((sb.lo, sb.hi, [r.tolist() for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_instance_sharding_matches_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lo, hi, (rets, iters, objs) = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (lo, hi) == (0, 3)
    ref = OracleBatch(lp_instances("lp_20_60_seed0.npz")[:5])
    r0 = ref.solve_iter(0, 20000)
    assert rets == r0.tolist()
    assert iters == [ref.counters(k)[0] for k in range(5)]
    assert objs == [ref.cal_obj(k) for k in range(5)]
