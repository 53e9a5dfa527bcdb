"""Instance sharding across the GPUs of a node: one process per GPU, every rank solves its own contiguous block of
instances, NO data-path collective (instances share nothing -- each reference solver object is self-contained,
LPh:199-262).  torch.distributed is only used to gather the per-instance results (tiny) at the end; the backend is
whatever the caller initialised ("nccl" = RCCL on the GPUs, "gloo" on CPU for tests)."""
import numpy as np


def shard_range(total, world, rank):
    """Contiguous block [lo, hi) of `total` items owned by `rank` (the first total % world ranks hold one extra)."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedLpBatch:
    """Solve a list of instances sharded over the ranks of the default process group.

    solver_factory(instances) must return an object with the LpBatch interface (solve_init, solve_iter, counters,
    cal_obj, get_x_sol); the default is the HIP LpBatch on this rank's GPU.
    """

    def __init__(self, instances, rank=0, world=1, device=None, solver_factory=None):
        self.total = len(instances)
        self.rank, self.world = int(rank), int(world)
        self.lo, self.hi = shard_range(self.total, self.world, self.rank)
        mine = instances[self.lo:self.hi]
        if solver_factory is None:
            from .lp import LpBatch

            def solver_factory(insts):
                return LpBatch(insts, device=device)
        self.local = solver_factory(mine) if mine else None

    def solve(self, max_iters=20000):
        """ADMM_lp_iters_init + ADMM_lp_iters(0, max_iters) on the local shard; returns local (rets, iters, objs)."""
        n_loc = self.hi - self.lo
        rets = np.zeros(n_loc, np.int64)
        iters = np.zeros(n_loc, np.int64)
        objs = np.zeros(n_loc, np.float64)
        if self.local is not None:
            self.local.solve_init()
            rets[:] = self.local.solve_iter(0, max_iters)
            for i in range(n_loc):
                iters[i] = self.local.counters(i)[0]
                objs[i] = self.local.cal_obj(i)
        return rets, iters, objs

    def gather(self, rets, iters, objs):
        """All ranks receive the results of all instances in original order (one small all_gather, control plane only)."""
        if self.world == 1:
            return rets, iters, objs
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        cap = -(-self.total // self.world)
        buf = torch.zeros(3, cap, dtype=torch.float64, device=dev)
        k = self.hi - self.lo
        buf[0, :k] = torch.as_tensor(rets, dtype=torch.float64)
        buf[1, :k] = torch.as_tensor(iters, dtype=torch.float64)
        buf[2, :k] = torch.as_tensor(objs, dtype=torch.float64)
        out = [torch.zeros_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf)
        R, I, Ob = [], [], []
        for r in range(self.world):
            lo, hi = shard_range(self.total, self.world, r)
            o = out[r].cpu().numpy()
            R.append(o[0, :hi - lo]); I.append(o[1, :hi - lo]); Ob.append(o[2, :hi - lo])
        return (np.concatenate(R).astype(np.int64), np.concatenate(I).astype(np.int64), np.concatenate(Ob))
