"""Stand-in for lpbox_hip.lp.LpBatch used ONLY by tests/test_bench_launch.py (LPBOX_BENCH_STUB=1): it lets `bench.py --gpus N` run its
launch path -- child processes, process group, barriers, the max-over-ranks timing and the one JSON line of rank 0 -- on a box with
no GPU.  It computes nothing; its counters are a fixed function of the instance so the line's totals can be checked."""
import time


class StubBatch:
    def __init__(self, instances, device=None):
        self.insts = list(instances)
        self.ms = 0.0
        self.launches = 0

    def config(self):
        return {"threads": 512, "elems_per_thread": 1, "lds_bytes": 0}

    def solve_init(self):
        return 1

    def solve_iter(self, i, j):
        time.sleep(0.002)
        self.ms += 2.0
        self.launches += 1
        return [0] * len(self.insts)

    def kernel_time(self, reset=False):
        out = (self.ms, self.launches)
        if reset:
            self.ms, self.launches = 0.0, 0
        return out

    def counters(self, idx=0):
        return 100 + self.insts[idx]["nnz"] % 7, 1500

    def cal_obj(self, idx=0):
        return -1.0 - idx

    def check_infeasible_l2f(self, idx=0):
        return 0

    def stop(self, idx=0):
        return (1, 0)

    def close(self):
        pass
