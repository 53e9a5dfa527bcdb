"""Config 4: 256 instances j=500/k=2000 per GPU.  Only 4 generated instances are committed (the reference's
generator needs ~4 s each), so the batch replicates them 64x: identical work per workgroup, same kernel, same occupancy."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np
from bench import load_instances, byte_model
from lpbox_hip.lp import LpBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
base = load_instances(os.path.join(ROOT, "tests", "golden", "lp_500_2000_seed0.npz"))
insts = [base[i % len(base)] for i in range(B)]
b = LpBatch(insts)
print("config", b.config())
for r in range(2):
    b.solve_init(); b.kernel_time(reset=True); b.solve_iter(0, N)
    ms, _ = b.kernel_time()
    o = np.array([b.counters(i) for i in range(B)], float)
bytes_alg = 0.0
for i in range(B):
    bf, bp = byte_model(insts[i]); bytes_alg += bf * o[i, 0] + bp * o[i, 1]
print("B=%d window %d: %.2f ms; outer iters mean %.0f max %.0f; pcg/outer %.2f; %.3f M inst-iters/s; %.2f us per outer iter (slowest); algorithmic %.0f GB/s" % (
    B, N, ms, o[:, 0].mean(), o[:, 0].max(), o[:, 1].sum() / o[:, 0].sum(), o[:, 0].sum() / ms / 1e3, 1e3 * ms / o[:, 0].max(), bytes_alg / ms / 1e6))
