"""The product loop on the 256-instance batch: windows of 100 ADMM iterations + the early-fixing policy on the device
(random-initialised GraphAttentionEncoder weights, torch.manual_seed-free generator seed 0: the reference ships no checkpoint)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np, torch
from bench import load_instances, FIXTURE
from lpbox_hip.lp import LpBatch
from lpbox_hip import l2f
from lpbox_hip.policy import EarlyFixPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "net"
xupd = sys.argv[3] if len(sys.argv) > 3 else "pcg"       # "direct": the opt-in exact x-update
insts = load_instances(FIXTURE)[:B]
from lpbox_hip.policy import FusedEarlyFixPolicy
pol = FusedEarlyFixPolicy.random(tokens=20, seed=0) if mode == "fused" else EarlyFixPolicy.random(tokens=20, seed=0, device="cuda", dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
score = pol if mode in ("net", "bf16", "fused") else (lambda x: x[:, -1, -1])          # "last": confident where the newest iterate is near 0/1
for rep in range(2):
    b = LpBatch(insts); b.set_x_update(xupd); b.solve_init(); tm = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = l2f.run_l2f_batch(b, score, ws=100, timing=tm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = sum(b.counters(i)[0] for i in range(B))
    fixed = sum(b.get_org_n(i) - b.get_n(i) for i in range(B))
    print(xupd, "rep %d: %.1f ms total (solve %.1f, policy %.1f, host %.1f), %d windows, %d outer iterations, %.2f M inst-iters/s, fixed %d of %d variables, mean objective %.2f, infeasible %d" % (
        rep, dt * 1e3, tm["solve"] * 1e3, tm["policy"] * 1e3, tm["host"] * 1e3, res["windows"], it, it / dt / 1e6, fixed, sum(I["n"] for I in insts), res["objective"].mean(), int((res["infeasible"] > 0).sum())))
    b.close()
# plain solve beside it
b = LpBatch(insts); b.solve_init(); t0 = time.perf_counter(); b.solve_iter(0, 20000); dt = time.perf_counter() - t0
print("plain solve: %.1f ms, mean objective %.2f" % (dt * 1e3, np.mean([-b.cal_obj(i) for i in range(B)])))
