"""Config 5 on ONE GPU: a single auction-like LP with n variables, early-fixing loop with the fused policy (random-initialised
weights: the reference ships no checkpoint) or a stand-in that fixes (score = newest iterate)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from lpbox_hip.big import BigLp
from lpbox_hip.synth import make_auction_like
from lpbox_hip import l2f
from lpbox_hip.policy import FusedEarlyFixPolicy
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "fused"
P = make_auction_like(n, 0)
pol = FusedEarlyFixPolicy.random(tokens=20, seed=0)
score = pol if mode == "fused" else (lambda x: x[:, -1, -1])
g = BigLp(P, use_torch_stream=True); g.solve_init()
# time the pieces of one window
torch.cuda.synchronize(); t0 = time.perf_counter()
g.solve_iter_l2f(0, 100, None, 0)
torch.cuda.synchronize(); t1 = time.perf_counter()
X = g.x_iters_torch(100)
torch.cuda.synchronize(); t2 = time.perf_counter()
off = torch.arange(X.shape[0], device="cuda", dtype=torch.int64) * 100
sig = pol.scores_from_xiters(X.reshape(-1), off, 5)
torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"n={n}: window of 100 iterations {1e3*(t1-t0):.1f} ms; pack x_iters {1e3*(t2-t1):.1f} ms; fused policy on {X.shape[0]} variables {1e3*(t3-t2):.1f} ms")
g2 = BigLp(P, use_torch_stream=True); g2.solve_init()
torch.cuda.synchronize(); t0 = time.perf_counter()
res = l2f.run_l2f_big(g2, score, ws=100, max_iter=100 * windows)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"loop ({mode}): {res['windows']} windows in {dt*1e3:.0f} ms, fixed {res['fixed']} of {n}, live {res['live']:.0f}, objective {res['objective']:.1f}")
