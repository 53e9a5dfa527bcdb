"""Time the fused policy ENCODER kernel alone (no head, no re-scoring): 128 000 variables x 20 tokens by default.
usage: python tools/policy_body.py [rows] [reps]; LPBOX_LIB_VARIANT selects an experiment build (csrc/Makefile `variant`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'))
import torch
from lpbox_hip.policy import FusedEarlyFixPolicy
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pol = FusedEarlyFixPolicy.random(tokens=20, seed=0)
xf = torch.rand(rows * 100, device="cuda", dtype=torch.float64)
off = torch.arange(rows, device="cuda") * 100
for _ in range(3):
    pol.encode(xf, off, 5)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pol.encode(xf, off, 5); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("encoder %s rows %d: %.3f ms (best of %d)" % (os.environ.get("LPBOX_LIB_VARIANT", "default"), rows, best, reps))
