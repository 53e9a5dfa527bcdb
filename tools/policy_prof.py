"""Time the policy forward alone (rows = 128000 = every variable of the 256-instance batch)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd'))
import torch
from lpbox_hip.policy import EarlyFixPolicy
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128000
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fused": None, "hip32": None, "mfma32": None}[mode]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
from lpbox_hip.policy import FusedEarlyFixPolicy, HipFp32Policy, MfmaFp32Policy
pol = (FusedEarlyFixPolicy.random(tokens=20, seed=0) if mode == "fused" else HipFp32Policy.random(tokens=20, seed=0) if mode == "hip32"
       else MfmaFp32Policy.random(tokens=20, seed=0) if mode == "mfma32" else EarlyFixPolicy.random(tokens=20, seed=0, device="cuda", dtype=dt))
x = torch.rand(rows, 20, 5, device="cuda")
if mode in ("fused", "hip32", "mfma32"):
    xf = x.to(torch.float64).reshape(-1); off = torch.arange(rows, device="cuda") * 100
    _call = pol.__call__; pol_call = lambda _x: pol.scores_from_xiters(xf, off)
else:
    pol_call = pol
for _ in range(2): pol_call(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): s = pol_call(x)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
flop = rows * (20 * 2 * (5 * 128 + 2 * (128 * 384 + 128 * 128 + 2 * 128 * 512) + 2 * 8 * 20 * 16 * 2) + 2 * (2560 * 256 + 256 * 128 + 128 * 16 + 16))
print("rows %d %s: %.2f ms per forward, %.1f TFLOP/s" % (rows, sys.argv[2] if len(sys.argv) > 2 else "fp32", ms, flop / ms / 1e9))
