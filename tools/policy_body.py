"""Time the fused policy ENCODER kernel alone (no head, no re-scoring): 128 000 variables x 20 tokens by default.
usage: python tools/policy_body.py [rows] [reps] [f16|f32]; LPBOX_LIB_VARIANT selects an experiment build (csrc/Makefile `variant`).
f32: the float32 encoder on v_mfma_f32_16x16x4_f32 (lpbox_policy_encode_f32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd'))
import torch
from lpbox_hip.policy import FusedEarlyFixPolicy, MfmaFp32Policy
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else "f16"
pol = (MfmaFp32Policy if kind == "f32" else FusedEarlyFixPolicy).random(tokens=20, seed=0)
torch.manual_seed(0)
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
if os.environ.get("POLICY_DIGEST"):
    import hashlib
    out = pol.encode(xf, off, 5)
    print("digest", hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest(), float(out.double().sum()))
flop = rows * 20 * 2 * (5 * 128 + 2 * (128 * 384 + 128 * 128 + 2 * 128 * 512) + 2 * 8 * 20 * 16 * 2)
print("encoder %s %s rows %d: %.3f ms (best of %d), %.0f TFLOP/s" % (kind, os.environ.get("LPBOX_LIB_VARIANT", "default"), rows, best, reps, flop / best / 1e9))
