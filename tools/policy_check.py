import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,os.path.join(ROOT,'tests','golden'))
import torch, numpy as np
from lpbox_hip import policy as P
from make_policy_fixture import deterministic_state
for tokens in (20, 5):
    for name, sd in (("det", deterministic_state(P.reference_state_shapes(tokens))), ("rand", P.random_state(tokens, 0))):
        ref = P.EarlyFixPolicy(sd, tokens=tokens, device="cuda")
        h16 = P.EarlyFixPolicy(sd, tokens=tokens, device="cuda", dtype=torch.float16)
        fu = P.FusedEarlyFixPolicy(sd, tokens=tokens, device="cuda")
        x = torch.rand(1003, tokens, 5, generator=torch.Generator().manual_seed(3)).cuda()
        lr, lh, lf = ref.logits(x), h16.logits(x), fu.logits(x)
        print("T=%d %s: logits ref mean %.4f std %.4f | torch-fp16 err max %.2e mean %+.2e | fused err max %.2e mean %+.2e | sigmoid err fused %.2e" % (
            tokens, name, lr.mean().item(), lr.std().item(), (lh - lr).abs().max().item(), (lh - lr).mean().item(),
            (lf - lr).abs().max().item(), (lf - lr).mean().item(), (torch.sigmoid(lf) - torch.sigmoid(lr)).abs().max().item()))
