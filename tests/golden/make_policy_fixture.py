"""Golden vectors for lpbox_hip/policy.py, produced by the reference's own network.

Run in the build container only (it imports /root/reference/LinerProgramming/LinearProgramming/mha.py and the SEG twin):
    python tests/golden/make_policy_fixture.py
writes tests/golden/policy_reference.npz = inputs + the reference module's (logit, sigmoid) in eval mode, for weights that
`deterministic_state` below defines by formula (so no weight file has to be stored, and no checkpoint exists to use).
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def deterministic_state(shapes):
    """name -> tensor, a fixed formula of (position in the sorted name list, element index)."""
    sd = {}
    for k, name in enumerate(sorted(shapes)):
        shape = shapes[name]
        n = int(np.prod(shape)) if len(shape) else 1
        t = torch.arange(n, dtype=torch.float64)
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(7, dtype=torch.long)
            continue
        if name.endswith("running_var"):
            v = 0.6 + 0.5 * torch.cos(0.11 * t + k) ** 2
        elif name.endswith("normalizer.weight"):
            v = 0.8 + 0.3 * torch.sin(0.23 * t + k)
        else:
            fan = shape[-1] if len(shape) > 1 else 16
            v = torch.sin(0.37 * t + 1.3 * k) * (1.7 / math.sqrt(fan))
        sd[name] = v.reshape(shape).to(torch.float32)
    return sd


def deterministic_input(rows, tokens, seed):
    rs = np.random.RandomState(seed)
    x = rs.rand(rows, tokens, 5)
    x[: rows // 3] = np.round(x[: rows // 3])          # many iterates sit at 0 / 1
    return x.astype(np.float32)


def main():
    out = {}
    for tag, pkg_root, pkg, tokens in (("lp", "/root/reference/LinerProgramming", "LinearProgramming", 20),
                                       ("seg", "/root/reference/Segmentation", "Segmentation", 5)):
        sys.path.insert(0, pkg_root)
        mha = __import__(pkg + ".mha", fromlist=["GraphAttentionEncoder"])
        net = mha.GraphAttentionEncoder()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(deterministic_state(shapes))
        net.eval()
        x = deterministic_input(96, tokens, 5 + tokens)
        with torch.no_grad():
            logit, sig = net(torch.from_numpy(x))
        out[tag + "_x"] = x
        out[tag + "_logit"] = logit.numpy().ravel()
        out[tag + "_sigmoid"] = sig.numpy().ravel()
        out[tag + "_names"] = np.array(sorted(shapes))
        out[tag + "_shapes"] = np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)])
        sys.path.pop(0)
    np.savez_compressed(os.path.join(HERE, "policy_reference.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
