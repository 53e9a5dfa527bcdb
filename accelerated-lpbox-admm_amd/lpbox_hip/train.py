"""Training the early-fixing policy on MI355X: the recipe of LP/trainer.py `_train_mha_100`
(:254-299) with the solver of this repository producing the data.

Reference recipe, per training instance: the iterate history of a PLAIN solve (what `print_info = 2` dumps, LPcpp:903-909, read by
trainer.py:32-48), labels = the final iterate rounded (getLabel :80-89), inputs = the first 10 windows of `ws` iterates reshaped to
(n, 20, ws/20) (getSubset :91-98, :283-285), window i weighted 1/i, `BCEWithLogitsLoss(weight)`, Adam(1e-4), one optimiser step per
instance.  Here the histories never leave the GPU: `LpBatch.set_record` keeps the first windows of a whole batch of instances in HBM
and `x_iters_torch` hands them to torch zero-copy.

`TrainablePolicy` is the network in trainable form.  Its `state_dict()` has exactly the reference's keys and shapes
(`GraphAttentionEncoder`, LP/mha.py:202-249), so checkpoints move freely between the reference, this class and the inference
classes of lpbox_hip.policy (tests pin its eval-mode output to golden vectors from the reference module).
"""
import math

import numpy as np
import torch
from torch import nn

from .policy import CODE_DIM, EMBED, FF_HIDDEN, N_HEADS, N_LAYERS, position_code


class _Residual(nn.Module):                     # key prefix `.module` (LP/mha.py:10-17)
    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return x + self.module(x)


class _Attention(nn.Module):                    # parameters W_query / W_key / W_val / W_out per head (LP/mha.py:20-56)
    def __init__(self):
        super().__init__()
        hd = EMBED // N_HEADS
        self.W_query = nn.Parameter(torch.empty(N_HEADS, EMBED, hd))
        self.W_key = nn.Parameter(torch.empty(N_HEADS, EMBED, hd))
        self.W_val = nn.Parameter(torch.empty(N_HEADS, EMBED, hd))
        self.W_out = nn.Parameter(torch.empty(N_HEADS, hd, EMBED))
        for p in self.parameters():
            bound = 1.0 / math.sqrt(p.size(-1))
            nn.init.uniform_(p, -bound, bound)

    def forward(self, h):                       # (rows, tokens, EMBED)
        q = torch.einsum("rte,hed->rhtd", h, self.W_query)
        k = torch.einsum("rte,hed->rhtd", h, self.W_key)
        v = torch.einsum("rte,hed->rhtd", h, self.W_val)
        a = nn.functional.scaled_dot_product_attention(q, k, v)          # softmax(q k^T / sqrt(16)) v
        return torch.einsum("rhtd,hde->rte", a, self.W_out)


class _Norm(nn.Module):                         # key prefix `.normalizer` (LP/mha.py:125-154): BatchNorm1d over (rows*tokens)
    def __init__(self):
        super().__init__()
        self.normalizer = nn.BatchNorm1d(EMBED, affine=True)

    def forward(self, x):
        return self.normalizer(x.reshape(-1, x.size(-1))).view_as(x)


class _Head(nn.Module):                         # LP/mha.py:185-199
    def __init__(self, tokens):
        super().__init__()
        self.fc1 = nn.Linear(tokens * EMBED, 256)
        self.fc2 = nn.Linear(256, 128)
        self.fc3 = nn.Linear(128, 16)
        self.fc4 = nn.Linear(16, 1)

    def forward(self, z):
        z = torch.relu(self.fc1(z))
        z = torch.relu(self.fc2(z))
        z = torch.relu(self.fc3(z))
        return self.fc4(z)


class TrainablePolicy(nn.Module):
    def __init__(self, tokens=20):
        super().__init__()
        self.tokens = tokens
        self.init_embed = nn.Linear(2 * CODE_DIM, EMBED)
        self.layers = nn.Sequential(*[
            nn.Sequential(_Residual(_Attention()), _Norm(),
                          _Residual(nn.Sequential(nn.Linear(EMBED, FF_HIDDEN), nn.ReLU(), nn.Linear(FF_HIDDEN, EMBED))), _Norm())
            for _ in range(N_LAYERS)])
        self.classify = _Head(tokens)
        self.register_buffer("_code", position_code(tokens), persistent=False)

    def forward(self, x):
        """x: (rows, tokens, 5) float32 -> (logit, sigmoid), each (rows, 1), like the reference's forward (LP/mha.py:224-249)."""
        rows = x.shape[0]
        h = self.init_embed(torch.cat([x, self._code.expand(rows, -1, -1)], dim=-1))
        h = self.layers(h)
        logit = self.classify(h.reshape(rows, -1))
        return logit, torch.sigmoid(logit)


def window_inputs(history, ws=100, windows=10, tokens=20):
    """history: (n, >= windows*ws) iterates of one instance -> inputs (windows*n, tokens, ws/tokens), weights (windows*n, 1)
    (LP/trainer.py:267-287: window i = iterates [(i-1) ws, i ws), weight 1/i)."""
    n = history.shape[0]
    xs = [history[:, i * ws:(i + 1) * ws].reshape(n, tokens, ws // tokens) for i in range(windows)]
    w = torch.cat([torch.full((n, 1), 1.0 / (i + 1), device=history.device) for i in range(windows)])
    return torch.cat(xs).to(torch.float32), w.to(torch.float32)


def collect_training_data(instances, ws=100, windows=10, max_iter=20000):
    """For a list of LP instances: (histories, labels) with histories[i] a CUDA tensor (n_i, windows*ws) of the first iterates of a
    plain solve and labels[i] (n_i, 1) = the converged iterate rounded (getLabel).  Two launches per call: the recorded windows, then
    the full solve."""
    from .lp import LpBatch
    b = LpBatch(instances)
    b.solve_init()
    b.set_record(True)
    b.solve_iter(0, ws * windows)
    flat, stride = b.x_iters_torch(ws * windows)
    hist = [flat[i * stride: i * stride + b.get_n(i) * ws * windows].view(b.get_n(i), ws * windows).clone() for i in range(b.B)]
    b.set_record(False)
    b.solve_init()                                                # labels: an uninterrupted solve from scratch (a resumed call would
    b.solve_iter(0, max_iter)                                     # overwrite z4 on its first iteration, LPcpp:920-923); deterministic,
                                                                  # so its first windows are bit-identical to the recorded ones
    labels = [torch.from_numpy(np.asarray(b.get_x_sol(i), np.float32).reshape(-1, 1)).cuda() for i in range(b.B)]
    objective = np.array([-b.cal_obj(i) for i in range(b.B)])
    b.close()
    return hist, labels, objective


def train(policy, hist, labels, epochs=10, lr=1e-4, ws=100, windows=10, log=None):
    """One Adam step per instance and epoch, as the reference does (LP/trainer.py:259-297).  Returns the mean loss per epoch."""
    opt = torch.optim.Adam(policy.parameters(), lr=lr)
    out = []
    for ep in range(epochs):
        policy.train()
        tot = 0.0
        for h, y in zip(hist, labels):
            x, w = window_inputs(h, ws, windows, policy.tokens)
            opt.zero_grad()
            logit, _ = policy(x)
            loss = nn.functional.binary_cross_entropy_with_logits(logit, y.repeat(windows, 1), weight=w)
            loss.backward()
            opt.step()
            tot += float(loss.item())
        out.append(tot / len(hist))
        if log:
            log("epoch %d: mean loss %.5f" % (ep, out[-1]))
    policy.eval()
    return out
