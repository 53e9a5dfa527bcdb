"""The learned-early-fixing loop around the solver (the caller either side of the hot path):
windows of ADMM iterations alternate with a policy that scores every live variable from its recent iterates and fixes the
confident ones.  Mirrors LP/trainer.py `_valid_2` (:504-545) with `deter_fix_2` (:101-135); the policy itself (the
reference's GraphAttentionEncoder, LP/mha.py) is passed in as a callable and runs unmodified on PyTorch-ROCm.
"""
import numpy as np


def fix_vector_from_scores(sig, C=0.9):
    """deter_fix_2 (LP/trainer.py:101-135): sigmoid score > C -> fix to 1, < 1-C -> fix to 0, else leave (-1)."""
    sig = np.asarray(sig, np.float64).ravel()
    vec = -np.ones(sig.shape[0])
    vec[sig > C] = 1.0
    vec[sig < 1 - C] = 0.0
    return vec, int(np.sum(sig > C)), int(np.sum(sig < 1 - C))


def run_l2f(solver, score_fn, ws=100, max_iter=10000, col=None, tokens=20, min_fix=10):
    """One instance through the reference's validation loop (LP/trainer.py:504-545).

    solver: a PyLPboxADMMsolver after solve_init(); score_fn(x) maps a float32 array (n_live, tokens, ws/tokens) to
    per-variable sigmoid scores.  Returns dict(objective, infeasible, windows, fixed)."""
    n = 0
    vec = np.zeros(col if col is not None else solver.get_n(), dtype=np.double)      # ignored while n == 0 (LPcpp:1124)
    windows = fixed = 0
    for i in range(int(max_iter / ws)):
        ret = solver.solve_iter_l2f(ws * i, ws * (i + 1), vec, n)
        windows += 1
        fixed += n
        if ret:
            break
        xiters = solver.get_x_iters_2d(ws)
        a, b = xiters.shape
        vec, f1, f0 = fix_vector_from_scores(score_fn(xiters.reshape(a, tokens, int(b / tokens)).astype(np.float32)))
        n = f1 + f0
        if n <= min_fix:                                                              # LP/trainer.py:533-535
            n = 0
    return dict(objective=-1.0 * solver.cal_Obj(), infeasible=solver.check_infeasible_l2f(), windows=windows, fixed=fixed)


def run_l2f_batch(batch, score_fn_torch, ws=100, max_iter=10000, tokens=20, min_fix=10, C=0.9):
    """The same loop for a whole LpBatch with the policy reading the iterates ON THE DEVICE (no host round trip of x_iters):
    score_fn_torch(x) maps a float32 CUDA tensor (n_live, tokens, ws/tokens) to sigmoid scores (n_live,)."""
    import torch
    B = batch.B
    nmax = max(batch.get_org_n(i) for i in range(B))
    vecs = np.zeros((B, nmax))
    nums = np.zeros(B, np.int32)
    done = np.zeros(B, bool)
    for w in range(int(max_iter / ws)):
        batch.set_active(~done)
        rets = batch.solve_iter_l2f(ws * w, ws * (w + 1), vecs, nums)
        done |= rets != 0
        if done.all():
            break
        flat, stride = batch.x_iters_torch(ws)
        nums[:] = 0
        for i in range(B):
            if done[i]:
                continue
            rows = batch.get_n(i)
            x = flat[i * stride: i * stride + rows * ws].view(rows, tokens, ws // tokens).to(torch.float32)
            sig = score_fn_torch(x).to(torch.float64).reshape(-1)
            vec = torch.where(sig > C, 1.0, torch.where(sig < 1 - C, 0.0, -1.0))
            k = int((vec != -1).sum().item())
            if k > min_fix:
                vecs[i, :rows] = vec.cpu().numpy()
                nums[i] = k
    return dict(objective=np.array([-batch.cal_obj(i) for i in range(B)]),
                infeasible=np.array([batch.check_infeasible_l2f(i) for i in range(B)]), windows=w + 1)


def sliding_windows(xiters, tokens=5, width=5):
    """SEG/trainer.py:721-725: token j of a variable = its iterates j .. j+width-1."""
    a = xiters.shape[0]
    out = np.zeros((a, tokens, width))
    for j in range(tokens):
        out[:, j, :] = xiters[:, j:j + width]
    return out


def run_l2f_seg(solver, score_fn, ws=10, max_iter=30, min_fix=10):
    """The segmentation validation loop (SEG/trainer.py:699-745): solver is a SEG PyLPboxADMMsolver after solve_init();
    score_fn maps a float32 array (n_live, 5, 5) to sigmoid scores.  Returns dict(energy, windows, fixed)."""
    n = 0
    vec = np.zeros(solver.get_n(), dtype=np.double)
    windows = fixed = 0
    for i in range(int(max_iter / ws)):
        ret = solver.solve_iter_l2f(ws * i, ws * (i + 1), vec, n)
        windows += 1
        fixed += n
        if ret:
            break
        x = sliding_windows(solver.get_x_iters_2d(ws)).astype(np.float32)
        vec, f1, f0 = fix_vector_from_scores(score_fn(x))
        n = f1 + f0
        if n <= min_fix:                                                              # SEG/trainer.py:735-736
            n = 0
    return dict(energy=solver.get_obj(), windows=windows, fixed=fixed)
