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


def run_l2f_batch(batch, score_fn_torch, ws=100, max_iter=10000, tokens=20, min_fix=10, C=0.9, timing=None):
    """The same loop for a whole LpBatch with the policy reading the iterates ON THE DEVICE (no host round trip of x_iters):
    score_fn_torch(x) maps a float32 CUDA tensor (rows, tokens, ws/tokens) to sigmoid scores (rows,); it is called once per
    window on the live variables of ALL unfinished instances stacked.  `timing`, if a dict, receives seconds per phase."""
    import time

    import torch
    B = batch.B
    nmax = max(batch.get_org_n(i) for i in range(B))
    vecs = np.zeros((B, nmax))
    nums = np.zeros(B, np.int32)
    done = np.zeros(B, bool)
    t = dict(solve=0.0, policy=0.0, host=0.0)
    windows = 0
    for w in range(int(max_iter / ws)):
        t0 = time.perf_counter()
        batch.set_active(~done)
        rets = batch.solve_iter_l2f(ws * w, ws * (w + 1), vecs, nums)
        windows += 1
        done |= rets != 0
        t1 = time.perf_counter()
        t["solve"] += t1 - t0
        if done.all():
            break
        flat, stride = batch.x_iters_torch(ws)
        act = np.flatnonzero(~done)
        rows = [batch.get_n(int(i)) for i in act]
        if hasattr(score_fn_torch, "scores_from_xiters"):
            # fused policy (lpbox_hip.policy.FusedEarlyFixPolicy): reads the fp64 iterates in place, one offset per live variable
            r = np.asarray(rows, np.int64)
            first = np.repeat(np.cumsum(r) - r, r)
            off = np.repeat(act.astype(np.int64) * stride, r) + (np.arange(int(r.sum()), dtype=np.int64) - first) * ws
            sig = score_fn_torch.scores_from_xiters(flat, torch.from_numpy(off).to(flat.device), ws // tokens).reshape(-1)
        else:
            X = torch.cat([flat[i * stride: i * stride + r * ws].view(r, ws) for i, r in zip(act.tolist(), rows)])
            sig = score_fn_torch(X.view(-1, tokens, ws // tokens).to(torch.float32)).reshape(-1)
        vec = torch.where(sig > C, 1.0, torch.where(sig < 1 - C, 0.0, -1.0)).to(torch.float64).cpu().numpy()   # deter_fix_2
        t2 = time.perf_counter()
        t["policy"] += t2 - t1
        nums[:] = 0
        off = 0
        for i, r in zip(act.tolist(), rows):
            v = vec[off:off + r]
            off += r
            k = int(np.count_nonzero(v != -1))
            if k > min_fix:                                                           # LP/trainer.py:533-535
                vecs[i, :r] = v
                nums[i] = k
        t["host"] += time.perf_counter() - t2
    if timing is not None:
        timing.update(t)
    return dict(objective=np.array([-batch.cal_obj(i) for i in range(B)]),
                infeasible=np.array([batch.check_infeasible_l2f(i) for i in range(B)]), windows=windows)


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


def run_l2f_seg_device(solver, policy, ws=10, max_iter=30, min_fix=10, C=0.9):
    """run_l2f_seg with the iterates read on the device: `policy` is a lpbox_hip.policy.FusedEarlyFixPolicy(tokens=5); token j of a
    variable = iterates j .. j+4 of the window (SEG/trainer.py:721-725), i.e. token stride 1 over the packed (n_live x ws) buffer."""
    import torch
    n = 0
    vec = np.zeros(solver.get_n(), dtype=np.double)
    windows = fixed = 0
    for i in range(int(max_iter / ws)):
        ret = solver.solve_iter_l2f(ws * i, ws * (i + 1), vec, n)
        windows += 1
        fixed += n
        if ret:
            break
        X = solver.x_iters_torch(ws)
        off = torch.arange(X.shape[0], device=X.device, dtype=torch.int64) * ws
        sig = policy.scores_from_xiters(X.reshape(-1), off, 1)
        vec = torch.where(sig > C, 1.0, torch.where(sig < 1 - C, 0.0, -1.0)).to(torch.float64).cpu().numpy()
        n = int(np.count_nonzero(vec != -1))
        if n <= min_fix:
            n = 0
    return dict(energy=solver.get_obj(), windows=windows, fixed=fixed)


def run_l2f_big(big, score_fn_torch, ws=100, max_iter=10000, tokens=20, min_fix=10, C=0.9):
    """The loop on ONE large variable-sharded instance (BASELINE config 5; lpbox_hip.big.BigLp, one process per GPU): every rank
    scores ITS OWN live variables from its device-resident x_iters (no gather); the only extra collective is the sum of the
    per-rank fix counts, which the "<= 10 fixes => none" rule (LP/trainer.py:533-535) and the shrunken sphere radius need."""
    import torch
    vec, num, windows, fixed = None, 0, 0, 0
    for w in range(int(max_iter / ws)):
        ret = big.solve_iter_l2f(ws * w, ws * (w + 1), vec, num)
        windows += 1
        fixed += num
        if ret:
            break
        X = big.x_iters_torch(ws)                                       # (local live rows, ws) fp64 on the device
        if hasattr(score_fn_torch, "scores_from_xiters"):
            off = torch.arange(X.shape[0], device=X.device, dtype=torch.int64) * ws
            sig = score_fn_torch.scores_from_xiters(X.reshape(-1), off, ws // tokens).reshape(-1)
        elif X.shape[0]:
            sig = score_fn_torch(X.view(-1, tokens, ws // tokens).to(torch.float32)).reshape(-1)
        else:
            sig = torch.zeros(0, device=X.device)
        v = torch.where(sig > C, 1.0, torch.where(sig < 1 - C, 0.0, -1.0)).to(torch.float64).cpu().numpy()
        num = big.sum_over_ranks(int(np.count_nonzero(v != -1)))
        if num <= min_fix:
            vec, num = None, 0
        else:
            vec = v
    return dict(objective=-big.cal_Obj(), windows=windows, fixed=fixed, live=big.scalar("n_live"))
