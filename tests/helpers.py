"""Shared helpers for the test-suite (oracle = checker, lpbox_hip = product)."""
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def lp_instances(name):
    return O.load_lp_batch(os.path.join(GOLDEN, name))


def make_oracle(I, order=O.ORDER_EIGEN, T=512, positions=None, npos=0, row_split=None, col_split=None, x_update="pcg", direct_rows=None):
    s = O.LpOracle(0, order=order, T=T, positions=positions, npos=npos, row_split=row_split, col_split=col_split, x_update=x_update,
                   direct_rows=direct_rows)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"))
    s.solve_init()
    return s


def scripted_fix_vec(xiters, lo=0.02, hi=0.98, last=20):
    """Deterministic stand-in for the trained policy (LP/trainer.py:101-135,216-252): fix a live variable to 1 (0) when
    its last `last` iterates are all > hi (< lo); everything else stays free (-1).  Returns (vec, num)."""
    tail = xiters[:, -last:]
    vec = -np.ones(xiters.shape[0])
    vec[np.all(tail > hi, axis=1)] = 1.0
    vec[np.all(tail < lo, axis=1)] = 0.0
    num = int(np.sum(vec != -1))
    return vec, num


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def oracle_like(g, I):
    """CPU oracle configured with the reduction tree of the HIP solver `g` (threads + storage positions)."""
    return oracle_for(g.batch if hasattr(g, "batch") else g, 0, I)


def oracle_for(batch, idx, I, x_update="pcg", direct_rows=None):
    cfg = batch.config()
    return make_oracle(I, O.ORDER_GPU, cfg["threads"], batch.layout(idx), cfg["threads"] * cfg["elems_per_thread"],
                       batch.row_split(idx), batch.col_split(idx), x_update=x_update, direct_rows=direct_rows)


def oracle_full_solve(args):
    """Worker (CPU): one instance solved to convergence by the oracle in the kernels' association."""
    I, T, npos, pos, rs, cs = args
    s = O.LpOracle(0, order=O.ORDER_GPU, T=T, positions=pos, npos=npos, row_split=rs, col_split=cs)
    s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    s.solve_init()
    ret = s.solve_iter(0, 20000)
    return ret, s.total_outer_iters, s.total_pcg_iters, s.cal_Obj(), s.vec("x"), s.get_x_sol().ravel()



# ------------------------------------------------------------------------------------------------
# The reference's own validation loops (LP/trainer.py:_valid_2, SEG/trainer.py:_my_valid) driven on the CPU oracle:
# adapters with the pyx surface that log every call, and a scripted stand-in for the trained network.
# Shared by tests/golden/make_trainer_fixtures.py (reference loop -> fixture) and tests/test_trainer_pins.py (our loop).
# ------------------------------------------------------------------------------------------------
class CallLog:
    """Per-solver call log, JSON-friendly: what the loop handed to the solver and what it read back."""

    def __init__(self):
        self.solvers = []

    def new_solver(self, tag):
        rec = dict(tag=tag, windows=[], final={})
        self.solvers.append(rec)
        return rec


def _vec_code(vec, n_live):
    return [int(v) for v in np.asarray(vec, np.float64).ravel()[:n_live]]


class LoggedLpSolver:
    """lpbox.PyLPboxADMMsolver (LP pyx:7-76) on oracle/lpbox_oracle.c; `instances` / `log` are set on the class by the caller."""
    instances = None
    log = None

    def __init__(self, print_info=0):
        self._o = O.LpOracle(int(print_info))
        self._rec = None

    def read_File(self, i, k, j):
        I = self.instances[int(i) - 1]                      # instance files are numbered from 1 (LP/trainer.py:502)
        self._o.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
        self._rec = self.log.new_solver("lp%d" % int(i))

    def solve_init(self):
        return self._o.solve_init()

    def solve_iter_l2f(self, i, j, vec, num):
        n_live = self._o.get_n()
        ret = self._o.solve_iter_l2f(i, j, np.ascontiguousarray(vec, np.float64), num)
        self._rec["windows"].append(dict(start=int(i), end=int(j), num=int(num), vec=_vec_code(vec, n_live) if num else [],
                                         ret=int(ret), n_live=int(self._o.get_n())))
        return ret

    def get_x_iters_2d(self, ws):
        X = self._o.get_x_iters_2d(ws)
        self._rec["windows"][-1]["xiters_shape"] = list(X.shape)
        self._rec["windows"][-1]["xiters_sum"] = float(X.sum())
        return X

    def check_infeasible_l2f(self):
        v = self._o.check_infeasible_l2f()
        self._rec["final"]["infeasible"] = int(v)
        return v

    def cal_Obj(self):
        v = self._o.cal_Obj()
        self._rec["final"]["cal_obj"] = float(v)
        return v

    def get_n(self):
        return self._o.get_n()


def synthetic_seg_problem(problem, rows=24, cols=20):
    """Small smooth-noise grayscale image -> (A, b, c) through the oracle's cost builder (SEGcpp:46-248)."""
    rs = np.random.RandomState(100 + int(problem))
    g = rs.rand(rows + 8, cols + 8)
    k = np.ones((5, 5)) / 25.0
    sm = np.zeros((rows, cols))
    for r in range(rows):
        for c in range(cols):
            sm[r, c] = (g[r:r + 5, c:c + 5] * k).sum()
    sm = (sm - sm.min()) / (sm.max() - sm.min())
    img = np.floor(255 * (0.25 * rs.rand(rows, cols) + 0.75 * sm)).astype(np.float64)
    return O.seg_build_costs(img)


class LoggedSegSolver:
    """lpbox.PyLPboxADMMsolver (SEG pyx:8-53) on oracle/seg_oracle.c; the image of `problem` is synthetic (the reference's
    numNodes = 1e4 is ignored: what is pinned is the loop, not the image pipeline)."""
    log = None

    def __init__(self, print_info, numNodes, problem):
        self._o = O.SegOracle(int(print_info), int(numNodes), int(problem))
        self._o.set_problem(synthetic_seg_problem(problem))
        self._rec = self.log.new_solver("seg%d" % int(problem))

    def solve_init(self):
        return self._o.solve_init()

    def solve_iter_l2f(self, i, j, vec, num):
        n_live = self._o.get_n()
        ret = self._o.solve_iter_l2f(i, j, np.ascontiguousarray(vec, np.float64), num)
        self._rec["windows"].append(dict(start=int(i), end=int(j), num=int(num), vec=_vec_code(vec, n_live) if num else [],
                                         ret=int(ret), n_live=int(self._o.get_n())))
        return ret

    def get_x_iters_2d(self, ws):
        X = self._o.get_x_iters_2d(ws)
        self._rec["windows"][-1]["xiters_shape"] = list(X.shape)
        self._rec["windows"][-1]["xiters_sum"] = float(X.sum())
        return X

    def get_obj(self):
        v = self._o.get_obj()
        self._rec["final"]["obj"] = float(v)
        return v

    def get_x_sol(self):
        x = self._o.get_x_sol()
        self._rec["final"]["x_sol_sum"] = float(x.sum())
        return x

    def get_n(self):
        return self._o.get_n()


def scripted_scores(x, hi=0.985, lo=0.015):
    """Deterministic stand-in for the trained network: x float32 (rows, tokens, width) -> sigmoid-like scores that sit far from
    the 0.9 / 0.1 thresholds: 0.95 where the mean of the LAST token's iterates exceeds `hi`, 0.05 below `lo`, else 0.5."""
    import torch
    x = torch.as_tensor(x, dtype=torch.float32)
    m = x[:, -1, :].mean(dim=1)
    return torch.where(m > hi, 0.95, torch.where(m < lo, 0.05, 0.5)).to(torch.float32)


def oracle_iters_fix(o, i, j, x_prev=None, consistency=5, fix_threshold=1e-3, min_fix=10):
    """ADMM_lp_iters_fix (LPcpp:1689-2286) restated in the reference's OWN order on the oracle's primitives: one iteration, the
    persistence counters (:1857-1871), the stop tests, then the fix AT THE END of the iteration (:1929-2043) -- applied here
    through a zero-length l2f call, i.e. the repaired fix block (see lpbox_hip.lp.PyLPboxADMMsolver.solve_iter_fix).
    Returns (ret, x_prev)."""
    n = o.get_n()
    prev = np.zeros(n) if x_prev is None or len(x_prev) != n else x_prev
    count, flag = np.zeros(n), np.zeros(n, bool)
    ret = 0
    for it in range(i, j):
        r = o.solve_iter_l2f(it, it + 1, np.zeros(n), 0)
        reason = o.last_stop_reason
        if reason in (3, 4) or (r and reason == 0):
            return 1, prev
        x = o.get_x_iters_2d(1)[:, 0]
        for k in range(n):                                   # the reference's loop, element by element
            if abs(x[k] - prev[k]) <= fix_threshold:
                count[k] += 1
                if count[k] >= consistency:
                    flag[k] = True
            else:
                count[k] = 0
        prev = x.copy()
        if reason == 1:
            break
        if reason == 2:
            ret = 1
            break
        fix_n = int(flag.sum())
        if fix_n <= min_fix:
            continue
        vec = np.where(flag, np.where(x >= 0.5, 1.0, 0.0), -1.0)
        r = o.solve_iter_l2f(it + 1, it + 1, vec, fix_n)     # the fix block alone
        keep = ~flag
        prev, count, flag = prev[keep], count[keep], flag[keep]
        n = int(keep.sum())
        if r:
            ret = 1
            break
    return ret, prev
