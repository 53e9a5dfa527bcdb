"""Generic constrained binary QP on the GPU:  min x'Ax + b'x  s.t.  Cx = d,  Ex <= f,  x in {0,1}^n.

The reference's `ADMM_bqp` (Segmentation/.../LPboxADMMsolver.cpp:1384-1832) with its four entry points `ADMM_bqp_unconstrained`,
`_linear_eq`, `_linear_ineq`, `_linear_eq_and_uneq` (:1834-2109) -- C++ only in the reference (nothing in its pyx reaches them);
here through the C-ABI `lpbox_bqp_*`.  Matrices: (rowptr, colidx, vals) CSR with ascending columns, or scipy.sparse matrices.
A must store every diagonal entry (explicit zeros where needed): the reference adds rho to `A.diagonal()` in place (:1483).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

PRESET_UNCONSTRAINED, PRESET_EQ, PRESET_INEQ, PRESET_EQ_INEQ = 0, 1, 2, 3
PARAM_NAMES = ("stop_threshold", "std_threshold", "gamma_val", "gamma_factor", "rho_change_step", "max_iters", "initial_rho",
               "history_size", "learning_fact", "pcg_tol", "pcg_maxiters")


def _csr(M, rows):
    if M is None:
        return None
    if hasattr(M, "tocsr"):                 # scipy.sparse
        M = M.tocsr().sorted_indices()
        M = (M.indptr, M.indices, M.data)
    p, i, v = (np.ascontiguousarray(M[0], np.int32), np.ascontiguousarray(M[1], np.int32), np.ascontiguousarray(M[2], np.float64))
    if p.shape[0] != rows + 1:
        raise ValueError("row pointer has %d entries, expected %d" % (p.shape[0], rows + 1))
    return p, i, v


def with_diagonal(rowptr, colidx, vals, n):
    """CSR -> CSR with an explicit (possibly zero) diagonal entry in every row, columns ascending."""
    rp, ci, va = [0], [], []
    for i in range(n):
        cols = list(colidx[rowptr[i]:rowptr[i + 1]])
        vs = list(vals[rowptr[i]:rowptr[i + 1]])
        if i not in cols:
            cols.append(i); vs.append(0.0)
        order = np.argsort(cols, kind="stable")
        ci += [cols[k] for k in order]; va += [vs[k] for k in order]
        rp.append(len(ci))
    return np.array(rp, np.int32), np.array(ci, np.int32), np.array(va, np.float64)


class BqpSolver:
    def __init__(self, n, A, b, x0, C_=None, d=None, E=None, f=None, preset=None, params=None, device=0):
        self._L = _lib.load()
        self.n = int(n)
        self.m = 0 if C_ is None else len(d)
        self.l = 0 if E is None else len(f)
        h = self._L.lpbox_bqp_create(int(device))
        if not h:
            check(-2, "lpbox_bqp_create")
        self._h = C.c_void_p(h)
        keep = []

        def ptrs(M):
            if M is None:
                return [None, None, None]
            keep.extend(M)
            return [a.ctypes.data_as(C.c_void_p) for a in M]

        def vec(v):
            if v is None:
                return None
            a = np.ascontiguousarray(v, np.float64)
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        A_ = _csr(A, self.n)
        if any(i not in A_[1][A_[0][i]:A_[0][i + 1]] for i in range(self.n)):
            A_ = with_diagonal(A_[0], A_[1], A_[2], self.n)
        args = [self._h, self.n] + ptrs(A_) + [vec(b), vec(x0), self.m] + ptrs(_csr(C_, self.m)) + [vec(d), self.l] + \
            ptrs(_csr(E, self.l)) + [vec(f)]
        check(self._L.lpbox_bqp_set_problem(*args), "lpbox_bqp_set_problem")
        ptype = (1 if self.m else 0) | (2 if self.l else 0)
        check(self._L.lpbox_bqp_preset(self._h, ptype if preset is None else int(preset)), "lpbox_bqp_preset")
        if params is not None:
            if isinstance(params, dict):
                raise TypeError("params: the 11 values in the order of bqp.PARAM_NAMES")
            check(self._L.lpbox_bqp_set_params(self._h, np.ascontiguousarray(params, np.float64)), "lpbox_bqp_set_params")

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpbox_bqp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self):
        it = C.c_int()
        check(self._L.lpbox_bqp_solve(self._h, C.byref(it)), "lpbox_bqp_solve")
        return it.value

    def vec(self, name):
        out = np.zeros(max(self.n, self.m, self.l, 1))
        k = check(self._L.lpbox_bqp_get_vec(self._h, name.encode(), out, len(out)), "lpbox_bqp_get_vec")
        return out[:k].copy()

    def scalar(self, name):
        v = C.c_double()
        check(self._L.lpbox_bqp_get_scalar(self._h, name.encode(), C.byref(v)), "lpbox_bqp_get_scalar")
        return v.value

    def solution(self):
        """The reference's Solution struct (LPh): x_sol, y1, y2, best_sol; the binary answer is x_sol >= 0.5."""
        return dict(x_sol=self.vec("x"), y1=self.vec("y1"), y2=self.vec("y2"), best_sol=self.vec("best_sol"))


# ---- the reference's four entry points by name (SEGcpp:1834-2109); each returns the Solution dict -------------------------------
def _run(n, A, b, x0, C_=None, d=None, E=None, f=None, preset=0, params=None, device=0):
    s = BqpSolver(n, A, b, x0, C_, d, E, f, preset=preset, params=params, device=device)
    it = s.solve()
    sol = s.solution()
    sol.update(iterations=it, stop=int(s.scalar("stop")), best_bin_obj=s.scalar("best_bin_obj"), time_elapsed_ms=s.scalar("kernel_ms"))
    s.close()
    return sol


def ADMM_bqp_unconstrained(n, A, b, x0, **kw):
    return _run(n, A, b, x0, preset=PRESET_UNCONSTRAINED, **kw)


def ADMM_bqp_linear_eq(n, A, b, x0, m, C_, d, **kw):
    return _run(n, A, b, x0, C_=C_, d=np.asarray(d)[:m], preset=PRESET_EQ, **kw)


def ADMM_bqp_linear_ineq(n, A, b, x0, l, E, f, **kw):
    return _run(n, A, b, x0, E=E, f=np.asarray(f)[:l], preset=PRESET_INEQ, **kw)


def ADMM_bqp_linear_eq_and_uneq(n, A, b, x0, m, C_, d, l, E, f, **kw):
    return _run(n, A, b, x0, C_=C_, d=np.asarray(d)[:m], E=E, f=np.asarray(f)[:l], preset=PRESET_EQ_INEQ, **kw)
