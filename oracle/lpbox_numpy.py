"""Second, independent CPU restatement of the reference's LP path in numpy/scipy.  TEST INFRASTRUCTURE ONLY.

Written from the reference source, not from oracle/lpbox_oracle.c, so that the two can pin each other
(the reference ships no golden vectors -- SURVEY.md section 8c):
  LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp
Reductions use numpy's own (pairwise) summation, i.e. a THIRD association besides Eigen's and the GPU tree;
agreement with the C oracle is therefore to rounding, not bitwise.
"""
import numpy as np
import scipy.sparse as sp


class NumpyLpBox:
    def __init__(self, n, l, colptr, rowidx, b, f=None, x_update="pcg"):
        # x_update "direct": the HIP kernels' opt-in exact x-update (no reference counterpart), here as a plain dense solve of
        # (dI I + rho4 E^T E) x = rhs -- none of the Woodbury algebra of the kernel / the C oracle's mirror, so it pins their math
        self.x_update = x_update
        data = np.ones(len(rowidx))
        self.E = sp.csc_matrix((data, rowidx, colptr), shape=(l, n))
        self.orgE = self.E.copy()
        self.b = np.array(b, float)
        self.f = np.ones(l) if f is None else np.array(f, float)

    # ADMM_lp_iters_init LPcpp:489-763
    def solve_init(self):
        self.stop_threshold = 1e-4
        self.gamma_val = 1.6
        self.gamma_factor = 0.95
        self.rho_change_step = 25
        self.learning_fact = 1 + 1.0 / 100
        self.pcg_tol = 1e-3
        self.pcg_maxiters = 1000
        self.std_threshold = 1e-12
        self.history_size = 10
        l, n = self.E.shape
        self.n, self.l, self.org_n = n, l, n
        self.x = np.ones(n)
        self.y1 = self.x.copy()
        self.y2 = self.x.copy()
        self.z1 = np.zeros(n)
        self.z2 = np.zeros(n)
        self.z4 = np.zeros(l)
        self.y3 = self.f - self.E @ self.x
        self.rho1 = self.rho2 = self.rho4 = 25.0
        self.prev_rho1 = self.prev_rho2 = self.prev_rho4 = 25.0
        self.rhoUpdated = True
        self.std_obj = 1.0
        self.cur_obj = 0.0
        self.obj_list = []
        self.best_bin_obj = self.b @ self.x
        self.left_idx = np.arange(n)
        self.fixed_idx = np.zeros(0, int)
        self.fixed_val = np.zeros(0)
        self.sum_fix_obj = 0.0
        self.iter = 0
        self.pcg_trace = []
        self.x_iters = None
        return 1

    # update_expression LPcpp:2289-2404
    def _update_expression(self):
        self.Et = self.E.T.tocsc()
        self.r4Et_scale = self.rho4          # every E value is 1 so rho4_E_transpose = scale * E^T
        self.dI = 0.0 + (self.rho1 + self.rho2)
        self.Esq = np.asarray(self.E.multiply(self.E).sum(axis=0)).ravel()
        self.pd = self.dI + self.rho4 * self.Esq

    def _matvec(self, v):  # calculate_mat_expr_multiplication LPcpp:115-162
        return self.dI * v + self.r4Et_scale * (self.Et @ (self.E @ v))

    def _pcg(self, rhs, x):  # LPcpp:251-335
        r = rhs - self._matvec(x)
        rhsNorm2 = rhs @ rhs
        if rhsNorm2 == 0:
            x[:] = 0
            return 1, 0
        thr = max(self.pcg_tol * self.pcg_tol * rhsNorm2, np.finfo(float).tiny)
        r2 = r @ r
        if r2 < thr:
            return 1, 0
        p = self.invdiag * r
        absNew = r @ p
        i = 0
        while i < self.pcg_maxiters:
            tmp = self._matvec(p)
            alpha = absNew / (p @ tmp)
            if alpha < 0:
                return -1, i
            x += alpha * p
            r -= alpha * tmp
            r2 = r @ r
            if r2 < thr:
                i += 1
                break
            z = self.invdiag * r
            absOld = absNew
            absNew = r @ z
            p = z + (absNew / absOld) * p
            i += 1
        return 1, i

    def _iteration(self, it, it_start, l2f):
        # y1, y2, y3  LPcpp:806-828
        self.y1 = np.clip(self.x + self.z1 / self.rho1, 0.0, 1.0)
        t = (self.x + self.z2 / self.rho2) - 0.5
        self.y2 = t * np.power(float(self.n), 0.5) / (2 * np.sqrt(t @ t)) + 0.5
        self.y3 = np.maximum(self.f - self.E @ self.x - self.z4 / self.rho4, 0.0)
        if it == 0:
            self._update_expression()
        if it != 0 and self.rhoUpdated:  # LPcpp:851-866
            rcr = self.learning_fact - 1.0
            self.dI += rcr * (self.prev_rho1 + self.prev_rho2)
            self.pd = self.pd + rcr * (self.prev_rho1 + self.prev_rho2)
            self.pd = self.pd + (rcr * self.prev_rho4) * self.Esq
            self.r4Et_scale = self.learning_fact * self.r4Et_scale
        rhs = (self.rho1 * self.y1 + self.rho2 * self.y2) - ((self.b + self.z1) + self.z2)  # LPcpp:872-878
        rhs = rhs + self.r4Et_scale * (self.Et @ (self.f - self.y3))
        rhs = rhs - self.Et @ self.z4
        if self.rhoUpdated:
            self.invdiag = np.where(self.pd != 0, 1.0 / self.pd, 1.0)
            self.rhoUpdated = False
        xt = self.y1.copy()
        if self.x_update == "direct":
            Ed = self.E.toarray()
            xt = np.linalg.solve(self.dI * np.eye(self.n) + self.r4Et_scale * (Ed.T @ Ed), rhs)
            cg, k = 1, 0
        else:
            cg, k = self._pcg(rhs, xt)
        self.pcg_trace.append(k)
        if l2f and cg == -1:
            return 2
        self.x = xt
        if l2f:
            self.x_iters[:, self.cc] = self.x
            self.cc += 1
        g = self.gamma_val
        self.z1 = self.z1 + (g * self.rho1) * (self.x - self.y1)
        self.z2 = self.z2 + (g * self.rho2) * (self.x - self.y2)
        upd = (g * self.rho4) * ((self.E @ self.x + self.y3) - self.f)
        self.z4 = upd if (not l2f and it == it_start) else self.z4 + upd
        t0 = max(np.sqrt(self.x @ self.x), 2.2204e-16)
        d1 = self.x - self.y1
        d2 = self.x - self.y2
        c1 = np.sqrt(d1 @ d1) / t0
        c2 = np.sqrt(d2 @ d2) / t0
        if c1 <= self.stop_threshold and c2 <= self.stop_threshold and (l2f or it != it_start):
            self.stop = "y1_y2"
            if l2f:
                self.ret = 1
            return 1
        if (it + 1) % self.rho_change_step == 0:  # LPcpp:951-970
            self.prev_rho1, self.prev_rho2, self.prev_rho4 = self.rho1, self.rho2, self.rho4
            self.rho1 *= self.learning_fact
            self.rho2 *= self.learning_fact
            self.rho4 *= self.learning_fact
            self.gamma_val = max(self.gamma_val * self.gamma_factor, 1.0)
            self.rhoUpdated = True
        self.obj_list.append(self.b @ self.x)
        if len(self.obj_list) >= self.history_size:
            h = np.array(self.obj_list[-self.history_size:])
            mean = h.sum() / len(h)
            var = ((h - mean) ** 2).sum() / (len(h) - 1)
            self.std_obj = (0.0 if var == 0 else np.sqrt(var)) / abs(h[-1])
        if self.std_obj <= self.std_threshold:
            self.ret = 1
            self.stop = "obj_std"
            return 1
        self.cur_obj = self.b @ (self.x >= 0.5).astype(float)
        if self.best_bin_obj >= self.cur_obj:
            self.best_bin_obj = self.cur_obj
        return 0

    def solve_iter(self, it_start, it_end):  # ADMM_lp_iters LPcpp:766-1095
        self.ret = 0
        self.stop = None
        it = it_start
        while it < it_end:
            if self._iteration(it, it_start, False):
                break
            it += 1
        self.last_plain_iter_plus1 = it + 1
        return self.ret

    def solve_iter_l2f(self, it_start, it_end, vec, fix_num):  # ADMM_lp_iters_l2f LPcpp:1098-1574
        self.ret = 0
        self.stop = None
        n = self.n
        self.x_iters = np.zeros((n - fix_num, 500))
        self.cc = 0
        if fix_num != 0:
            vec = np.asarray(vec)[:n]
            fixed = (vec == 1) | (vec == 0)
            assert fixed.sum() == fix_num
            keep = ~fixed
            E1 = self.E[:, keep]
            E2 = self.E[:, fixed]
            x2 = vec[fixed].astype(float)
            self.fixed_idx = np.concatenate([self.fixed_idx, self.left_idx[fixed]])
            self.fixed_val = np.concatenate([self.fixed_val, x2])
            self.left_idx = self.left_idx[keep]
            if n - fix_num == 0:
                self.ret = 1
                self.n = 0
                it_end = it_start
            else:
                self.x = self.x[keep]
                if np.sqrt(self.x @ self.x) < 1e-3:
                    self.ret = 1
                self.y1, self.y2, self.z1, self.z2 = self.y1[keep], self.y2[keep], self.z1[keep], self.z2[keep]
                b2 = self.b[fixed]
                self.b = self.b[keep]
                self.sum_fix_obj += b2 @ x2
                self.f = self.f - E2 @ x2
                self.n = n - fix_num
                self.E = sp.csc_matrix(E1)
                self._update_expression()
        self.iter = it_start
        while self.iter < it_end:
            rc = self._iteration(self.iter, it_start, True)
            if rc == 2:
                return 1
            if rc:
                break
            self.iter += 1
        return self.ret

    def get_x_sol(self):
        out = np.zeros(self.org_n)
        out[self.fixed_idx] = self.fixed_val
        if self.n != 0:
            out[self.left_idx] = (self.x >= 0.5).astype(float)
        return out

    def cal_obj(self):
        return self.sum_fix_obj + self.cur_obj if self.n != 0 else self.sum_fix_obj
