"""ctypes binding of the CPU oracle (oracle/lpbox_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (accelerated-lpbox-admm_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("LPBOX_ORACLE_LIB") or os.path.join(HERE, "liblpbox_oracle.so")     # override: e.g. a sanitizer build

ORDER_EIGEN = 0
ORDER_GPU = 1

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    if os.environ.get("LPBOX_ORACLE_LIB"):
        return LIB
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()                     # (re)builds only when the library is missing or older than its sources
    L = C.CDLL(LIB)
    L.lpo_create.restype = C.c_void_p
    L.lpo_create.argtypes = [C.c_int]
    L.lpo_destroy.argtypes = [C.c_void_p]
    L.lpo_set_order.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.lpo_set_verbose.argtypes = [C.c_void_p, C.c_int]
    L.lpo_set_chunk.argtypes = [C.c_void_p, C.c_int]
    L.lpo_set_ranks.argtypes = [C.c_void_p, C.c_int]
    L.lpo_set_pcg_lean.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.lpo_set_x_update.argtypes = [C.c_void_p, C.c_int]
    L.lpo_set_direct_rows.argtypes = [C.c_void_p, _ip, C.c_int]
    L.lpo_set_log.argtypes = [C.c_void_p, C.c_char_p]
    L.lpo_set_positions.argtypes = [C.c_void_p, _ip, C.c_int, C.c_int]
    L.lpo_set_row_split.argtypes = [C.c_void_p, _ip, C.c_int]
    L.lpo_set_col_split.argtypes = [C.c_void_p, _ip, _ip, C.c_int]
    L.lpo_set_problem.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _ip, _ip, C.c_void_p, _dp, _dp]
    L.lpo_read_files.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    L.lpo_init.argtypes = [C.c_void_p]
    L.lpo_iters.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.lpo_iters_l2f.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int]
    for f in ("lpo_get_n", "lpo_get_org_n", "lpo_get_l", "lpo_get_iter", "lpo_get_x_iters_rows",
              "lpo_check_infeasible_lpbox", "lpo_check_infeasible_l2f", "lpo_last_plain_iter_plus1",
              "lpo_last_pcg_iters", "lpo_last_stop_reason"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int
    for f in ("lpo_total_pcg_iters", "lpo_total_outer_iters"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_long
    L.lpo_get_x_iters.argtypes = [C.c_void_p, C.c_int, _dp]
    L.lpo_get_x_sol.argtypes = [C.c_void_p, _dp]
    L.lpo_get_final_x_sol.argtypes = [C.c_void_p, _dp]
    L.lpo_cal_obj.argtypes = [C.c_void_p]
    L.lpo_cal_obj.restype = C.c_double
    L.lpo_cur_bin_obj.argtypes = [C.c_void_p]
    L.lpo_cur_bin_obj.restype = C.c_double
    L.lpo_get_vec.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
    L.lpo_get_scalar.argtypes = [C.c_void_p, C.c_char_p]
    L.lpo_get_scalar.restype = C.c_double
    L.lpo_get_pcg_trace.argtypes = [C.c_void_p, _ip, C.c_int]
    L.lpo_project_box.argtypes = [C.c_int, _dp, _dp]
    L.lpo_project_shifted_lp_ball.argtypes = [C.c_int, _dp, _dp]
    _lib = L
    return L


class LpOracle:
    """Mirror of the reference's PyLPboxADMMsolver (lpbox.pyx:7-76) on the CPU oracle."""

    def __init__(self, print_info=0, order=ORDER_EIGEN, T=512, verbose=False, positions=None, npos=0, row_split=None, chunk=0, col_split=None, ranks=0, x_update="pcg", direct_rows=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.lpo_create(int(print_info)))
        self.L.lpo_set_order(self.h, order, T)
        self.L.lpo_set_verbose(self.h, int(verbose))
        if chunk:
            self.L.lpo_set_chunk(self.h, int(chunk))
        if ranks:
            self.L.lpo_set_ranks(self.h, int(ranks))
        if x_update != "pcg":        # the kernels' opt-in direct x-update (no reference counterpart)
            assert x_update == "direct"
            self.L.lpo_set_x_update(self.h, 1)
            if direct_rows is not None:
                direct_rows = np.ascontiguousarray(direct_rows, np.int32)
                self.L.lpo_set_direct_rows(self.h, direct_rows, len(direct_rows))
        if positions is not None:
            positions = np.ascontiguousarray(positions, np.int32)
            self.L.lpo_set_positions(self.h, positions, len(positions), int(npos))
        if row_split is not None:
            row_split = np.ascontiguousarray(row_split, np.int32)
            self.L.lpo_set_row_split(self.h, row_split, len(row_split))
        if col_split is not None:
            own = np.ascontiguousarray(col_split[0], np.int32)
            help4 = np.ascontiguousarray(col_split[1], np.int32).ravel()
            self.L.lpo_set_col_split(self.h, own, help4, len(own))

    def set_pcg_lean(self, on=True, row_chunk=256):
        """Mirror of BigLp.set_pcg_mode("lean"): the kernels' opt-in comm-lean PCG (no reference counterpart)."""
        self.L.lpo_set_pcg_lean(self.h, 1 if on else 0, int(row_chunk))

    def set_x_update(self, mode, direct_rows=None):
        """Switch between the reference's PCG and the kernels' opt-in direct x-update between calls (lpbox_set_x_update)."""
        assert mode in ("pcg", "direct")
        self.L.lpo_set_x_update(self.h, 1 if mode == "direct" else 0)
        if direct_rows is not None:
            direct_rows = np.ascontiguousarray(direct_rows, np.int32)
            self.L.lpo_set_direct_rows(self.h, direct_rows, len(direct_rows))

    def __del__(self):
        try:
            if self.h:
                self.L.lpo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_log(self, path):
        """The reference's default per-iteration log file (does_log); None turns it off."""
        if self.L.lpo_set_log(self.h, os.fsencode(path) if path else None):
            raise OSError(f"cannot open {path}")

    # -- problem input --
    def set_problem(self, n, l, colptr, rowidx, b, f=None, vals=None):
        colptr = np.ascontiguousarray(colptr, np.int32)
        rowidx = np.ascontiguousarray(rowidx, np.int32)
        b = np.ascontiguousarray(b, np.float64)
        f = np.ones(l) if f is None else np.ascontiguousarray(f, np.float64)
        vp = None
        if vals is not None:
            vals = np.ascontiguousarray(vals, np.float64)
            vp = vals.ctypes.data_as(C.c_void_p)
        rc = self.L.lpo_set_problem(self.h, n, l, len(rowidx), colptr, rowidx, vp, b, f)
        if rc != 0:
            raise ValueError(f"lpo_set_problem failed: {rc}")

    def read_files(self, path_C, path_b, k=100):
        rc = self.L.lpo_read_files(self.h, path_C.encode(), path_b.encode(), k)
        if rc != 0:
            raise IOError(f"lpo_read_files failed: {rc}")

    # -- the pyx surface --
    def solve_init(self):
        rc = self.L.lpo_init(self.h)
        if rc < 0:
            raise RuntimeError(f"lpo_init failed: {rc}")
        return rc

    def solve_iter(self, i, j):
        rc = self.L.lpo_iters(self.h, int(i), int(j))
        if rc < 0:
            raise RuntimeError(f"lpo_iters failed: {rc}")
        return rc

    def solve_iter_l2f(self, i, j, vec, num):
        vec = np.ascontiguousarray(vec, np.float64)
        rc = self.L.lpo_iters_l2f(self.h, int(i), int(j), vec, int(num))
        if rc < 0:
            raise RuntimeError(f"lpo_iters_l2f failed: {rc}")
        return rc

    def get_n(self):
        return self.L.lpo_get_n(self.h)

    def get_org_n(self):
        return self.L.lpo_get_org_n(self.h)

    def get_iter(self):
        return self.L.lpo_get_iter(self.h)

    def get_x_iters_2d(self, ws):
        rows = self.L.lpo_get_x_iters_rows(self.h)
        out = np.zeros((rows, ws))
        if rows:
            self.L.lpo_get_x_iters(self.h, ws, out)
        return out

    def get_x_sol(self, n=None):
        out = np.zeros(self.get_org_n())
        self.L.lpo_get_x_sol(self.h, out)
        return out.reshape(-1, 1)

    def get_final_x_sol(self, n=None):
        out = np.zeros(self.get_org_n())
        k = self.L.lpo_get_final_x_sol(self.h, out)
        return out[:k].reshape(-1, 1)

    def cal_Obj(self):
        return self.L.lpo_cal_obj(self.h)

    def get_curBinObj(self):
        return self.L.lpo_cur_bin_obj(self.h)

    def check_infeasible_lpbox(self):
        return self.L.lpo_check_infeasible_lpbox(self.h)

    def check_infeasible_l2f(self):
        return self.L.lpo_check_infeasible_l2f(self.h)

    # -- inspection --
    def vec(self, name):
        cap = max(self.get_org_n(), self.L.lpo_get_l(self.h)) + 8
        out = np.zeros(cap)
        k = self.L.lpo_get_vec(self.h, name.encode(), out, cap)
        if k < 0:
            raise KeyError(name)
        return out[:k].copy()

    def scalar(self, name):
        return self.L.lpo_get_scalar(self.h, name.encode())

    def pcg_trace(self):
        out = np.zeros(32768, np.int32)
        k = self.L.lpo_get_pcg_trace(self.h, out, len(out))
        return out[:k].copy()

    @property
    def total_pcg_iters(self):
        return self.L.lpo_total_pcg_iters(self.h)

    @property
    def total_outer_iters(self):
        return self.L.lpo_total_outer_iters(self.h)

    @property
    def last_stop_reason(self):
        return self.L.lpo_last_stop_reason(self.h)

    @property
    def last_plain_iter_plus1(self):
        return self.L.lpo_last_plain_iter_plus1(self.h)


def load_lp_batch(path):
    """Split a tests/golden/lp_*.npz fixture into per-instance dicts (n, l, colptr, rowidx, b=-price)."""
    d = np.load(path)
    out = []
    cp = ri = pr = 0
    for n, l, nnz in zip(d["n"], d["l"], d["nnz"]):
        n, l, nnz = int(n), int(l), int(nnz)
        out.append(dict(n=n, l=l, colptr=d["colptr"][cp:cp + n + 1].astype(np.int32),
                        rowidx=d["rowidx"][ri:ri + nnz].astype(np.int32),
                        b=-1.0 * d["price"][pr:pr + n]))
        cp += n + 1; ri += nnz; pr += n
    return out


def project_box(x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x)
    lib().lpo_project_box(len(x), x, y)
    return y


def project_shifted_lp_ball(x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x)
    lib().lpo_project_shifted_lp_ball(len(x), x, y)
    return y


# ------------------------------------------------------------------------------------------------
# segmentation flavour (oracle/seg_oracle.c)
# ------------------------------------------------------------------------------------------------
_seg_bound = False


def _seg_lib():
    global _seg_bound
    L = lib()
    if _seg_bound:
        return L
    pi, pd_ = C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.sego_create.restype = C.c_void_p
    L.sego_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.sego_destroy.argtypes = [C.c_void_p]
    L.sego_set_order.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.sego_set_verbose.argtypes = [C.c_void_p, C.c_int]
    L.sego_build_costs.argtypes = [C.c_int, C.c_int, _dp, pi, C.POINTER(pi), C.POINTER(pi), C.POINTER(pd_), C.POINTER(pd_), pd_]
    L.sego_free_arrays.argtypes = [pi, pi, pd_, pd_]
    L.sego_resize_linear_u8.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, pi, pi, C.c_void_p]
    L.sego_set_problem.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp, _dp, C.c_double, C.c_int, C.c_int]
    L.sego_init.argtypes = [C.c_void_p]
    L.sego_legacy.argtypes = [C.c_void_p]
    L.sego_l2f.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int]
    for f in ("sego_get_n", "sego_get_org_n", "sego_get_x_iters_rows", "sego_last_stop", "sego_legacy_iter_plus1"):
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("sego_total_pcg", "sego_total_outer"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_long
    L.sego_get_x_iters.argtypes = [C.c_void_p, C.c_int, _dp]
    L.sego_get_x_sol.argtypes = [C.c_void_p, _dp]
    L.sego_get_final_obj.argtypes = [C.c_void_p]
    L.sego_get_final_obj.restype = C.c_double
    L.sego_get_c.argtypes = [C.c_void_p]
    L.sego_get_c.restype = C.c_double
    L.sego_get_trace.argtypes = [C.c_void_p, _ip, C.c_int]
    L.sego_get_vec.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
    L.sego_get_scalar.argtypes = [C.c_void_p, C.c_char_p]
    L.sego_get_scalar.restype = C.c_double
    _seg_bound = True
    return L


def seg_build_costs(img):
    """SEGcpp:46-248: grayscale image (rows x cols, 0..255) -> dict(n, rowptr, colidx, vals, b, c, rows, cols)."""
    L = _seg_lib()
    img = np.ascontiguousarray(img, np.float64)
    rows, cols = img.shape
    n = C.c_int()
    rp, ci = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    va, bb = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    c = C.c_double()
    nnz = L.sego_build_costs(rows, cols, img, C.byref(n), C.byref(rp), C.byref(ci), C.byref(va), C.byref(bb), C.byref(c))
    out = dict(n=n.value, rows=rows, cols=cols, c=c.value,
               rowptr=np.ctypeslib.as_array(rp, (n.value + 1,)).copy(), colidx=np.ctypeslib.as_array(ci, (nnz,)).copy(),
               vals=np.ctypeslib.as_array(va, (nnz,)).copy(), b=np.ctypeslib.as_array(bb, (n.value,)).copy())
    L.sego_free_arrays(rp, ci, va, bb)
    return out


def seg_resize_u8(img_u8, scale):
    L = _seg_lib()
    img_u8 = np.ascontiguousarray(img_u8, np.uint8)
    r, c = C.c_int(), C.c_int()
    L.sego_resize_linear_u8(img_u8.shape[0], img_u8.shape[1], img_u8.ctypes.data_as(C.c_void_p), float(scale), C.byref(r), C.byref(c), None)
    out = np.zeros((r.value, c.value), np.uint8)
    L.sego_resize_linear_u8(img_u8.shape[0], img_u8.shape[1], img_u8.ctypes.data_as(C.c_void_p), float(scale), C.byref(r), C.byref(c),
                            out.ctypes.data_as(C.c_void_p))
    return out


class SegOracle:
    """Mirror of Segmentation/Segmentation/cython/src/lpbox.pyx:8-53 on the CPU oracle."""

    def __init__(self, print_info=0, numNodes=10000, problem=0, order=ORDER_EIGEN, T=256, chunk=512):
        self.L = _seg_lib()
        self.h = C.c_void_p(self.L.sego_create(int(print_info), int(numNodes), int(problem)))
        self.L.sego_set_order(self.h, order, T, chunk)

    def __del__(self):
        try:
            if self.h:
                self.L.sego_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_problem(self, P):
        rc = self.L.sego_set_problem(self.h, P["n"], len(P["colidx"]), np.ascontiguousarray(P["rowptr"], np.int32),
                                     np.ascontiguousarray(P["colidx"], np.int32), np.ascontiguousarray(P["vals"], np.float64),
                                     np.ascontiguousarray(P["b"], np.float64), float(P["c"]), P["rows"], P["cols"])
        if rc:
            raise ValueError(rc)

    def solve_init(self):
        return self.L.sego_init(self.h)

    def solve_iter(self):
        return self.L.sego_legacy(self.h)

    def solve_iter_l2f(self, i, j, vec, num):
        rc = self.L.sego_l2f(self.h, int(i), int(j), np.ascontiguousarray(vec, np.float64), int(num))
        if rc < 0:
            raise RuntimeError(rc)
        return rc

    def get_n(self):
        return self.L.sego_get_n(self.h)

    def get_org_n(self):
        return self.L.sego_get_org_n(self.h)

    def get_x_iters_2d(self, ws):
        rows = self.L.sego_get_x_iters_rows(self.h)
        out = np.zeros((rows, ws))
        if rows:
            self.L.sego_get_x_iters(self.h, ws, out)
        return out

    def get_obj(self):
        return self.L.sego_get_final_obj(self.h)

    def get_x_sol(self):
        out = np.zeros(self.get_org_n())
        self.L.sego_get_x_sol(self.h, out)
        return out.reshape(-1, 1)

    def vec(self, name):
        cap = self.get_org_n() + 8
        out = np.zeros(cap)
        k = self.L.sego_get_vec(self.h, name.encode(), out, cap)
        if k < 0:
            raise KeyError(name)
        return out[:k].copy()

    def scalar(self, name):
        return self.L.sego_get_scalar(self.h, name.encode())

    def pcg_trace(self):
        out = np.zeros(20000, np.int32)
        k = self.L.sego_get_trace(self.h, out, len(out))
        return out[:k].copy()

    @property
    def total_pcg_iters(self):
        return self.L.sego_total_pcg(self.h)

    @property
    def total_outer_iters(self):
        return self.L.sego_total_outer(self.h)

    @property
    def last_stop(self):
        return self.L.sego_last_stop(self.h)

    @property
    def legacy_iter_plus1(self):
        return self.L.sego_legacy_iter_plus1(self.h)


# ------------------------------------------------------------------------------------------------
# generic constrained BQP (oracle/bqp_oracle.c: ADMM_bqp, SEGcpp:1384-1832)
# ------------------------------------------------------------------------------------------------
_bqp_bound = False


def _bqp_lib():
    global _bqp_bound
    L = lib()
    if _bqp_bound:
        return L
    vp = C.c_void_p
    L.bqpo_create.restype = vp
    L.bqpo_destroy.argtypes = [vp]
    L.bqpo_set_order.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.bqpo_preset.argtypes = [vp, C.c_int]
    L.bqpo_set_params.argtypes = [vp, _dp]
    L.bqpo_set_problem.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    L.bqpo_solve.argtypes = [vp]
    L.bqpo_get_vec.argtypes = [vp, C.c_char_p, _dp, C.c_int]
    L.bqpo_get_scalar.argtypes = [vp, C.c_char_p]
    L.bqpo_get_scalar.restype = C.c_double
    L.bqpo_get_trace.argtypes = [vp, _ip, C.c_int]
    _bqp_bound = True
    return L


def _csr_ptrs(M, keep):
    """M = (rowptr, colidx, vals) or None -> ctypes pointers (arrays appended to `keep` so that they stay alive)."""
    if M is None:
        return None, None, None
    p, i, v = (np.ascontiguousarray(M[0], np.int32), np.ascontiguousarray(M[1], np.int32), np.ascontiguousarray(M[2], np.float64))
    keep += [p, i, v]
    return p.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)


class BqpOracle:
    """problem: dict(n, A=(rowptr, colidx, vals), b, x0[, C=(...), d][, E=(...), f])."""

    def __init__(self, problem, preset=None, params=None, order=ORDER_EIGEN, T=256, chunk=512):
        self.L = _bqp_lib()
        self.h = C.c_void_p(self.L.bqpo_create())
        self.n = int(problem["n"])
        self.m = len(problem["d"]) if problem.get("C") is not None else 0
        self.l = len(problem["f"]) if problem.get("E") is not None else 0
        keep = []
        A = _csr_ptrs(problem["A"], keep)
        Cm = _csr_ptrs(problem.get("C"), keep)
        Em = _csr_ptrs(problem.get("E"), keep)
        vec = lambda k: (keep.append(np.ascontiguousarray(problem[k], np.float64)) or keep[-1].ctypes.data_as(C.c_void_p)) if problem.get(k) is not None else None
        rc = self.L.bqpo_set_problem(self.h, self.n, *A, vec("b"), vec("x0"), self.m, *Cm, vec("d"), self.l, *Em, vec("f"))
        if rc:
            raise ValueError(f"bqpo_set_problem failed: {rc}")
        ptype = (1 if self.m else 0) | (2 if self.l else 0)
        self.L.bqpo_preset(self.h, ptype if preset is None else int(preset))
        if params is not None:
            self.L.bqpo_set_params(self.h, np.ascontiguousarray(params, np.float64))
        self.L.bqpo_set_order(self.h, order, T, chunk)

    def __del__(self):
        try:
            if self.h:
                self.L.bqpo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self):
        return self.L.bqpo_solve(self.h)

    def vec(self, name):
        out = np.zeros(max(self.n, self.m, self.l, 1))
        k = self.L.bqpo_get_vec(self.h, name.encode(), out, len(out))
        if k < 0:
            raise KeyError(name)
        return out[:k].copy()

    def scalar(self, name):
        return self.L.bqpo_get_scalar(self.h, name.encode())

    def pcg_trace(self):
        out = np.zeros(20000, np.int32)
        k = self.L.bqpo_get_trace(self.h, out, len(out))
        return out[:k].copy()
