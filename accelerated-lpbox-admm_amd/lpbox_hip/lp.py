"""Host-side mirror of the reference's LP solver interface on top of the C-ABI (include/lpbox_hip.h).

`PyLPboxADMMsolver` has exactly the methods of the reference's Cython class
(LinerProgramming/LinearProgramming/cython_solver/lpbox.pyx:7-76, "LP pyx" below); `LpBatch` is the batched
extension: B independent instances, one workgroup (one CU) each, per launch.

Reference citations: LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib, files
from ._lib import E_TOOLARGE, LpboxError, check  # noqa: F401

STOP_NAMES = {0: None, 1: "y1_y2", 2: "obj_std", 3: "pcg_alpha_negative", 4: "all_fixed"}


def _as_int(v, name):
    """Cython's `int` argument conversion accepts integral floats such as 1e4 (LP/cython_solver/test.py:10)."""
    iv = int(v)
    if iv != v:
        raise TypeError(f"{name} must be integral, got {v!r}")
    return iv


class LpBatch:
    """B independent LP instances solved together on one GPU (extension; no reference counterpart).

    instances: iterable of dicts with keys n, l, colptr, rowidx (CSC of E, 0/1 pattern), b (already negated,
    LPcpp:2520) and optionally f (default ones, LPcpp:2522).
    """

    def __init__(self, instances=None, print_info=0, device=None, batch=None):
        self._L = _lib.load()
        if device is not None:
            check(self._L.lpbox_set_device(int(device)), "lpbox_set_device")
        instances = list(instances) if instances is not None else None
        self.B = len(instances) if instances is not None else int(batch)
        self.print_info = int(print_info)
        h = self._L.lpbox_create(_lib.FLAVOUR_LP, self.B, self.print_info)
        if not h:
            check(-2, "lpbox_create")
        self._h = C.c_void_p(h)
        if instances is not None:
            for i, I in enumerate(instances):
                self.set_problem(i, I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], I.get("f"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpbox_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- problem input ----
    def set_problem(self, idx, n, l, colptr, rowidx, b, f=None, vals=None):
        colptr = np.ascontiguousarray(colptr, np.int32)
        rowidx = np.ascontiguousarray(rowidx, np.int32)
        b = np.ascontiguousarray(b, np.float64)
        fp = vp = None
        if f is not None:
            f = np.ascontiguousarray(f, np.float64)
            fp = f.ctypes.data_as(C.c_void_p)
        if vals is not None:
            vals = np.ascontiguousarray(vals, np.float64)
            vp = vals.ctypes.data_as(C.c_void_p)
        check(self._L.lpbox_set_problem_lp(self._h, idx, int(n), int(l), len(rowidx), colptr, rowidx, vp, b, fp),
              "lpbox_set_problem_lp")

    def read_files(self, idx, path_C, path_b, k=100):
        check(self._L.lpbox_read_files_lp(self._h, idx, os.fsencode(path_C), os.fsencode(path_b), int(k)),
              "lpbox_read_files_lp")

    def read_file(self, idx, i, k, j, root=None):
        r = os.fsencode(root) if root is not None else None
        check(self._L.lpbox_read_file(self._h, idx, r, int(i), int(k), int(j)), "lpbox_read_file")

    def get_problem(self, idx=0):
        """The instance as set_problem / read_file left it in the handle: dict(n, l, colptr, rowidx, b, f)."""
        n, l, nnz = C.c_int(), C.c_int(), C.c_int()
        check(self._L.lpbox_get_problem_lp(self._h, idx, C.byref(n), C.byref(l), C.byref(nnz), None, None, None, None), "lpbox_get_problem_lp")
        colptr, rowidx = np.zeros(n.value + 1, np.int32), np.zeros(max(nnz.value, 1), np.int32)
        b, f = np.zeros(n.value), np.zeros(l.value)
        check(self._L.lpbox_get_problem_lp(self._h, idx, None, None, None, colptr.ctypes.data_as(C.c_void_p), rowidx.ctypes.data_as(C.c_void_p),
                                           b.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)), "lpbox_get_problem_lp")
        return dict(n=n.value, l=l.value, colptr=colptr, rowidx=rowidx[:nnz.value], b=b, f=f)

    # ---- solver ----
    def solve_init(self):
        return check(self._L.lpbox_init(self._h), "lpbox_init")

    def set_record(self, on=True):
        """Plain loop keeps x of every iteration of a solve_iter call (print_fix_info 2/3, LPcpp:903-909); read them with
        get_x_iters_2d(j - i)."""
        check(self._L.lpbox_set_record(self._h, 1 if on else 0), "lpbox_set_record")

    def set_log(self, on=True):
        """The reference's per-iteration log values (does_log, LPcpp:1013-1067), opt-in: solve_iter keeps one record per iteration."""
        check(self._L.lpbox_set_log(self._h, 1 if on else 0), "lpbox_set_log")

    def get_log(self, idx=0, max_rows=20000):
        """(rows, 12) records of the last solve_iter call: PCG iterations, |x_sol|, |y1|, |y2|, |y3|, |z1|, |z2|, |z4|, dou_obj, bin_obj,
        seconds since the call started, iteration."""
        out = np.zeros((int(max_rows), 12))
        rows = check(self._L.lpbox_get_log(self._h, _as_int(idx, "idx"), out.ravel(), int(max_rows)), "lpbox_get_log")
        return out[:rows].copy()

    def set_x_update(self, mode="pcg"):
        """How the x-update solves its linear system.  "pcg" (default) is the reference's Jacobi-PCG to 1e-3 (LPcpp:251-335, :894) and
        the only mode that reproduces the reference's iterates.  "direct" is an opt-in WITHOUT reference counterpart: an exact solve
        through a dense l x l inverse kept on chip -- several times faster per ADMM iteration, different
        iterates / iteration counts; raises for batches it does not fit (n <= 512, at most 128 rows sharing columns)."""
        if mode not in ("pcg", "direct"):
            raise ValueError("x-update mode must be 'pcg' or 'direct'")
        check(self._L.lpbox_set_x_update(self._h, 1 if mode == "direct" else 0), "lpbox_set_x_update")

    def direct_rows(self, idx=0):
        """Row split of the direct x-update (lpbox_get_direct_rows): dense index per row of E, -1 = closed-form row."""
        out = np.zeros(self.get_l(idx), np.int32)
        check(self._L.lpbox_get_direct_rows(self._h, idx, out.ctypes.data_as(C.c_void_p)), "lpbox_get_direct_rows")
        return out

    def solve_iter(self, i, j):
        rets = np.zeros(self.B, np.int32)
        check(self._L.lpbox_iterate(self._h, _as_int(i, "i"), _as_int(j, "j"), rets.ctypes.data_as(C.c_void_p)),
              "lpbox_iterate")
        return rets

    def solve_iter_l2f(self, i, j, vecs=None, nums=None):
        rets = np.zeros(self.B, np.int32)
        vp = npn = None
        stride = 0
        if nums is not None and np.any(np.asarray(nums) != 0):
            vecs = np.ascontiguousarray(vecs, np.float64).reshape(self.B, -1)
            stride = vecs.shape[1]
            nums = np.ascontiguousarray(nums, np.int32)
            for b in range(self.B):
                if nums[b] and stride < self.get_n(b):
                    raise ValueError("fix vector shorter than the number of live variables")
            vp = vecs.ctypes.data_as(C.c_void_p)
            npn = nums.ctypes.data_as(C.c_void_p)
        check(self._L.lpbox_iterate_l2f(self._h, _as_int(i, "i"), _as_int(j, "j"), vp, stride, npn,
                                        rets.ctypes.data_as(C.c_void_p)), "lpbox_iterate_l2f")
        return rets

    def set_active(self, active=None):
        """Park (False) / resume (True) instances of the batch; parked ones are skipped by solve_iter / solve_iter_l2f."""
        a = None if active is None else np.ascontiguousarray(np.asarray(active) != 0, np.int32)
        if a is not None and a.shape != (self.B,):
            raise ValueError("active must have one entry per instance")
        check(self._L.lpbox_set_active(self._h, None if a is None else a.ctypes.data_as(C.c_void_p)), "lpbox_set_active")

    # ---- results ----
    def get_n(self, idx=0):
        return check(self._L.lpbox_get_n(self._h, idx), "lpbox_get_n")

    def get_org_n(self, idx=0):
        return check(self._L.lpbox_get_org_n(self._h, idx), "lpbox_get_org_n")

    def get_l(self, idx=0):
        return check(self._L.lpbox_get_l(self._h, idx), "lpbox_get_l")

    def get_iter(self, idx=0):
        return check(self._L.lpbox_get_iter(self._h, idx), "lpbox_get_iter")

    def get_x_iters_2d(self, ws, idx=0):
        ws = _as_int(ws, "ws")
        rows = check(self._L.lpbox_get_x_iters(self._h, idx, ws, None), "lpbox_get_x_iters")
        out = np.zeros((rows, ws), np.float64)
        if rows and ws:
            check(self._L.lpbox_get_x_iters(self._h, idx, ws, out.ctypes.data_as(C.c_void_p)), "lpbox_get_x_iters")
        return out

    def x_iters_torch(self, ws):
        """The (n_live_i x ws) iterate windows of the whole batch as ONE torch CUDA tensor view (zero copy): returns
        (flat, stride) where instance i is flat[i*stride : i*stride + rows_i*ws].view(rows_i, ws)."""
        import torch
        ptr, stride = C.c_void_p(), C.c_long()
        check(self._L.lpbox_get_x_iters_device(self._h, _as_int(ws, "ws"), C.byref(ptr), C.byref(stride)), "lpbox_get_x_iters_device")

        class _Dev:     # the CUDA array interface is how torch adopts foreign device memory without copying
            __cuda_array_interface__ = {"shape": (self.B * stride.value,), "typestr": "<f8", "data": (ptr.value, False), "version": 2}
        return torch.as_tensor(_Dev(), device="cuda"), stride.value

    def get_x_sol(self, idx=0):
        out = np.zeros(self.get_org_n(idx), np.float64)
        check(self._L.lpbox_get_x_sol(self._h, idx, out), "lpbox_get_x_sol")
        return out

    def get_final_x_sol(self, idx=0):
        out = np.zeros(self.get_org_n(idx), np.float64)
        k = check(self._L.lpbox_get_final_x_sol(self._h, idx, out), "lpbox_get_final_x_sol")
        return out[:k].copy()

    def cal_obj(self, idx=0):
        v = C.c_double()
        check(self._L.lpbox_cal_obj(self._h, idx, C.byref(v)), "lpbox_cal_obj")
        return v.value

    def cur_bin_obj(self, idx=0):
        v = C.c_double()
        check(self._L.lpbox_cur_bin_obj(self._h, idx, C.byref(v)), "lpbox_cur_bin_obj")
        return v.value

    def check_infeasible_lpbox(self, idx=0):
        return check(self._L.lpbox_check_infeasible_lpbox(self._h, idx), "lpbox_check_infeasible_lpbox")

    def check_infeasible_l2f(self, idx=0):
        return check(self._L.lpbox_check_infeasible_l2f(self._h, idx), "lpbox_check_infeasible_l2f")

    # ---- extensions ----
    def config(self):
        t, e, l = C.c_int(), C.c_int(), C.c_int()
        check(self._L.lpbox_get_config(self._h, C.byref(t), C.byref(e), C.byref(l)), "lpbox_get_config")
        return dict(threads=t.value, elems_per_thread=e.value, lds_bytes=l.value)

    def layout(self, idx=0):
        """Storage position of every variable (the reduction tree of the kernels is defined over positions)."""
        pos = np.zeros(self.get_org_n(idx), np.int32)
        check(self._L.lpbox_get_layout(self._h, idx, pos), "lpbox_get_layout")
        return pos

    def row_split(self, idx=0):
        """Lanes (1,2,4,8) sharing the sum of each row of E inside the kernels."""
        g = np.zeros(self.get_l(idx), np.int32)
        check(self._L.lpbox_get_row_split(self._h, idx, g), "lpbox_get_row_split")
        return g

    def col_split(self, idx=0):
        """(own[n], help4[n, 4]): how the kernels associate the sum over each column of E (include/lpbox_hip.h)."""
        n = self.get_org_n(idx)
        own, help4 = np.zeros(n, np.int32), np.zeros(4 * n, np.int32)
        check(self._L.lpbox_get_col_split(self._h, idx, own, help4), "lpbox_get_col_split")
        return own, help4.reshape(n, 4)

    def counters(self, idx=0):
        o, p = C.c_longlong(), C.c_longlong()
        check(self._L.lpbox_get_counters(self._h, idx, C.byref(o), C.byref(p)), "lpbox_get_counters")
        return o.value, p.value

    def stop(self, idx=0):
        r, p = C.c_int(), C.c_int()
        check(self._L.lpbox_get_stop(self._h, idx, C.byref(r), C.byref(p)), "lpbox_get_stop")
        return r.value, p.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_longlong()
        check(self._L.lpbox_kernel_time(self._h, C.byref(ms), C.byref(n), int(bool(reset))), "lpbox_kernel_time")
        return ms.value, n.value

    def debug_vec(self, name, idx=0):
        cap = max(self.get_org_n(idx), self.get_l(idx))
        out = np.zeros(cap, np.float64)
        k = check(self._L.lpbox_debug_get_vec(self._h, idx, name.encode(), out, cap), "lpbox_debug_get_vec")
        return out[:k].copy()

    def debug_scalar(self, name, idx=0):
        v = C.c_double()
        check(self._L.lpbox_debug_get_scalar(self._h, idx, name.encode(), C.byref(v)), "lpbox_debug_get_scalar")
        return v.value


ONCHIP_MAX = 2048       # storage positions of the one-workgroup-per-instance kernel (512 threads x 4 slots): max(n, l) beyond it cannot stay on a CU


class _LargeInstance:
    """The slice of LpBatch's surface that PyLPboxADMMsolver uses, served by the large-instance path (lpbox_big_*, one rank): the
    instance lives in HBM and an iteration is a chain of kernels instead of one persistent workgroup.  Same algorithm, same
    quirks, its own (two-level) summation order -- bit-exact against the oracle in that order (tests/test_dropin_large_gpu.py)."""

    B = 1
    can_record = True        # the plain loop's per-iteration dump (print_info 2/3): lpbox_big_set_record stages the iterates on the device

    def __init__(self, problem, device=None):
        from .big import BigLp
        self._g = BigLp(problem, device=0 if device is None else int(device))
        self._org_n = int(problem["n"])
        self._l = int(problem["l"])

    def close(self):
        self._g.close()

    def solve_init(self):
        return self._g.solve_init()

    def set_record(self, on=True):
        check(self._g._L.lpbox_big_set_record(self._g._h, 1 if on else 0), "lpbox_big_set_record")

    def solve_iter(self, i, j):
        return np.array([self._g.solve_iter(_as_int(i, "i"), _as_int(j, "j"))], np.int32)

    def solve_iter_l2f(self, i, j, vecs=None, nums=None):
        num = 0 if nums is None else int(np.asarray(nums).ravel()[0])
        vec = None if (vecs is None or num == 0) else np.asarray(vecs, np.float64).ravel()
        return np.array([self._g.solve_iter_l2f(_as_int(i, "i"), _as_int(j, "j"), vec, num)], np.int32)

    def get_n(self, idx=0):
        return self._g.get_n()

    def get_org_n(self, idx=0):
        return self._org_n

    def get_l(self, idx=0):
        return self._l

    def get_iter(self, idx=0):
        return int(self._g.scalar("iter"))

    def get_x_iters_2d(self, ws, idx=0):
        return self._g.get_x_iters_2d(_as_int(ws, "ws"))

    def get_x_sol(self, idx=0):
        return self._g.local_x_sol()

    def get_final_x_sol(self, idx=0):
        return self._g.local_x()[self._g.vec("live")[: self._org_n] != 0]          # raw live x, compact order (LPcpp:1668-1685)

    def cal_obj(self, idx=0):
        return self._g.cal_Obj()

    def cur_bin_obj(self, idx=0):
        return self._g.scalar("cur_obj")

    def check_infeasible_lpbox(self, idx=0):
        return check(self._g._L.lpbox_big_check_infeasible(self._g._h, 0), "lpbox_big_check_infeasible")

    def check_infeasible_l2f(self, idx=0):
        return check(self._g._L.lpbox_big_check_infeasible(self._g._h, 1), "lpbox_big_check_infeasible")

    def stop(self, idx=0):
        return int(self._g.scalar("stop")), int(self._g.scalar("plain_iter_p1"))

    def counters(self, idx=0):
        return int(self._g.scalar("outer_total")), int(self._g.scalar("pcg_total"))

    def debug_scalar(self, name, idx=0):
        return self._g.scalar(name)

    def debug_vec(self, name, idx=0):
        return self._g.vec(name)

    def config(self):
        """Reduction geometry of this path (what the oracle needs to mirror it): threads per workgroup and columns per workgroup."""
        return dict(threads=int(self._g.scalar("threads")), chunk=int(self._g.scalar("chunk")), large=True)

    @property
    def big(self):
        return self._g


class PyLPboxADMMsolver:
    """Same surface as the reference's `cdef class PyLPboxADMMsolver` (LP pyx:7-76), one instance per object.

    Instances up to max(n, l) = 2048 run on the on-chip kernel (one workgroup per instance); larger ones are handed to the
    large-instance path at solve_init, transparently (`large` tells which; the reference has no size limit, LPcpp:2446-2545).

    print_info: 0 quiet, 1 print fix sizes (LPcpp:1188-1190).  Set `verbose = True` to echo the reference's
    stdout messages (constructor banner LPcpp:478, stop reasons :935/:984, fix summary :1333).
    `data_root` (or the LPBOX_DATA_ROOT environment variable) replaces the reference's CWD-relative
    "../cython_solver/data" (LPcpp:2451).
    """

    data_root = None
    write_log = False        # True: solve_iter writes the reference's per-iteration text log <root>/log/<k>_<j>_log_<i>.txt (LPcpp:1013-1067)
    verbose = False
    # Side-effect files of ADMM_lp_iters (LPcpp:776-783, 903-909, 940-946, 986-992, 1081): <root>/xiter/allres.csv gets one line
    # per plain solve and, for print_info 2/3, <root>/xiter/<k>_<j>_xiters_<i>.csv the iterates.  None: write them iff
    # <root>/xiter exists (the reference crashes without it); True: create the directory; False: never.
    write_files = None

    def __init__(self, print_info=0, *unused):
        if unused:
            # lpbox.pyx defines __cinit__ twice (:10-11, :13-14); the 1-argument form is the one every caller uses
            raise TypeError("PyLPboxADMMsolver takes a single int (print_info)")
        self._b = LpBatch(batch=1, print_info=_as_int(print_info, "print_info"))
        self.print_info = int(print_info)
        if self.verbose:
            print("Object with fix_info is created!")

    def _small(self):
        """A new problem on an object that had been routed to the large path starts from the on-chip handle again."""
        self._x_prev = None
        if not isinstance(self._b, LpBatch):
            self._b.close()
            self._b = LpBatch(batch=1, print_info=self.print_info)
        return self._b

    # LP pyx:16-17
    def read_File(self, i, k, j):
        self._small()
        root = self.data_root or os.environ.get("LPBOX_DATA_ROOT")
        self._b.read_file(0, _as_int(i, "i"), _as_int(k, "k"), _as_int(j, "j"), root)
        self._file_id = (int(i), int(k), int(j))
        self._root = root if root is not None else os.path.join("..", "cython_solver", "data")      # LPcpp:2451
        return None

    # extension: hand the problem over in memory instead of through the instance files
    def set_problem(self, n, l, colptr, rowidx, b, f=None):
        self._small().set_problem(0, n, l, colptr, rowidx, b, f)

    # LP pyx:19-20
    def solve_init(self):
        self._x_prev = None                                # x_prev = Zero(n) in ADMM_lp_iters_init (LPcpp:572)
        if isinstance(self._b, LpBatch):
            P = self._b.get_problem(0)
            fits = max(P["n"], P["l"]) <= ONCHIP_MAX
            if fits:
                try:
                    return self._b.solve_init()
                except LpboxError as e:                    # fits the register slots but not the CU's 160 KiB of LDS (very dense E)
                    if e.code != E_TOOLARGE:
                        raise
            small = self._b                                # does not fit one CU: same algorithm on the multi-kernel path
            self._b = _LargeInstance(P)
            small.close()
        return self._b.solve_init()

    @property
    def large(self):
        """True once solve_init has routed this instance to the large-instance path."""
        return isinstance(self._b, _LargeInstance)

    # LP pyx:22-23
    def solve_iter(self, i, j):
        out_dir = self._xiter_dir()
        dump = out_dir is not None and self.print_info in (2, 3) and _as_int(j, "j") > _as_int(i, "i") and getattr(self._b, "can_record", True)
        # print_info 3 writes only the iterate of the stop (LPcpp:940-946): on the large-instance route that is read back after the solve
        # instead of staging every iterate of a long call in HBM
        self._final_only = dump and self.print_info == 3 and isinstance(self._b, _LargeInstance)
        self._b.set_record(dump and not self._final_only)
        log_path = self._log_path() if _as_int(j, "j") > _as_int(i, "i") else None
        if hasattr(self._b, "set_log"):
            self._b.set_log(log_path is not None)
        t0 = time.perf_counter()
        ret = int(self._b.solve_iter(i, j)[0])
        secs = int((time.perf_counter() - t0) * 1000) / 1000.0          # the reference truncates to whole ms (LPcpp:1079-1080)
        self._echo_stop(plain=True)
        if out_dir is not None:
            self._write_plain_files(out_dir, int(i), int(j), secs, dump)
        if log_path is not None:
            reason, p1 = self._b.stop(0)
            files.write_iteration_log(log_path, self._b.get_log(0, int(j) - int(i)), stopped_in=p1 - 1 if reason in (1, 2) else None)
        return ret

    def _log_path(self):
        """<root>/log/<k>_<j>_log_<i>.txt (LPcpp:2496) when `write_log` is set -- the reference writes it by default (does_log = 1, LPh:148);
        here it is opt-in, and only the on-chip PCG kernels produce it."""
        if not getattr(self, "write_log", False) or getattr(self, "_file_id", None) is None or not isinstance(self._b, LpBatch):
            return None
        d = os.path.join(self._root, "log")
        os.makedirs(d, exist_ok=True)
        fi, k, jj = self._file_id
        return os.path.join(d, "%d_%d_log_%d.txt" % (k, jj, fi))

    def _xiter_dir(self):
        if self.write_files is False or getattr(self, "_file_id", None) is None:
            return None
        d = os.path.join(self._root, "xiter")
        if self.write_files:
            os.makedirs(d, exist_ok=True)
        return d if os.path.isdir(d) else None

    def _write_plain_files(self, out_dir, i, j, secs, dump):
        fi, k, jj = self._file_id
        reason, p1 = self._b.stop(0)
        if dump and getattr(self, "_final_only", False):
            if reason in (1, 2):
                files.write_xiters_csv(os.path.join(out_dir, "%d_%d_xiters_%d.csv" % (k, jj, fi)), self._b.get_final_x_sol(0).reshape(1, -1), p1 - 1)
        elif dump:
            done = (p1 - i) if reason in (1, 2) else (j - i)           # iterations this call ran (break leaves iter at the stop)
            X = self._b.get_x_iters_2d(j - i, 0)[:, :done].T           # one row per iteration
            lo = 0 if self.print_info == 2 else (done - 1 if reason in (1, 2) else done)   # 3: only the iterate of the stop
            files.write_xiters_csv(os.path.join(out_dir, "%d_%d_xiters_%d.csv" % (k, jj, fi)), X[lo:done], i + lo)
        files.append_allres(os.path.join(out_dir, "allres.csv"), fi, -self._b.cur_bin_obj(0), p1, secs)

    # LP pyx:25-26
    def cal_Obj(self):
        return self._b.cal_obj(0)

    # LP pyx:28-29
    def get_curBinObj(self):
        return self._b.cur_bin_obj(0)

    # LP pyx:31-32
    def solve_iter_l2f(self, i, j, vec, num):
        vec = np.ascontiguousarray(vec, np.float64).ravel()
        num = _as_int(num, "num")
        n_before = self._b.get_n(0)
        if num != 0 and vec.shape[0] < n_before:
            raise ValueError(f"vec has {vec.shape[0]} entries, need at least n_live = {n_before}")
        ret = int(self._b.solve_iter_l2f(i, j, vec.reshape(1, -1), np.array([num], np.int32))[0])
        if num != 0 and self.verbose:
            print("Iter: %d; Fixed %d Elements; Totally Fixed %d Elements; Left %d Elements; Sum_fix_obj: %f" % (
                int(i), num, self._b.get_org_n(0) - (n_before - num), n_before - num,
                self._b.debug_scalar("sum_fix_obj", 0)))
        self._echo_stop(plain=False)
        return ret

    # not in the pyx: ADMM_lp_iters_fix (LPcpp:1689-2286), the rule-based early fixing the C++ class carries but exposes to no caller
    def solve_iter_fix(self, i, j, consistency=5, fix_threshold=1e-3, min_fix=10):
        """Iterations [i, j) with the persistence rule of ADMM_lp_iters_fix: a variable whose iterate moved by <= fix_threshold
        for `consistency` consecutive iterations is flagged (LPcpp:1857-1871); once more than `min_fix` variables are flagged
        (:1932) they are fixed at their rounded value (:2006) and leave the problem.

        REPAIRED semantics, stated once: as written, the reference's fix branch (tmp == 1, :1935-2043) compacts x, y, z, b, E and f
        but never rebuilds E^T, rho4*E^T, the diagonal and the preconditioner (no update_expression), so its next iteration adds
        vectors of the old and the new length (:1777-1783) -- an Eigen assertion / out-of-bounds access: the function cannot run
        past its first fix, and nothing calls it.  Here a fix is the fix block of ADMM_lp_iters_l2f (:1124-1335, which the dead
        tmp == 2 branch :2045-2280 copies, update_expression included), and the per-variable counters follow their variable
        through the compaction (the reference would index counters of the old length with new indices, :1861-1871).
        Everything else is the loop as written: counters re-zeroed per call (:1702-1703), x_prev kept across calls (:572, :1871),
        z4 always accumulated, y1_y2 stop returns 0 (:1880-1886), obj_std stop and a failed PCG return 1 (:1804-1807, :1913).
        Host-driven: one single-iteration l2f window per iteration (the rule needs every iterate on the host anyway)."""
        i, j = _as_int(i, "i"), _as_int(j, "j")
        n_live = self._b.get_n(0)
        prev = getattr(self, "_x_prev", None)
        if prev is None or prev.shape[0] != n_live:
            prev = np.zeros(n_live)                                   # x_prev = Zero(n) (:572)
        count, flag = np.zeros(n_live), np.zeros(n_live, bool)
        vec, num, ret = None, 0, 0
        for it in range(i, j):
            r = self.solve_iter_l2f(it, it + 1, vec if num else np.zeros(n_live), num)
            if num:                                                    # the fix went in at the start of this call
                keep = vec == -1
                prev, count, flag = prev[keep], count[keep], flag[keep]
                n_live = int(keep.sum())
                vec, num = None, 0
            reason, _ = self._b.stop(0)
            if reason in (3, 4) or (r and reason == 0):                # PCG alpha < 0 (:1804-1807) / everything fixed / |x| < 1e-3 (:2000)
                ret = 1
                break
            x = self._b.get_x_iters_2d(1, 0)[:, 0]
            det = np.abs(x - prev) <= fix_threshold                    # :1861
            count = np.where(det, count + 1, 0.0)
            flag |= det & (count >= consistency)
            prev = x.copy()                                            # :1871
            if reason == 1:                                            # y1_y2: break, ret stays 0 (:1880-1886)
                break
            if reason == 2:                                            # obj_std (:1913-1919)
                ret = 1
                break
            fix_n = int(flag.sum())                                    # :1929-1932
            if fix_n > min_fix:
                vec = np.where(flag, np.where(x >= 0.5, 1.0, 0.0), -1.0)
                num = fix_n
        if num:                                                        # a fix decided by the last iteration goes in now (:1935), zero-length window
            r = self.solve_iter_l2f(j, j, vec, num)
            keep = vec == -1
            prev = prev[keep]
            ret = ret or int(r)
        self._x_prev = prev
        return ret

    # LP pyx:35-41
    def get_x_iters_1d(self, ws):
        ws = _as_int(ws, "ws")
        if ws < 20:
            raise ValueError("get_x_iters_1d reads n*20 entries (LP pyx:38-40): ws must be >= 20")
        flat = self._b.get_x_iters_2d(ws, 0).ravel()
        n = self._b.get_n(0)
        return flat[: n * 20].reshape(n * 20, 1).copy()

    # LP pyx:43-50
    def get_x_iters_2d(self, ws):
        return self._b.get_x_iters_2d(ws, 0)

    # LP pyx:52-53
    def get_n(self):
        return self._b.get_n(0)

    # LP pyx:55-56
    def get_iter(self):
        return self._b.get_iter(0)

    # LP pyx:58-63
    def get_x_sol(self, n=None):
        x = self._b.get_x_sol(0)
        n = x.shape[0] if n is None else _as_int(n, "n")
        return x[:n].reshape(n, 1)

    # LP pyx:65-70
    def get_final_x_sol(self, n=None):
        x = self._b.get_final_x_sol(0)
        n = x.shape[0] if n is None else _as_int(n, "n")
        out = np.zeros((n, 1))
        out[: min(n, x.shape[0]), 0] = x[:n]
        return out

    # LP pyx:72-73
    def check_infeasible_lpbox(self):
        return self._b.check_infeasible_lpbox(0)

    # LP pyx:75-76
    def check_infeasible_l2f(self):
        return self._b.check_infeasible_l2f(0)

    # ---- extensions ----
    @property
    def batch(self):
        return self._b

    def _echo_stop(self, plain):
        if not self.verbose:
            return
        reason, p1 = self._b.stop(0)
        it = (p1 - 1) if plain else self._b.get_iter(0)
        if reason == 1:
            c = max(self._b.debug_scalar("cvg1"), self._b.debug_scalar("cvg2"))
            print(("Stop because y1_y2. iter: %d, stop_threshold: %.6f" if plain else
                   "Stop becuase y1_y2. iter: %d, stop_threshold: %.6f") % (it, c))
        elif reason == 2:
            print(("Stop because obj_std. iter: %d, std_threshold: %.6f" if plain else
                   "Stop because std_obj. iter: %d, std_threshold: %.6f") % (it, self._b.debug_scalar("std_obj")))
