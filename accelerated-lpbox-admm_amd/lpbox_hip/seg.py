"""Host-side mirror of the reference's SEGMENTATION solver interface
(Segmentation/Segmentation/cython/src/lpbox.pyx:8-53, "SEG pyx") on top of the C-ABI (include/lpbox_hip.h).

The reference's init decodes `../data/<problem>.jpg` with OpenCV (SEGcpp:690-714).  OpenCV is not available; the library reads
the JPEG itself (lpbox_read_jpeg_gray: libjpeg's Y plane with its default inverse DCT, which is what cv::imread(path, 0) asks
libjpeg for; equal to PIL's grayscale draft bit for bit); the resize (cv::resize INTER_LINEAR) and the cost construction
(SEGcpp:46-248) run in the C-ABI library as well.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib, files
from ._lib import LpboxError, check  # noqa: F401
from .lp import _as_int


def load_gray(path):
    """cv::imread(path, 0): 8-bit grayscale pixels, rows x cols.  Sequential Huffman JPEGs (the reference's inputs) are read by the
    library itself (lpbox_read_jpeg_gray: libjpeg's luminance plane, bit for bit); anything else it refuses -- a progressive JPEG, a
    PNG -- goes through PIL when that is installed."""
    L = _lib.load()
    r, c = C.c_int(), C.c_int()
    rc = L.lpbox_read_jpeg_gray(os.fsencode(path), None, 0, C.byref(r), C.byref(c))
    if rc == 0:
        out = np.zeros((r.value, c.value), np.uint8)
        check(L.lpbox_read_jpeg_gray(os.fsencode(path), out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(c)), "lpbox_read_jpeg_gray")
        return out
    if rc == -4:                      # LPBOX_E_IO: no such file
        check(rc, "lpbox_read_jpeg_gray")
    try:
        from PIL import Image
    except ImportError:
        check(rc, "lpbox_read_jpeg_gray")
    im = Image.open(path)
    im.draft("L", im.size)
    return np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))


class PyLPboxADMMsolver:
    """Same surface as the reference's segmentation `cdef class PyLPboxADMMsolver` (SEG pyx:8-53)."""

    data_root = None      # directory holding <problem>.jpg; default: the reference's CWD-relative "../data" (SEGcpp:690)
    result_root = None    # directory for save_img(); default "../result" (SEGcpp:691)
    verbose = False
    # Side-effect files of the legacy loop (SEGcpp:690-700, 1209-1216, 1270-1277, 1376): <result_root>/xiter_all.csv gets one
    # line per solve and, for print_info 1, <xiter_root>/<problem>.csv every iterate.  None: write where the directory exists
    # (the reference crashes without it); True: create the directories; False: never.
    write_files = None
    xiter_root = None     # default "../xiter" (SEGcpp:696)

    def __init__(self, print_info=0, numNodes=10000, problem=0):
        self._L = _lib.load()
        self.print_info = _as_int(print_info, "print_info")
        self.numNodes = _as_int(numNodes, "numNodes")     # LP/SEG trainers pass 1e4 as a float (SEG/trainer.py:699)
        self.problem = _as_int(problem, "problem")
        h = self._L.lpbox_create(_lib.FLAVOUR_SEG, 1, self.print_info)
        if not h:
            check(-2, "lpbox_create")
        self._h = C.c_void_p(h)
        self._have_problem = False

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpbox_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- extensions: hand the image / the problem over in memory ----
    def set_image(self, gray, numNodes=None):
        gray = np.ascontiguousarray(gray, np.uint8)
        if numNodes is None:
            numNodes = self.numNodes
        check(self._L.lpbox_seg_set_image(self._h, gray.ctypes.data_as(C.c_void_p), gray.shape[0], gray.shape[1], int(numNodes)),
              "lpbox_seg_set_image")
        self._have_problem = True

    def set_problem(self, P):
        check(self._L.lpbox_set_problem_bqp(self._h, int(P["n"]), len(P["colidx"]), np.ascontiguousarray(P["rowptr"], np.int32),
                                            np.ascontiguousarray(P["colidx"], np.int32), np.ascontiguousarray(P["vals"], np.float64),
                                            np.ascontiguousarray(P["b"], np.float64), float(P["c"]), int(P["rows"]), int(P["cols"])),
              "lpbox_set_problem_bqp")
        self._have_problem = True

    def get_problem(self):
        n, nnz, c = C.c_int(), C.c_int(), C.c_double()
        check(self._L.lpbox_seg_get_problem(self._h, C.byref(n), C.byref(nnz), None, None, None, None, C.byref(c)), "lpbox_seg_get_problem")
        rp, ci = np.zeros(n.value + 1, np.int32), np.zeros(nnz.value, np.int32)
        va, b = np.zeros(nnz.value), np.zeros(n.value)
        check(self._L.lpbox_seg_get_problem(self._h, C.byref(n), C.byref(nnz), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                            va.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(c)), "lpbox_seg_get_problem")
        r, cc = C.c_int(), C.c_int()
        check(self._L.lpbox_seg_get_shape(self._h, C.byref(r), C.byref(cc)), "lpbox_seg_get_shape")
        return dict(n=n.value, rowptr=rp, colidx=ci, vals=va, b=b, c=c.value, rows=r.value, cols=cc.value)

    # SEG pyx:17-18
    def solve_init(self):
        if not self._have_problem:
            root = self.data_root or os.environ.get("LPBOX_SEG_DATA_ROOT") or "../data"
            self.set_image(load_gray(os.path.join(root, f"{self.problem}.jpg")))
        return check(self._L.lpbox_init(self._h), "lpbox_init")

    # SEG pyx:20-21
    def solve_iter(self):
        xdir = self._out_dir(self.xiter_root or "../xiter") if self.print_info == 1 else None
        rdir = self._out_dir(self.result_root or "../result")
        check(self._L.lpbox_set_record(self._h, 1 if xdir else 0), "lpbox_set_record")
        e = C.c_int()
        t0 = time.perf_counter()
        check(self._L.lpbox_seg_legacy(self._h, C.byref(e)), "lpbox_seg_legacy")
        ms = int((time.perf_counter() - t0) * 1000)
        if xdir:
            self._write_xiters(os.path.join(xdir, "%d.csv" % self.problem))
        if rdir:
            obj, c = self.debug_scalar("cur_obj"), self.debug_scalar("c")
            files.append_xiter_all(os.path.join(rdir, "xiter_all.csv"), self.problem, obj, obj + c, self.stop()[1], ms * 1.0 / 1000)
        return e.value

    def _out_dir(self, d):
        if self.write_files is False:
            return None
        if self.write_files:
            os.makedirs(d, exist_ok=True)
        return d if os.path.isdir(d) else None

    def x_history(self):
        """(iterations x org_n) iterates of the last recorded legacy solve (lpbox_set_record)."""
        k = check(self._L.lpbox_seg_get_x_history(self._h, 0, 0, None), "lpbox_seg_get_x_history")
        out = np.zeros((k, self.get_org_n()))
        if k:
            check(self._L.lpbox_seg_get_x_history(self._h, 0, k, out.ctypes.data_as(C.c_void_p)), "lpbox_seg_get_x_history")
        return out

    def _write_xiters(self, path):
        files.write_xiters_csv(path, self.x_history())                          # "Iter%d,%lf,...,%lf" (SEGcpp:1270-1277)

    # SEG pyx:23-24
    def solve_iter_l2f(self, i, j, vec, num):
        vec = np.ascontiguousarray(vec, np.float64).ravel()
        num = _as_int(num, "num")
        if num != 0 and vec.shape[0] < self.get_n():
            raise ValueError("fix vector shorter than the number of live variables")
        nums = (C.c_int * 1)(num)
        rets = (C.c_int * 1)(0)
        check(self._L.lpbox_iterate_l2f(self._h, _as_int(i, "i"), _as_int(j, "j"), vec.ctypes.data_as(C.c_void_p), vec.shape[0],
                                        C.cast(nums, C.c_void_p), C.cast(rets, C.c_void_p)), "lpbox_iterate_l2f")
        return rets[0]

    def x_iters_torch(self, ws):
        """The (n_live x ws) iterate window of the last solve_iter_l2f as a zero-copy torch CUDA tensor."""
        import torch
        ptr, stride = C.c_void_p(), C.c_long()
        check(self._L.lpbox_get_x_iters_device(self._h, _as_int(ws, "ws"), C.byref(ptr), C.byref(stride)), "lpbox_get_x_iters_device")
        rows = stride.value // int(ws)
        if rows == 0:
            return torch.zeros((0, int(ws)), dtype=torch.float64, device="cuda")

        class _Dev:
            __cuda_array_interface__ = {"shape": (stride.value,), "typestr": "<f8", "data": (ptr.value, False), "version": 2}
        return torch.as_tensor(_Dev(), device="cuda").view(rows, int(ws))

    # SEG pyx:26-33
    def get_x_iters_2d(self, ws):
        ws = _as_int(ws, "ws")
        rows = check(self._L.lpbox_get_x_iters(self._h, 0, ws, None), "lpbox_get_x_iters")
        out = np.zeros((rows, ws))
        if rows and ws:
            check(self._L.lpbox_get_x_iters(self._h, 0, ws, out.ctypes.data_as(C.c_void_p)), "lpbox_get_x_iters")
        return out

    # SEG pyx:35-39
    def get_n(self):
        return check(self._L.lpbox_get_n(self._h, 0), "lpbox_get_n")

    def get_org_n(self):
        return check(self._L.lpbox_get_org_n(self._h, 0), "lpbox_get_org_n")

    # SEG pyx:41-42
    def get_obj(self):
        v = C.c_double()
        check(self._L.lpbox_seg_get_obj(self._h, C.byref(v)), "lpbox_seg_get_obj")
        return v.value

    # SEG pyx:44-50
    def get_x_sol(self):
        out = np.zeros(self.get_org_n())
        check(self._L.lpbox_get_x_sol(self._h, 0, out), "lpbox_get_x_sol")
        return out.reshape(-1, 1)

    # SEG pyx:52-53 (SEGcpp:812-837): white where x >= 0.5, reshaped column-major to the scaled image
    def save_img(self, path=None):
        from PIL import Image
        r, c = C.c_int(), C.c_int()
        check(self._L.lpbox_seg_get_shape(self._h, C.byref(r), C.byref(c)), "lpbox_seg_get_shape")
        x = self.get_x_sol().ravel()
        img = (x.reshape(c.value, r.value).T >= 0.5).astype(np.uint8) * 255      # Eigen::Map<DenseMatrix>(xx, rows, cols) is column-major
        if path is None:
            root = self.result_root or "../result"
            os.makedirs(root, exist_ok=True)
            path = os.path.join(root, f"output_{self.problem}.png")
        Image.fromarray(img).save(path)
        return path

    # ---- extensions ----
    def config(self):
        t, e, g = C.c_int(), C.c_int(), C.c_int()
        check(self._L.lpbox_get_config(self._h, C.byref(t), C.byref(e), C.byref(g)), "lpbox_get_config")
        return dict(threads=t.value, elems_per_thread=e.value, groups=g.value)

    def counters(self):
        o, p = C.c_longlong(), C.c_longlong()
        check(self._L.lpbox_get_counters(self._h, 0, C.byref(o), C.byref(p)), "lpbox_get_counters")
        return o.value, p.value

    def stop(self):
        r, p = C.c_int(), C.c_int()
        check(self._L.lpbox_get_stop(self._h, 0, C.byref(r), C.byref(p)), "lpbox_get_stop")
        return r.value, p.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_longlong()
        check(self._L.lpbox_kernel_time(self._h, C.byref(ms), C.byref(n), int(bool(reset))), "lpbox_kernel_time")
        return ms.value, n.value

    def debug_vec(self, name):
        out = np.zeros(self.get_org_n())
        check(self._L.lpbox_debug_get_vec(self._h, 0, name.encode(), out, len(out)), "lpbox_debug_get_vec")
        return out

    def debug_scalar(self, name):
        v = C.c_double()
        check(self._L.lpbox_debug_get_scalar(self._h, 0, name.encode(), C.byref(v)), "lpbox_debug_get_scalar")
        return v.value


def solve_batch(solvers):
    """solve_init() + solve_iter() for a list of segmentation solvers (each holding its image / problem) advanced in lockstep by ONE
    launch chain on the GPU (C-ABI lpbox_seg_legacy_batch) -- the reference's own workload, image_segmentation.cpp:24-29: images
    0..99 at 1e4 nodes, where one solve at a time leaves the GPU idle.  Every problem keeps its own control state, so each result is
    bit-identical to solver.solve_init(); solver.solve_iter().  Returns the list of int energies; afterwards every solver answers
    get_obj() / get_x_sol() / counters() as after its own solve.  (The per-solve result files are not written here.)"""
    solvers = list(solvers)
    if not solvers:
        return []
    for s in solvers:
        if not s._have_problem:
            root = s.data_root or os.environ.get("LPBOX_SEG_DATA_ROOT") or "../data"
            s.set_image(load_gray(os.path.join(root, f"{s.problem}.jpg")))
    L = solvers[0]._L
    hs = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
    en = (C.c_int * len(solvers))()
    check(L.lpbox_seg_legacy_batch(hs, len(solvers), en), "lpbox_seg_legacy_batch")
    return [int(v) for v in en]
