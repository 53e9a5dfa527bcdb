"""One large LP, variable-sharded over the ranks of a process group (BASELINE config 5).

Every rank holds a contiguous block of columns of E and runs the large-instance kernels of liblpbox_hip.so on its GPU.  Where the
algorithm sums over all variables (E*v: an l-vector per PCG iteration; a handful of scalars per reduction) the ranks exchange their
contributions and every rank adds them in rank order (reproducible; exact CPU model in the oracle).  Transport:

* `transport="rccl"` (default when torch.distributed runs on "nccl"): the LIBRARY drives RCCL itself on its stream -- grouped
  send/recv of row blocks + one all-gather per E*v, one all-gather per scalar group; Python only hands the 128-byte communicator id
  from rank 0 to the others once (control plane).
* `transport="callback"` (default on "gloo", used by the tests on the one-GPU box): the library calls back into `_allgather`, a
  `torch.distributed.all_gather` through host staging.

With world == 1 no collective is issued and torch is not needed (transport="rccl" with world == 1 builds a one-rank communicator
and runs the whole exchange path against itself: a test of that path on a single GPU).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .dist import shard_range


def _dev_tensor(ptr, count):
    """Zero-copy torch view of `count` doubles of device memory owned by the library."""
    import torch

    class _Dev:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Dev(), device="cuda")


class BigLp:
    def __init__(self, problem, rank=0, world=1, device=0, use_torch_stream=None, transport=None, pcg_mode="reference"):
        """problem: dict(n, l, colptr, rowidx, b[, f]) of the WHOLE instance (CSC, 0/1 pattern, b already negated)."""
        self._L = _lib.load()
        self.rank, self.world = int(rank), int(world)
        n, l = int(problem["n"]), int(problem["l"])
        self.n, self.l = n, l
        self.c0, self.c1 = shard_range(n, world, rank)
        cp = np.asarray(problem["colptr"], np.int64)
        lo, hi = cp[self.c0], cp[self.c1]
        colptr = np.ascontiguousarray(cp[self.c0:self.c1 + 1] - lo, np.int32)
        rowidx = np.ascontiguousarray(np.asarray(problem["rowidx"])[lo:hi], np.int32)
        b = np.ascontiguousarray(np.asarray(problem["b"], np.float64)[self.c0:self.c1])
        f = problem.get("f")
        h = self._L.lpbox_big_create(self.rank, self.world, int(device))
        if not h:
            check(-2, "lpbox_big_create")
        self._h = C.c_void_p(h)
        if transport is None and world > 1:
            import torch.distributed as dist
            transport = "rccl" if dist.get_backend() == "nccl" else "callback"
        self.transport = transport
        self._stream = None
        if world > 1 or use_torch_stream or transport:
            import torch
            torch.cuda.set_device(device)
            self._stream = torch.cuda.current_stream()
            check(self._L.lpbox_big_set_stream(self._h, C.c_void_p(self._stream.cuda_stream)), "lpbox_big_set_stream")
        if transport == "rccl":
            uid = (C.c_ubyte * 128)()
            if self.rank == 0:
                check(self._L.lpbox_big_rccl_unique_id(uid), "lpbox_big_rccl_unique_id")
            if world > 1:                                    # control plane: the id travels once, through whatever group exists
                import torch
                import torch.distributed as dist
                t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.broadcast(t, 0)
                uid = (C.c_ubyte * 128)(*t.cpu().tolist())
            check(self._L.lpbox_big_rccl_init(self._h, uid), "lpbox_big_rccl_init")
        elif transport == "callback":
            self._cb = _lib.ALLGATHER_FN(self._allgather)
            check(self._L.lpbox_big_set_allgather(self._h, C.cast(self._cb, C.c_void_p), None), "lpbox_big_set_allgather")
        elif world > 1:
            raise ValueError(f"unknown transport {transport!r}")
        fp = None
        if f is not None:
            f = np.ascontiguousarray(f, np.float64)
            fp = f.ctypes.data_as(C.c_void_p)
        check(self._L.lpbox_big_set_problem(self._h, n, self.c0, self.c1 - self.c0, l, colptr, rowidx, b, fp), "lpbox_big_set_problem")
        if pcg_mode not in ("reference", "lean"):
            raise ValueError("pcg_mode must be 'reference' or 'lean'")
        if pcg_mode == "lean":       # opt-in comm-lean PCG: not the reference's arithmetic (lpbox_big_set_pcg_mode in include/lpbox_hip.h)
            check(self._L.lpbox_big_set_pcg_mode(self._h, 1), "lpbox_big_set_pcg_mode")

    def _allgather(self, send_ptr, count, recv_ptr, user):
        """recv[r*count:(r+1)*count] := rank r's send[0:count], ordered on the library's stream."""
        try:
            import torch
            import torch.distributed as dist
            with torch.cuda.stream(self._stream):
                send = _dev_tensor(send_ptr, count)
                recv = _dev_tensor(recv_ptr, count * self.world)
                if self.world == 1:
                    recv.copy_(send)
                elif dist.get_backend() == "nccl":
                    dist.all_gather_into_tensor(recv, send)
                else:                       # gloo (tests): stage through the host
                    c = send.cpu()
                    parts = [torch.empty_like(c) for _ in range(self.world)]
                    dist.all_gather(parts, c)
                    recv.copy_(torch.cat(parts))
            return 0
        except Exception as e:          # never let an exception cross the C boundary
            print("lpbox big all-gather failed:", e)
            return 1

    def close(self):
        if getattr(self, "_h", None):
            self._L.lpbox_big_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve_init(self):
        return check(self._L.lpbox_big_init(self._h), "lpbox_big_init")

    def solve_iter(self, i, j):
        ret = C.c_int()
        check(self._L.lpbox_big_iterate(self._h, int(i), int(j), C.byref(ret)), "lpbox_big_iterate")
        return ret.value

    def local_x(self):
        out = np.zeros(self.c1 - self.c0)
        check(self._L.lpbox_big_get_x(self._h, out), "lpbox_big_get_x")
        return out

    def vec(self, name):
        out = np.zeros(max(self.c1 - self.c0, self.l))
        k = check(self._L.lpbox_big_get_vec(self._h, name.encode(), out, len(out)), "lpbox_big_get_vec")
        return out[:k].copy()

    def scalar(self, name):
        v = C.c_double()
        check(self._L.lpbox_big_get_scalar(self._h, name.encode(), C.byref(v)), "lpbox_big_get_scalar")
        return v.value

    def cal_Obj(self):
        v = C.c_double()                       # LPcpp:1630-1642
        check(self._L.lpbox_big_cal_obj(self._h, C.byref(v)), "lpbox_big_cal_obj")
        return v.value

    # ---- early fixing on the sharded instance (ADMM_lp_iters_l2f, LPcpp:1098-1574) ----
    def get_n(self):
        """Live variables of THIS rank."""
        return check(self._L.lpbox_big_get_n(self._h), "lpbox_big_get_n")

    def solve_iter_l2f(self, i, j, vec_local=None, num_global=0):
        """vec_local: fix vector over this rank's live variables (1 / 0 fix, -1 leave); num_global: fixes over all ranks
        (None: count the local ones and sum them over the process group)."""
        vp = None
        if vec_local is not None:
            vec_local = np.ascontiguousarray(vec_local, np.float64).ravel()
            if vec_local.shape[0] < self.get_n():
                raise ValueError("fix vector shorter than this rank's live variables")
            vp = vec_local.ctypes.data_as(C.c_void_p)
        if num_global is None:
            k = 0 if vec_local is None else int(np.count_nonzero((vec_local[:self.get_n()] == 1) | (vec_local[:self.get_n()] == 0)))
            num_global = self.sum_over_ranks(k)
        ret = C.c_int()
        check(self._L.lpbox_big_iterate_l2f(self._h, int(i), int(j), vp, int(num_global), C.byref(ret)), "lpbox_big_iterate_l2f")
        return ret.value

    def sum_over_ranks(self, k):
        if self.world == 1:
            return int(k)
        import torch
        import torch.distributed as dist
        t = torch.tensor([int(k)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        return int(t.item())

    def get_x_iters_2d(self, ws):
        rows = check(self._L.lpbox_big_get_x_iters(self._h, int(ws), None), "lpbox_big_get_x_iters")
        out = np.zeros((rows, int(ws)))
        if rows:
            check(self._L.lpbox_big_get_x_iters(self._h, int(ws), out.ctypes.data_as(C.c_void_p)), "lpbox_big_get_x_iters")
        return out

    def x_iters_torch(self, ws):
        """This rank's (rows x ws) iterate windows as a zero-copy torch CUDA tensor."""
        import torch
        ptr, rows = C.c_void_p(), C.c_int()
        check(self._L.lpbox_big_get_x_iters_device(self._h, int(ws), C.byref(ptr), C.byref(rows)), "lpbox_big_get_x_iters_device")
        if rows.value == 0:
            return torch.zeros((0, int(ws)), dtype=torch.float64, device="cuda")

        class _Dev:
            __cuda_array_interface__ = {"shape": (rows.value * int(ws),), "typestr": "<f8", "data": (ptr.value, False), "version": 2}
        return torch.as_tensor(_Dev(), device="cuda").view(rows.value, int(ws))

    def local_x_sol(self):
        out = np.zeros(self.c1 - self.c0)
        check(self._L.lpbox_big_get_x_sol(self._h, out), "lpbox_big_get_x_sol")
        return out
