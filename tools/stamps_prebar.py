"""Both sides of the two LDS hand-over barriers of a PCG iteration, per wavefront (diagnostic build:
   make -C accelerated-lpbox-admm_amd/csrc variant NAME=stampsprebar EXTRA="-DLPBOX_STAMPS -DLPBOX_STAMPS_PREBAR").
Slots: 12 = p -> LDS up to the barrier, 3 = waiting at that barrier, 4 = row gather, 13 = q -> LDS up to the barrier, 5 = waiting at it,
6 = column gather + Mp, 7 = reduction 1, 8 = alpha + updates, 9 = reduction 2, 10 = beta + p.  Cycles per PCG iteration, first 2000
iterations of the 256-instance batch, averaged over 8 instances; one run per stamped wave (LPBOX_STAMP_WAVE)."""
import os, sys, ctypes as C
os.environ["LPBOX_LIB_VARIANT"] = "stampsprebar"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from bench import load_instances, FIXTURE
from lpbox_hip.lp import LpBatch
insts = load_instances(FIXTURE)[:256]
slots = [(12, "D1a p->LDS"), (3, "D1b wait B1"), (4, "D2 rows"), (13, "D3a q->LDS"), (5, "D3b wait B2"), (6, "D4 cols+Mp"), (7, "D5 red1"),
         (8, "D6 alpha,upd"), (9, "D7 red2"), (10, "D8 beta,p")]
print("wave " + " ".join("%12s" % n for _, n in slots) + "   PCG-iteration")
for wave in range(8):
    os.environ["LPBOX_STAMP_WAVE"] = str(wave)
    b = LpBatch(insts); b.solve_init(); b.solve_iter(0, 2000)
    L = b._L; L.lpbox_debug_get_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    tot = np.zeros(16)
    for i in range(0, 256, 32):
        out = (C.c_ulonglong * 16)(); L.lpbox_debug_get_stamps(b._h, i, out)
        tot += np.array(list(out), float) / b.counters(i)[1]
    tot /= 8
    print("%4d " % wave + " ".join("%12.0f" % tot[k] for k, _ in slots) + "   %8.0f" % sum(tot[k] for k, _ in slots))
    b.close()
