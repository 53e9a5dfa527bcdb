"""Drop-in for the reference's Cython module `Segmentation.cython.src.lpbox`
(Segmentation/Segmentation/cython/src/lpbox.pyx): same class, same methods, HIP kernels underneath.

    from Segmentation.cython.src import lpbox                 # SEG/trainer.py:12
    solver = lpbox.PyLPboxADMMsolver(0, 1e4, it)
"""
from lpbox_hip.seg import PyLPboxADMMsolver, load_gray  # noqa: F401
