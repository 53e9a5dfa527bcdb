"""Drop-in for the reference's Cython module `LinearProgramming.cython_solver.lpbox`
(LinerProgramming/LinearProgramming/cython_solver/lpbox.pyx): same class, same methods, HIP kernels underneath.

    from LinearProgramming.cython_solver import lpbox      # LP/trainer.py:12
    solver = lpbox.PyLPboxADMMsolver(0)
"""
from lpbox_hip.lp import LpBatch, PyLPboxADMMsolver  # noqa: F401
