"""lpbox_hip -- MI355X-native Lp-Box ADMM inner solver behind the reference's Cython solver API.

`lpbox_hip.lp.PyLPboxADMMsolver` mirrors LinerProgramming/LinearProgramming/cython_solver/lpbox.pyx;
`lpbox_hip.lp.LpBatch` runs many independent instances per GPU (one workgroup each).
"""
from . import _lib  # noqa: F401
from .lp import LpBatch, PyLPboxADMMsolver  # noqa: F401
