"""On-disk formats of the plain loop: <root>/xiter/<k>_<j>_xiters_<i>.csv (print_info 2/3,
LPcpp:776-779,903-909,940-946) and <root>/xiter/allres.csv (LPcpp:1081), read back the way LP/trainer.py does
(readFile :32-48, get_lpbox_info :189-203)."""
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN
from lpbox_hip.lp import PyLPboxADMMsolver
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _root(tmp_path):
    root = tmp_path / "data"
    shutil.copytree(os.path.join(GOLDEN, "instance"), root / "instance")
    return str(root)


def _read_xiters(path):                       # LP/trainer.py:32-48
    rows = []
    with open(path) as f:
        for line in f:
            rows.append(list(map(float, line.split(",")[1:])))
    return np.transpose(np.array(rows))


def _solver(root, print_info, write):
    class S(PyLPboxADMMsolver):
        data_root = root
        write_files = write
    s = S(print_info)
    s.read_File(1, 100, 500)
    s.solve_init()
    return s


def test_xiters_csv_and_allres(tmp_path):
    root = _root(tmp_path)
    s = _solver(root, 2, True)
    s.solve_iter(0, 1e4)
    X = _read_xiters(os.path.join(root, "xiter", "100_500_xiters_1.csv"))
    reason, p1 = s.batch.stop(0)
    assert reason in (1, 2) and X.shape == (500, p1)
    with open(os.path.join(root, "xiter", "100_500_xiters_1.csv")) as f:
        lines = f.read().splitlines()
    assert lines[0].startswith("Iter1,") and lines[-1].startswith("Iter%d," % p1)
    # last row = the iterate the solver stopped on, printed with %lf
    x_final = s.batch.debug_vec("x")
    assert lines[-1] == "Iter%d," % p1 + ",".join("%f" % v for v in x_final)
    # first row = x after the first iteration: the oracle in the same reduction order agrees to the printed digits
    cfg = s.batch.config()
    o = O.LpOracle(0, order=O.ORDER_GPU, T=cfg["threads"], positions=s.batch.layout(0),
                   npos=cfg["threads"] * cfg["elems_per_thread"], row_split=s.batch.row_split(0), col_split=s.batch.col_split(0))
    o.read_files(os.path.join(root, "instance/100_500/instance_1_C.txt"), os.path.join(root, "instance/100_500/instance_1_b.txt"), 100)
    o.solve_init()
    o.solve_iter(0, 1)
    assert np.array_equal(X[:, 0], np.array([float("%f" % v) for v in o.vec("x")]))
    # allres.csv: idx,-binary objective,iter+1,seconds  (LP/trainer.py:189-203 parses it with float())
    with open(os.path.join(root, "xiter", "allres.csv")) as f:
        rec = [list(map(float, ln.split(","))) for ln in f]
    assert len(rec) == 1 and rec[0][0] == 1 and rec[0][2] == p1 and rec[0][3] >= 0
    assert rec[0][1] == float("%f" % -s.get_curBinObj())
    # a second solve appends
    s2 = _solver(root, 0, None)
    s2.solve_iter(0, 50)
    with open(os.path.join(root, "xiter", "allres.csv")) as f:
        assert len(f.read().splitlines()) == 2


def test_print_info_3_keeps_only_the_stop_iterate(tmp_path):
    root = _root(tmp_path)
    s = _solver(root, 3, True)
    s.solve_iter(0, 1e4)
    with open(os.path.join(root, "xiter", "100_500_xiters_1.csv")) as f:
        lines = f.read().splitlines()
    reason, p1 = s.batch.stop(0)
    assert len(lines) == 1 and lines[0].startswith("Iter%d," % p1)


def test_no_directory_no_files(tmp_path):
    root = _root(tmp_path)
    s = _solver(root, 2, None)                 # default: only write where the reference's directory layout exists
    s.solve_iter(0, 30)
    assert not os.path.exists(os.path.join(root, "xiter"))


def test_recording_does_not_change_the_iteration(tmp_path):
    root = _root(tmp_path)
    a = _solver(root, 2, True)
    b = _solver(root, 0, False)
    a.solve_iter(0, 400)
    b.solve_iter(0, 400)
    assert np.array_equal(a.batch.debug_vec("x"), b.batch.debug_vec("x"))
    assert np.array_equal(a.get_x_iters_2d(400)[:, -1], b.batch.debug_vec("x"))


@pytest.mark.parametrize("fixture,inst,iters", [("lp_100_500_seed0.npz", 0, 20000), ("lp_500_2000_seed0.npz", 2, 300)])
def test_iteration_log_matches_the_oracles(fixture, inst, iters, tmp_path):
    """does_log (LPh:148; LPcpp:789, :898-901, :1013-1067): the per-iteration text log the reference writes by default, opt-in here
    (`write_log`, C-ABI lpbox_set_log / lpbox_get_log: six extra norms per iteration in a logging instantiation of the window kernel).
    Line for line against the file the oracle writes in the kernels' reduction order: iteration headers, PCG counts and the objective
    values exactly as printed, the seven norms to the printed nine decimals (the log's norms use Eigen's association in the oracle and
    the kernel's tree on the GPU: 1e-8 absolute); the elapsed-time lines are not compared.  A full solve (one slot per thread) ends on
    a stop test, so the trailing header without a block is covered; the j=500/k=2000 instance runs the four-slot variant."""
    from helpers import lp_instances, oracle_like
    I = lp_instances(fixture)[inst]
    g = PyLPboxADMMsolver(0)
    g.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    g._file_id, g._root, g.write_files, g.write_log = (1, 7, 9), str(tmp_path), False, True
    g.solve_init()
    o = oracle_like(g, I)
    o.set_log(str(tmp_path / "oracle_log.txt"))
    assert g.solve_iter(0, iters) == o.solve_iter(0, iters)
    o.set_log(None)
    got = open(tmp_path / "log" / "7_9_log_1.txt").read().splitlines()
    want = open(tmp_path / "oracle_log.txt").read().splitlines()
    assert len(got) == len(want) and len(got) > 12 * min(iters, 200)
    for a, b in zip(got, want):
        if a.startswith("Time elapsed"):
            assert b.startswith("Time elapsed")
        elif a.startswith("norm of"):
            ka, va = a.rsplit(": ", 1)
            kb, vb = b.rsplit(": ", 1)
            assert ka == kb and abs(float(va) - float(vb)) <= 1e-8, (a, b)
        elif a.startswith("LongkangIter"):
            fa, fb = a.replace(";", "").split(), b.replace(";", "").split()
            assert fa[:2] == fb[:2] and abs(float(fa[3]) - float(fb[3])) <= 1e-5 and fa[4:] == fb[4:], (a, b)
        else:
            assert a == b
    # the log does not perturb the iteration
    h = PyLPboxADMMsolver(0)
    h.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    h.solve_init()
    h.solve_iter(0, iters)
    assert np.array_equal(h.batch.debug_vec("x").view(np.uint64), g.batch.debug_vec("x").view(np.uint64)) and h.cal_Obj() == g.cal_Obj()
