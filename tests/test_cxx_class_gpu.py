"""The C++ host-side class (accelerated-lpbox-admm_amd/cxx/LinearProgramming/cython_solver/LPboxADMMsolver.{h,cpp}: the reference's class
interface over the C-ABI) driven by its command-line program `lp_solve` -- the counterpart of the reference's `./test i k j`
(test.cpp:10-33, BASELINE configs[0]).  The binary is built here with g++ and run as a child process; its results must be those of
the Python class on the same instance (which the other tests hold bit-exact against the oracle) and of the oracle itself."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, lp_instances, oracle_like
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX_DIR = os.path.join(ROOT, "accelerated-lpbox-admm_amd", "cxx", "LinearProgramming", "cython_solver")
SEG_CXX_DIR = os.path.join(ROOT, "accelerated-lpbox-admm_amd", "cxx", "Segmentation", "cython", "src")


def build_driver(tmp_path):
    exe = str(tmp_path / "lp_solve")
    subprocess.check_call(["make", "-s", "-C", CXX_DIR, "OUT=" + exe])
    return exe


def run_driver(exe, root, *args):
    env = dict(os.environ, LPBOX_DATA_ROOT=str(root))
    p = subprocess.run([exe] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    m = re.search(r"^RESULT (.*)$", p.stdout, re.M)
    assert m, p.stdout
    return dict(kv.split("=") for kv in m.group(1).split()), p.stdout


@pytest.mark.gpu
def test_cxx_driver_equals_python_class_and_oracle(tmp_path):
    from lpbox_hip.lp import PyLPboxADMMsolver
    exe = build_driver(tmp_path)
    res, out = run_driver(exe, GOLDEN, 1, 100, 500)
    assert "Object with fix_info is created!" in out and "Stop because" in out and "Total constraints: [189]" in out
    assert "this is feasiblibity: %s" % res["infeasible"] in out
    g = PyLPboxADMMsolver(0)
    g.data_root = GOLDEN
    g.write_files = False
    g.read_File(1, 100, 500)
    g.solve_init()
    ret = g.solve_iter(0, 20000)
    assert int(res["ret"]) == ret and int(res["large"]) == 0
    assert float(res["objective"]) == -g.cal_Obj()
    assert int(res["iterations"]) == g.batch.counters()[0]
    assert int(res["infeasible"]) == g.check_infeasible_l2f()
    assert int(res["ones"]) == int(g.get_x_sol().sum()) and int(res["n"]) == 500
    I = lp_instances("lp_100_500_seed0.npz")[0]
    o = oracle_like(g, I)
    assert o.solve_iter(0, 20000) == ret and float(res["objective"]) == -o.cal_Obj()
    # the early-fixing entry point and the iterate window from C++: 3 windows of 100 iterations, nothing fixed
    res2, _ = run_driver(exe, GOLDEN, 1, 100, 500, 300, 100)
    h = PyLPboxADMMsolver(0)
    h.data_root = GOLDEN
    h.read_File(1, 100, 500)
    h.solve_init()
    chk = 0.0
    for w in range(3):
        h.solve_iter_l2f(100 * w, 100 * (w + 1), np.zeros(500), 0)
        for v in h.get_x_iters_2d(100).ravel():          # the driver adds the entries one by one in this order
            chk += v
    assert float(res2["checksum"]) == chk
    assert float(res2["objective"]) == -h.cal_Obj()


@pytest.mark.gpu
def test_cxx_side_effect_files_equal_python(tmp_path):
    """print_info = 2 (LPcpp:776-783, :903-909, :1081): the C++ class writes <root>/xiter/<k>_<j>_xiters_<i>.csv and appends to allres.csv
    like the Python wrapper, whose formats are pinned to the reference's own readers (tests/test_trainer_pins.py): same bytes."""
    import shutil
    from lpbox_hip.lp import PyLPboxADMMsolver
    exe = build_driver(tmp_path)
    roots = []
    for name in ("cxx", "py"):
        root = tmp_path / name
        os.makedirs(root / "instance" / "100_500")
        os.makedirs(root / "xiter")
        for f in ("instance_1_C.txt", "instance_1_b.txt"):
            shutil.copy(os.path.join(GOLDEN, "instance", "100_500", f), root / "instance" / "100_500" / f)
        roots.append(root)
    run_driver(exe, roots[0], 1, 100, 500, 300, 0, 2)
    g = PyLPboxADMMsolver(2)
    g.data_root = str(roots[1])
    g.read_File(1, 100, 500)
    g.solve_init()
    g.solve_iter(0, 300)
    a = open(roots[0] / "xiter" / "100_500_xiters_1.csv").read()
    b = open(roots[1] / "xiter" / "100_500_xiters_1.csv").read()
    assert a == b and a.count("\n") == 300 and a.startswith("Iter1,")
    la = open(roots[0] / "xiter" / "allres.csv").read().strip().split(",")
    lb = open(roots[1] / "xiter" / "allres.csv").read().strip().split(",")
    assert la[:3] == lb[:3] and len(la) == 4                       # instance, -objective, iterations (the fourth field is the wall-clock)


@pytest.mark.gpu
def test_cxx_iteration_log_and_large_route_dump_equal_python(tmp_path):
    """(1) does_log (LPcpp:1013-1067): the C++ class writes <root>/log/<k>_<j>_log_<i>.txt like the Python wrapper (whose lines are held
    against the oracle's log in tests/test_lp_files_gpu.py): same bytes except the elapsed-time lines.  (2) print_info 2 on an instance
    beyond the on-chip kernel: the iterate dump of the large-instance route, same bytes as the Python wrapper's."""
    import shutil
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like, write_instance_files
    exe = build_driver(tmp_path)
    roots = []
    for name in ("cxx", "py"):
        root = tmp_path / name
        os.makedirs(root / "instance" / "100_500")
        os.makedirs(root / "log")
        for f in ("instance_1_C.txt", "instance_1_b.txt"):
            shutil.copy(os.path.join(GOLDEN, "instance", "100_500", f), root / "instance" / "100_500" / f)
        roots.append(root)
    run_driver(exe, roots[0], 1, 100, 500, 20000, 0, 0, 1)
    g = PyLPboxADMMsolver(0)
    g.data_root, g.write_files, g.write_log = str(roots[1]), False, True
    g.read_File(1, 100, 500)
    g.solve_init()
    g.solve_iter(0, 20000)
    a = [ln for ln in open(roots[0] / "log" / "100_500_log_1.txt") if not ln.startswith("Time elapsed")]
    b = [ln for ln in open(roots[1] / "log" / "100_500_log_1.txt") if not ln.startswith("Time elapsed")]
    assert a == b and len(a) > 50000 and a[-1].startswith("Iteration: ")
    P = make_auction_like(2300, 7)
    for root in roots:
        d = root / "instance" / "1000_2300"
        os.makedirs(d)
        os.makedirs(root / "xiter")
        write_instance_files(P, str(d / "instance_1_C.txt"), str(d / "instance_1_b.txt"))
    res, _ = run_driver(exe, roots[0], 1, 1000, 2300, 25, 0, 2)
    assert int(res["large"]) == 1
    h = PyLPboxADMMsolver(2)
    h.data_root = str(roots[1])
    h.read_File(1, 1000, 2300)
    h.solve_init()
    h.solve_iter(0, 25)
    assert h.large
    xa = open(roots[0] / "xiter" / "1000_2300_xiters_1.csv").read()
    assert xa == open(roots[1] / "xiter" / "1000_2300_xiters_1.csv").read() and xa.count("\n") == 25


@pytest.mark.gpu
def test_cxx_rule_based_fixing_equals_python(tmp_path):
    """ADMM_lp_iters_fix through the C++ class (LPcpp:1689-2286, repaired semantics of DESIGN.md section 16) takes the decisions of
    PyLPboxADMMsolver.solve_iter_fix, which tests/test_lp_fix_rule_gpu.py holds against the oracle: same fixes, same end state."""
    from lpbox_hip.lp import PyLPboxADMMsolver
    exe = build_driver(tmp_path)
    res, _ = run_driver(exe, GOLDEN, 1, 100, 500, 1500, -1)
    g = PyLPboxADMMsolver(0)
    g.data_root = GOLDEN
    g.read_File(1, 100, 500)
    g.solve_init()
    ret = g.solve_iter_fix(0, 1500)
    assert int(res["ret"]) == ret
    assert int(res["live"]) == g.get_n() and g.get_n() < 500
    assert float(res["objective"]) == -g.cal_Obj()
    assert int(res["iterations"]) == g.batch.counters()[0]
    assert int(res["ones"]) == int(g.get_x_sol().sum())


@pytest.mark.gpu
def test_cxx_driver_routes_an_oversize_instance(tmp_path):
    from lpbox_hip.lp import PyLPboxADMMsolver
    from lpbox_hip.synth import make_auction_like, write_instance_files
    exe = build_driver(tmp_path)
    P = make_auction_like(2300, 7)
    d = tmp_path / "instance" / "1000_2300"
    os.makedirs(d)
    write_instance_files(P, str(d / "instance_1_C.txt"), str(d / "instance_1_b.txt"))
    res, _ = run_driver(exe, tmp_path, 1, 1000, 2300, 400)
    assert int(res["large"]) == 1 and int(res["n"]) == 2300
    g = PyLPboxADMMsolver(0)
    g.data_root = str(tmp_path)
    g.write_files = False
    g.read_File(1, 1000, 2300)
    g.solve_init()
    ret = g.solve_iter(0, 400)
    assert g.large and int(res["ret"]) == ret
    assert float(res["objective"]) == -g.cal_Obj()
    assert int(res["iterations"]) == g.batch.counters()[0]
    assert int(res["infeasible"]) == g.check_infeasible_l2f()
    assert int(res["ones"]) == int(g.get_x_sol().sum())


def test_cxx_driver_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """CPU part: the class and its driver compile against the C-ABI header alone; without a HIP device the program reads the
    instance, then stops with the library's error text and a non-zero status (no CPU fallback)."""
    from lpbox_hip import _lib
    exe = build_driver(tmp_path)
    if _lib.load().lpbox_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu tests")
    p = subprocess.run([exe, "1", "100", "500"], env=dict(os.environ, LPBOX_DATA_ROOT=GOLDEN), capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "no HIP device" in p.stderr
    p = subprocess.run([exe, "1", "100", "501"], env=dict(os.environ, LPBOX_DATA_ROOT=GOLDEN), capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "cannot open" in p.stderr


@pytest.mark.gpu
def test_cxx_segmentation_class_equals_python_class(tmp_path):
    """The segmentation flavour of the C++ class through `seg_solve` (= the reference's image_segmentation.cpp main): JPEG read by the
    library, resize + costs + legacy loop on the GPU, result image written as PNG -- same energy, objective, iteration count, solution
    and output image as the Python class on the reference's sample image, and the reference's result line in xiter_all.csv."""
    from lpbox_hip.seg import PyLPboxADMMsolver as SegSolver
    exe = str(tmp_path / "seg_solve")
    subprocess.check_call(["make", "-s", "-C", SEG_CXX_DIR, "OUT=" + exe])
    res_dir = tmp_path / "result"
    os.makedirs(res_dir)
    env = dict(os.environ, LPBOX_SEG_DATA_ROOT=os.path.join(GOLDEN, "seg"), LPBOX_SEG_RESULT_ROOT=str(res_dir), LPBOX_SEG_XITER_ROOT=str(tmp_path / "nox"))
    p = subprocess.run([exe, "10000", "0", "0", "0", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "Object with node is created with three inputs!" in p.stdout and "Reshaped image size: 87 X 115 = 10005" in p.stdout
    res = dict(kv.split("=") for kv in re.search(r"^RESULT (.*)$", p.stdout, re.M).group(1).split())
    g = SegSolver(0, 10000, 0)
    g.data_root = os.path.join(GOLDEN, "seg")
    g.write_files = False
    g.solve_init()
    energy = g.solve_iter()
    assert int(res["energy"]) == energy and float(res["objective"]) == g.get_obj()
    assert int(res["n"]) == g.get_org_n() == 10005 and int(res["ones"]) == int(g.get_x_sol().sum())
    assert int(res["iterations"]) == g.counters()[0]
    from PIL import Image
    png = np.asarray(Image.open(res_dir / "output_0.png"))
    ref = str(tmp_path / "ref.png")
    g.save_img(ref)
    assert png.dtype == np.uint8 and np.array_equal(png, np.asarray(Image.open(ref)))
    line = open(res_dir / "xiter_all.csv").read().strip().split(",")
    assert int(line[0]) == 0 and int(line[3]) == g.stop()[1] and abs(float(line[1]) - g.debug_scalar("cur_obj")) < 1e-5
