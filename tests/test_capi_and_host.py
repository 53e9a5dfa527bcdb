"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/lpbox_hip.h declares, validates its
arguments on the host, and refuses to compute without a HIP device (no fallback).  No compute call needs a GPU here."""
import inspect
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN, lp_instances

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "lpbox_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lpbox_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from lpbox_hip import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"liblpbox_hip.so does not export {n}"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and header disagree"
    assert _lib.load().lpbox_version().startswith(b"lpbox_hip")


def test_python_class_mirrors_the_pyx_surface():
    """Method names and arities of LinerProgramming/LinearProgramming/cython_solver/lpbox.pyx:7-76."""
    from LinearProgramming.cython_solver import lpbox
    want = {"read_File": 3, "solve_init": 0, "solve_iter": 2, "cal_Obj": 0, "get_curBinObj": 0, "solve_iter_l2f": 4,
            "get_x_iters_1d": 1, "get_x_iters_2d": 1, "get_n": 0, "get_iter": 0, "get_x_sol": 1, "get_final_x_sol": 1,
            "check_infeasible_lpbox": 0, "check_infeasible_l2f": 0}
    for name, nargs in want.items():
        fn = getattr(lpbox.PyLPboxADMMsolver, name)
        params = [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]
        assert len(params) == nargs, name
    with pytest.raises(TypeError):
        lpbox.PyLPboxADMMsolver(5, 0.5)     # the 2-argument __cinit__ (pyx:10-11) is shadowed by the 1-argument one


def test_host_side_validation_and_no_cpu_fallback():
    from lpbox_hip import _lib
    from lpbox_hip.lp import LpBatch, LpboxError, PyLPboxADMMsolver
    I = lp_instances("lp_20_60_seed0.npz")[0]
    b = LpBatch(batch=1)
    with pytest.raises(LpboxError, match="row index"):
        bad = I["rowidx"].copy(); bad[0] = 10 ** 6
        b.set_problem(0, I["n"], I["l"], I["colptr"], bad, I["b"])
    with pytest.raises(LpboxError, match="!= 1"):
        b.set_problem(0, I["n"], I["l"], I["colptr"], I["rowidx"], I["b"], vals=2 * np.ones(len(I["rowidx"])))
    with pytest.raises(LpboxError, match="range"):
        b.set_problem(3, I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    b.set_problem(0, I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    assert b.get_org_n(0) == I["n"] and b.get_l(0) == I["l"]
    # a malformed CSC (an interior colptr far beyond nnz) is rejected before rowidx is read through it, and a rejected call leaves
    # the instance that was set before untouched
    J = lp_instances("lp_20_60_seed0.npz")[1]
    badp = J["colptr"].copy(); badp[1] = 10 ** 6
    with pytest.raises(LpboxError, match="colptr"):
        b.set_problem(0, J["n"], J["l"], badp, J["rowidx"], J["b"])
    assert b.get_org_n(0) == I["n"] and b.get_l(0) == I["l"]
    with pytest.raises(LpboxError, match="solve_init"):
        b.solve_iter(0, 10)
    s = PyLPboxADMMsolver(0)
    with pytest.raises(LpboxError, match="cannot open"):
        s.data_root = "/nonexistent"
        s.read_File(1, 100, 500)
    s.data_root = GOLDEN
    s.read_File(1, 100, 500)                  # <root>/instance/100_500/instance_1_{C,b}.txt (LPcpp:2492-2494)
    assert s.batch.get_org_n(0) == 500 and s.batch.get_l(0) == 189
    if _lib.load().lpbox_device_count() == 0:
        with pytest.raises(LpboxError, match="no HIP device"):
            s.solve_init()                    # the product path fails loudly without its GPU


def test_instance_comes_back_out_of_the_handle():
    """lpbox_get_problem_lp returns what set_problem / readFile left in the handle (the hand-over of an oversize instance to the
    large-instance path starts from it): arrays equal to the input, default f = ones, the file route equal to the fixture."""
    from lpbox_hip.lp import LpBatch, PyLPboxADMMsolver
    I = lp_instances("lp_100_500_seed0.npz")[0]
    b = LpBatch(batch=2)
    b.set_problem(1, I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    P = b.get_problem(1)
    assert (P["n"], P["l"]) == (I["n"], I["l"])
    assert np.array_equal(P["colptr"], I["colptr"]) and np.array_equal(P["rowidx"], I["rowidx"])
    assert np.array_equal(P["b"], I["b"]) and np.array_equal(P["f"], np.ones(I["l"]))
    from lpbox_hip.lp import LpboxError
    with pytest.raises(LpboxError, match="no problem"):
        b.get_problem(0)
    s = PyLPboxADMMsolver(0)
    s.data_root = GOLDEN
    s.read_File(1, 100, 500)
    Q = s.batch.get_problem(0)
    assert np.array_equal(Q["colptr"], I["colptr"]) and np.array_equal(Q["rowidx"], I["rowidx"]) and np.array_equal(Q["b"], I["b"])


def test_jpeg_reader_equals_libjpeg_luminance(tmp_path):
    """lpbox_read_jpeg_gray against PIL's grayscale draft (= libjpeg's JCS_GRAYSCALE output, what cv::imread(path, 0) yields) on the
    reference's sample images: every pixel equal.  Files it does not read are refused with a reason, not approximated."""
    PIL = pytest.importorskip("PIL.Image")
    from lpbox_hip.seg import load_gray
    from lpbox_hip.lp import LpboxError
    from lpbox_hip import _lib
    import ctypes as C
    for name in ("0.jpg", "7.jpg"):
        path = os.path.join(GOLDEN, "seg", name)
        im = PIL.open(path)
        im.draft("L", im.size)
        ref = np.asarray(im.convert("L"), dtype=np.uint8)
        got = load_gray(path)
        assert got.shape == ref.shape and np.array_equal(got, ref), name
    L = _lib.load()
    r, c = C.c_int(), C.c_int()
    prog = str(tmp_path / "progressive.jpg")
    PIL.fromarray(np.arange(64 * 48, dtype=np.uint8).reshape(48, 64)).save(prog, progressive=True)
    assert L.lpbox_read_jpeg_gray(prog.encode(), None, 0, C.byref(r), C.byref(c)) == -2 and b"progressive" in L.lpbox_last_error()
    assert np.array_equal(load_gray(prog), np.asarray(PIL.open(prog).convert("L")))          # ... and load_gray falls back to PIL for it
    with pytest.raises(LpboxError, match="cannot open"):
        load_gray(str(tmp_path / "missing.jpg"))
    # a grayscale JPEG with restart markers and odd dimensions
    odd = str(tmp_path / "odd.jpg")
    rng = np.random.RandomState(0)
    PIL.fromarray((rng.rand(37, 53) * 255).astype(np.uint8)).save(odd, quality=83, restart_marker_blocks=3)
    im = PIL.open(odd)
    assert np.array_equal(load_gray(odd), np.asarray(im.convert("L"), dtype=np.uint8))


def _patch_sof_sampling(data, comp, hv):
    """Return the JPEG bytes with the sampling-factor byte of frame component `comp` replaced by `hv` (e.g. 0x22)."""
    m = bytearray(data)
    i = 2
    while i + 4 <= len(m):
        assert m[i] == 0xFF
        mk, L = m[i + 1], (m[i + 2] << 8) | m[i + 3]
        if mk in (0xC0, 0xC1):
            m[i + 4 + 6 + 3 * comp + 1] = hv
            return bytes(m)
        i += 2 + L
    raise AssertionError("no SOF")


def test_jpeg_sampling_factor_corner_cases(tmp_path):
    """(1) A one-component JPEG whose frame header states sampling factors other than 1x1 is still one 8x8 block per MCU in raster order
    (T.81 A.2.2; libjpeg ignores the factors when comps_in_scan == 1): the reader must decode it pixel for pixel like libjpeg (PIL).
    (2) A colour file whose luminance is NOT the most finely sampled component (Y 1x1, Cb 2x2) would need libjpeg's upsampling of Y:
    refused with a reason -- in the previous round this file read past the end of the luminance plane."""
    PIL = pytest.importorskip("PIL.Image")
    from lpbox_hip.seg import load_gray
    from lpbox_hip import _lib
    import ctypes as C
    L = _lib.load()
    rng = np.random.RandomState(3)
    gray = str(tmp_path / "gray.jpg")
    PIL.fromarray((rng.rand(45, 70) * 255).astype(np.uint8)).save(gray, quality=90)
    for hv in (0x22, 0x21, 0x14):
        patched = str(tmp_path / ("gray_%02x.jpg" % hv))
        open(patched, "wb").write(_patch_sof_sampling(open(gray, "rb").read(), 0, hv))
        ref = np.asarray(PIL.open(patched).convert("L"), dtype=np.uint8)
        assert np.array_equal(ref, np.asarray(PIL.open(gray).convert("L"), dtype=np.uint8))      # libjpeg does ignore the factors
        r, c = C.c_int(), C.c_int()
        assert L.lpbox_read_jpeg_gray(patched.encode(), None, 0, C.byref(r), C.byref(c)) == 0 and (r.value, c.value) == ref.shape
        out = np.zeros(ref.shape, np.uint8)
        assert L.lpbox_read_jpeg_gray(patched.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(c)) == 0
        assert np.array_equal(out, ref), hex(hv)
    src = open(os.path.join(GOLDEN, "seg", "0.jpg"), "rb").read()
    bad = _patch_sof_sampling(_patch_sof_sampling(src, 0, 0x11), 1, 0x22)
    path = str(tmp_path / "y_subsampled.jpg")
    open(path, "wb").write(bad)
    r, c = C.c_int(), C.c_int()
    assert L.lpbox_read_jpeg_gray(path.encode(), None, 0, C.byref(r), C.byref(c)) == 0           # size query: header only
    out = np.zeros(r.value * c.value, np.uint8)
    assert L.lpbox_read_jpeg_gray(path.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(c)) == -2
    assert b"luminance is subsampled" in L.lpbox_last_error()


def test_jpeg_reader_survives_damaged_files(tmp_path):
    """Truncated files and files with flipped bytes are either refused or decoded to SOME image -- never a crash or an out-of-bounds
    access (the same mutations ran 6 000 times under AddressSanitizer + UBSan on the CPU build of csrc/lpbox_jpeg_host.cpp while it was
    written: clean)."""
    from lpbox_hip import _lib
    import ctypes as C
    L = _lib.load()
    src = open(os.path.join(GOLDEN, "seg", "0.jpg"), "rb").read()
    rng = np.random.RandomState(1)
    path = str(tmp_path / "m.jpg")
    decoded = refused = 0
    for trial in range(150):
        m = bytearray(src)
        mode = trial % 3
        if mode == 0:
            m = m[: rng.randint(0, len(m))]
        else:
            for _ in range(rng.randint(1, 8)):
                m[rng.randint(0, 700 if mode == 1 else len(m))] = rng.randint(0, 256)
        open(path, "wb").write(bytes(m))
        r, c = C.c_int(), C.c_int()
        rc = L.lpbox_read_jpeg_gray(path.encode(), None, 0, C.byref(r), C.byref(c))
        if rc == 0 and 0 < r.value * c.value < 4_000_000:
            out = np.zeros(r.value * c.value, np.uint8)
            rc = L.lpbox_read_jpeg_gray(path.encode(), out.ctypes.data_as(C.c_void_p), out.size, C.byref(r), C.byref(c))
        if rc == 0:
            decoded += 1
        else:
            refused += 1
            assert L.lpbox_last_error()
    assert decoded > 10 and refused > 10


def test_shard_range_partitions():
    from lpbox_hip.dist import shard_range
    for total in (0, 1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
