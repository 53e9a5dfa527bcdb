"""On-disk formats the reference's solver writes and its trainers read back.  Pure Python, no device:
the wrappers in lp.py / seg.py call these with iterates they fetched from the GPU, and tests/golden/make_trainer_fixtures.py feeds
the files written here to the reference's OWN readers (LP/trainer.py `readFile` :32-48, `get_lpbox_info` :189-201; SEG/trainer.py
`readFile` :36-51, `get_lpbox_info` :239-250), which pins the formats.

LPcpp = LinerProgramming/LinearProgramming/cython_solver/LPboxADMMsolver.cpp, SEGcpp = Segmentation/Segmentation/cython/src/LPboxADMMsolver.cpp.
"""
import numpy as np


def write_xiters_csv(path, X, first_iter=0, mode="w"):
    """One row per iteration: "Iter<k>,x0,...,x{n-1}" with k = first_iter + r + 1 and every value printed with C's "%lf"
    (LPcpp:903-909 / :940-946 / :986-992; SEGcpp:1270-1277).  X: (iterations, n)."""
    X = np.asarray(X, np.float64)
    with open(path, mode) as f:                                     # the reference opens with "w+" (LPcpp:778, SEGcpp:1212)
        for r in range(X.shape[0]):
            f.write("Iter%d," % (first_iter + r + 1) + ",".join(map("%f".__mod__, X[r])) + "\n")


def write_iteration_log(path, records, stopped_in=None, mode="w"):
    """The per-iteration text log of ADMM_lp_iters (`does_log`, LPh:148; LPcpp:789, :898-901, :1013-1067; path LPcpp:2496, opened "w+"
    at :772).  records: (rows, 12) array from lpbox_get_log -- PCG iterations, |x_sol|, |y1|, |y2|, |y3|, |z1|, |z2|, |z4|, dou_obj,
    bin_obj, seconds, iteration.  stopped_in: the iteration whose stop test ended the loop (its header line is written, its block is
    not -- the reference breaks before the log block)."""
    with open(path, mode) as f:
        for r in np.asarray(records, np.float64).reshape(-1, 12):
            it = int(r[11])
            f.write("Iteration: %d\n" % it)
            f.write("Conjugate gradient stops after %d iterations\n" % int(r[0]))
            f.write("norm of x_sol: %.9f\nnorm of y1: %.9f\nnorm of y2: %.9f\nnorm of y3: %.9f\n" % (r[1], r[2], r[3], r[4]))
            f.write("norm of z1: %.9f\nnorm of z2: %.9f\nFor z4\nnorm of z4: %.9f\n" % (r[5], r[6], r[7]))
            f.write("LongkangIter: %d;  x_sol: %f; dou_obj:%f; bin_obj: %f\n" % (it + 1, r[1], r[8], r[9]))
            f.write("Time elapsed: %fs\n" % r[10])
            f.write("-------------------------------------------------\n")
        if stopped_in is not None:
            f.write("Iteration: %d\n" % int(stopped_in))


def append_allres(path, idx, obj, iters, secs):
    """LP result line "%d,%f,%d,%f" = instance, -objective, iterations, seconds appended to allres.csv (LPcpp:1081)."""
    with open(path, "a") as f:
        f.write("%d,%f,%d,%f\n" % (idx, obj, iters, secs))


def append_xiter_all(path, problem, obj, energy, iters, secs):
    """SEG result line "%d,%f,%f,%d,%f" = problem, objective, energy (= objective + c), iterations, seconds appended to
    xiter_all.csv (SEGcpp:1376)."""
    with open(path, "a") as f:
        f.write("%d,%f,%f,%d,%f\n" % (problem, obj, energy, iters, secs))


def read_xiters_csv(path):
    """What the trainers' `readFile` returns: (n, iterations) float64 -- the first field of every line ("Iter<k>") dropped,
    the rest parsed as floats, transposed (LP/trainer.py:32-48)."""
    rows = []
    with open(path) as f:
        for line in f:
            rows.append([float(v) for v in line.split(",")[1:]])
    return np.array(rows).T


def read_results_csv(path):
    """`get_lpbox_info` (LP/trainer.py:189-201, SEG/trainer.py:239-250): list of float lists, one per line."""
    with open(path) as f:
        return [[float(v) for v in line.split(",")] for line in f if line.strip()]


def labels_from_iterates(dataset):
    """`getLabel` (LP/trainer.py:80-89): the final iterate of every variable rounded at 0.5 -> (n, 1) of {0., 1.}."""
    dataset = np.asarray(dataset)
    return (dataset[:, -1] >= 0.5).astype(np.float64)[:, None]


def window_subset(data, idx, ws):
    """`getSubset` (LP/trainer.py:91-98): iterates of window idx (1-based) = columns (idx-1)*ws .. idx*ws-1."""
    return np.asarray(data)[:, (idx - 1) * ws: idx * ws]
