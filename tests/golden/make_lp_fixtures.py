#!/usr/bin/env python3
"""Generate the LP (combinatorial-auction) instance fixtures under tests/golden/.

Runs ONLY in the authoring container: it imports the reference's own generator
  /root/reference/LinerProgramming/LinearProgramming/generate_data/generate_instances.py:137-360
(`generate_cauctions`, pure numpy) exactly the way its `__main__` does (:374, :393-396): one
`numpy.random.RandomState(seed)` stream shared by consecutive instances, `add_item_prob=0.7`.
The generator module does `import utilities` (-> pyscipopt, absent here) only for the argparse
type `valid_seed` used in `__main__`; an empty stand-in module satisfies that import.

Outputs (data only -- inputs of the solver; no reference source text is stored):
  lp_100_500_seed0.npz        first 256 instances j=100 items / k=500 bids   (BASELINE configs[0], [1])
  lp_500_2000_seed0.npz       first 256 instances j=500 / k=2000             (BASELINE configs[3]: one GPU's share of the 2048)
  lp_20_60_seed0.npz          first 8 tiny instances j=20 / k=60             (fast parity cases)
  instance/100_500/instance_{1,2}_{C,b}.txt   the generator's own text files (reader tests)

npz layout (instances concatenated): n[B], l[B], nnz[B], colptr (sum(n+1)) int32, rowidx (sum nnz) int32
(CSC of E, rows ascending inside a column), price (sum n) float64 (the `_b.txt` values; the solver uses b=-price).
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/LinerProgramming/LinearProgramming/generate_data"


def load_generator():
    stub = types.ModuleType("utilities")
    stub.valid_seed = int
    sys.modules["utilities"] = stub
    sys.path.insert(0, REF)
    import generate_instances  # noqa: E402

    return generate_instances


def read_instance(prefix):
    C = np.loadtxt(prefix + "_C.txt", delimiter=",", ndmin=2)
    price = np.loadtxt(prefix + "_b.txt", ndmin=1)
    rows = C[:, 0].astype(np.int64) - 1
    cols = C[:, 1].astype(np.int64) - 1
    assert np.all(C[:, 2] == 1.0)
    l = int(rows.max()) + 1          # readSparseMat: max_row (LPcpp:2426-2441)
    n = int(cols.max()) + 1
    assert n == price.shape[0]
    order = np.lexsort((rows, cols))  # column-major, rows ascending
    rows, cols = rows[order], cols[order]
    assert len(set(zip(rows.tolist(), cols.tolist()))) == len(rows)  # no duplicates
    colptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(colptr, cols + 1, 1)
    colptr = np.cumsum(colptr).astype(np.int32)
    return n, l, colptr, rows.astype(np.int32), price.astype(np.float64)


def make_batch(gi, n_items, n_bids, count, seed, keep_text=0, text_dir=None):
    rng = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    ns, ls, nnzs, colptrs, rowidxs, prices = [], [], [], [], [], []
    for i in range(count):
        prefix = os.path.join(tmp, f"instance_{i + 1}")
        with contextlib.redirect_stdout(io.StringIO()):
            gi.generate_cauctions(rng, prefix, n_items=n_items, n_bids=n_bids, add_item_prob=0.7)
        n, l, colptr, rowidx, price = read_instance(prefix)
        ns.append(n); ls.append(l); nnzs.append(len(rowidx))
        colptrs.append(colptr); rowidxs.append(rowidx); prices.append(price)
        if i < keep_text:
            os.makedirs(text_dir, exist_ok=True)
            for suffix in ("_C.txt", "_b.txt"):
                shutil.copy(prefix + suffix, os.path.join(text_dir, f"instance_{i + 1}{suffix}"))
        os.remove(prefix + ".lp")
    shutil.rmtree(tmp)
    return dict(
        n=np.array(ns, np.int32), l=np.array(ls, np.int32), nnz=np.array(nnzs, np.int32),
        colptr=np.concatenate(colptrs).astype(np.int32), rowidx=np.concatenate(rowidxs).astype(np.int32),
        price=np.concatenate(prices), n_items=np.int32(n_items), n_bids=np.int32(n_bids), seed=np.int32(seed),
    )


def main():
    gi = load_generator()
    only = sys.argv[1:]
    jobs = [
        ("lp_20_60_seed0.npz", 20, 60, 8, 0, None),
        ("lp_100_500_seed0.npz", 100, 500, 256, 2, os.path.join(HERE, "instance", "100_500")),
        ("lp_500_2000_seed0.npz", 500, 2000, 256, 0, None),
    ]
    for name, j, k, count, keep, tdir in jobs:
        if only and name not in only:
            continue
        d = make_batch(gi, j, k, count, 0, keep, tdir)
        if count > 64:                                 # large batches: row indices as uint16 (loaders widen them again)
            assert d["rowidx"].max() < 65536
            d["rowidx"] = d["rowidx"].astype(np.uint16)
        np.savez_compressed(os.path.join(HERE, name), **d)
        print(name, "instances", count, "n", d["n"][:4], "l", d["l"][:4], "nnz", d["nnz"][:4])


if __name__ == "__main__":
    main()
