"""Differential fuzz of the large-instance LP path (tools/fuzz_big.py, a fixed seed here): empty rows, rows with thousands of entries
(runs inside a column slice longer than the unrolled six), one-entry and duplicate columns, 50 ... 33 000 variables, plain CSR and two slice
widths; plain and early-fixing windows bit for bit against the oracle in that path's order."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from fuzz_big import check, random_instance  # noqa: E402

pytestmark = pytest.mark.gpu


def test_odd_structures_on_the_large_route_bit_exact():
    rs = np.random.RandomState(3)
    kinds = set()
    for t in range(12):
        I, kind = random_instance(rs)
        kinds.add(kind)
        assert check(I, [None, 8, 64][t % 3]), (t, kind, I["n"], I["l"])
    assert len(kinds) >= 4


@pytest.mark.parametrize("case,world,mode", [(16, 3, "lean"), (16, 2, "reference"), (2, 3, "reference"), (2, 2, "lean")])
def test_odd_structures_sharded_over_ranks_bit_exact(case, world, mode):
    """tools/fuzz_big_ranks.py: the same odd instances variable-sharded over W ranks on the one GPU (callback transport over gloo) -- case
    16: 50 variables and 6 rows over 3 ranks (17 variables and a row block of 2 per rank); case 2: rows with thousands of entries -- in the
    reference arithmetic and in the opt-in comm-lean PCG, against the oracle's rank model."""
    from fuzz_big_ranks import run_case
    ok, I, kind = run_case(0, case, world, mode)
    assert ok, (case, world, mode, I["n"], I["l"])
