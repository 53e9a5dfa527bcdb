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
