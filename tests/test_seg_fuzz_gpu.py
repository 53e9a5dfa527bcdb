"""Differential fuzz of the segmentation flavour (tools/fuzz_seg.py, a fixed seed here): gray images of awkward shapes (two columns, primes,
wide, tall, tiny), several node budgets for the resize, noise / two-region / salt-and-pepper / CONSTANT content -- the last one makes the
reference's cost construction divide by a zero variance, every weight is NaN and the iterates are NaN on both sides, bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from fuzz_seg import check, random_image  # noqa: E402

pytestmark = pytest.mark.gpu


def test_awkward_images_bit_exact():
    from lpbox_hip._lib import LpboxError
    rs = np.random.RandomState(5)
    done, kinds, refused = 0, set(), 0
    for t in range(60):
        gray, nodes, kind = random_image(rs)
        try:
            ok, n = check(gray, nodes, legacy=kind != 2)
        except LpboxError as e:                        # an image that scales to one pixel in a direction is refused loudly: not a solve
            assert "too small" in str(e), str(e)
            refused += 1
            continue
        assert ok, (t, kind, gray.shape, nodes, n)
        done += 1; kinds.add(kind)
        if done >= 14:
            break
    assert done >= 14 and kinds == {0, 1, 2, 3}


def test_one_pixel_wide_images_are_refused_loudly():
    from lpbox_hip._lib import LpboxError
    from lpbox_hip.seg import PyLPboxADMMsolver
    g = PyLPboxADMMsolver(0, 37, 0)
    with pytest.raises(LpboxError):
        g.set_image(np.zeros((1, 37), np.uint8))
