"""The restated instance generator (lpbox_hip/auction.py) against the reference's own: instances written by the reference's
generate_cauctions (tests/golden/lp_*_seed0.npz, made by make_lp_fixtures.py) and the digests / generator states of the whole 2048-instance
streams (tests/golden/lp_stream_*_seed0.npz, made by make_lp_stream_fixture.py from the same function).  Bit for bit: these are integer
index sets and prices printed with repr()."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import oracle as O
from lpbox_hip import auction as A


def same(a, b):
    return a["n"] == b["n"] and a["l"] == b["l"] and all(np.array_equal(a[k], b[k]) for k in ("colptr", "rowidx", "b"))


def test_first_instances_equal_the_reference_generators():
    ref = O.load_lp_batch(os.path.join(GOLDEN, "lp_100_500_seed0.npz"))
    rng = np.random.RandomState(0)
    for i in range(24):
        assert same(A.auction_instance(rng, 100, 500), ref[i]), i
    tiny = O.load_lp_batch(os.path.join(GOLDEN, "lp_20_60_seed0.npz"))
    rng = np.random.RandomState(0)
    for i in range(len(tiny)):
        assert same(A.auction_instance(rng, 20, 60), tiny[i]), i


def test_stream_fixture_agrees_with_the_instance_fixture():
    """Two files written from the reference generator at different times: the digests of the stream's first 256 instances are those of
    the instances stored in full."""
    ref = O.load_lp_batch(os.path.join(GOLDEN, "lp_100_500_seed0.npz"))
    fx = np.load(A.default_stream_fixture(100, 500))
    assert len(fx["n"]) == 2048 and int(fx["every"]) == 16 and fx["key"].shape == (128, 624)
    for i in (0, 1, 100, 255):
        assert np.array_equal(A.digest(ref[i]), fx["digest"][i]) and (fx["n"][i], fx["l"][i], fx["nnz"][i]) == (ref[i]["n"], ref[i]["l"], len(ref[i]["rowidx"]))


def test_a_rank_shard_starts_from_a_stored_generator_state():
    """Rank 4's first instances (1024 ...) and a range that starts inside a block and crosses into the next one: every digest equals
    the reference generator's (stream_instances checks them; the comparison is repeated here)."""
    fx = np.load(A.default_stream_fixture(100, 500))
    for first, count in ((1024, 6), (1003, 20)):
        got = A.stream_instances(100, 500, first, count)
        assert len(got) == count
        for k, inst in enumerate(got):
            assert np.array_equal(A.digest(inst), fx["digest"][first + k])
            assert (inst["n"], inst["l"], len(inst["rowidx"])) == (fx["n"][first + k], fx["l"][first + k], fx["nnz"][first + k])
    # the same instance reached from the start of the stream: the stored states ARE the stream's
    rng = np.random.RandomState(0)
    for _ in range(17):
        last = A.auction_instance(rng, 100, 500)
    assert same(last, A.stream_instances(100, 500, 16, 1)[0])
    with pytest.raises(ValueError):
        A.stream_instances(100, 500, 2040, 16)


def test_worker_pool_gives_the_same_shard():
    a = A.stream_instances(100, 500, 512, 40, workers=1)
    b = A.stream_instances(100, 500, 512, 40, workers=3)
    assert all(same(x, y) for x, y in zip(a, b))


def test_config4_size_first_instance_and_a_checkpoint():
    ref = O.load_lp_batch(os.path.join(GOLDEN, "lp_500_2000_seed0.npz"))
    rng = np.random.RandomState(0)
    assert same(A.auction_instance(rng, 500, 2000), ref[0])
    path = A.default_stream_fixture(500, 2000)
    if not os.path.exists(path):
        pytest.skip("tests/golden/lp_stream_500_2000_seed0.npz not generated yet (make_lp_stream_fixture.py 500 2000: two hours)")
    fx = np.load(path)
    assert len(fx["n"]) == 2048
    assert np.array_equal(A.digest(ref[255]), fx["digest"][255])
    inst = A.stream_instances(500, 2000, 1792, 1)[0]           # rank 7's first instance
    assert np.array_equal(A.digest(inst), fx["digest"][1792])
