"""The reference's combinatorial-auction instances, reproduced draw for draw (BASELINE configs[1], [3]).

The reference makes its LP instances with `generate_cauctions` (LP/generate_data/generate_instances.py:137-360: the "arbitrary" scheme of
Leyton-Brown, Pearson and Shoham, EC-00 section 4.3) from ONE `numpy.random.RandomState(seed)` shared by consecutive instances (:374,
:393-396).  BASELINE configs[3] is 2048 such instances, 256 per GPU; the reference's code does not travel to the GPU box, so this module
restates the generator: the same random draws in the same order and the same floating-point expressions, hence the same instances bit for
bit -- pinned by tests/test_auction_stream.py against instances written by the reference's own function (tests/golden/lp_*_seed0.npz) and
against the digests of all 2048 (tests/golden/lp_stream_*_seed0.npz, which also holds the generator state every 16 instances so that a
rank can start at ITS first instance without replaying the stream before it).

Output format: the instance dict of oracle.load_lp_batch -- n, l, colptr, rowidx (CSC of E, rows ascending), b = -price -- i.e. what
readFile (LPcpp:2446-2545) makes of the generator's `_C.txt` / `_b.txt`.
"""
import hashlib
import os

import numpy as np

# the reference's defaults (generate_instances.py:137-140); its __main__ overrides add_item_prob with 0.7 (:393-396)
VALUE_LO, VALUE_HI = 1, 100
VALUE_DEVIATION = 0.5
MAX_SUBSTITUTES = 5
ADDITIVITY = 0.2
BUDGET_FACTOR = 1.5
RESALE_FACTOR = 0.5


class _Bidder:
    """One bidder's view: interests, private values, and the draw of the next item of a bundle (:180-186, :203-205)."""

    def __init__(self, rng, common_values, compat):
        self.rng, self.compat = rng, compat
        self.interest = rng.rand(len(common_values))
        self.value = common_values + VALUE_HI * VALUE_DEVIATION * (2 * self.interest - 1)

    def first_item(self):
        return self.rng.choice(len(self.interest), p=self.interest / self.interest.sum())

    def next_item(self, taken):
        # `taken` is a 0/1 INTEGER vector and the reference indexes the compatibility matrix with it as such (:184): rows 0 and 1, one per
        # item, not the rows of the items taken.  The draw probabilities -- and with them the whole stream -- depend on that, so it stays.
        w = (1 - taken) * self.interest * self.compat[taken, :].mean(axis=0)
        w /= w.sum()
        return self.rng.choice(len(self.interest), p=w)

    def grow(self, taken, until):
        while taken.sum() < until:
            taken[self.next_item(taken)] = 1
        return np.nonzero(taken)[0]

    def price(self, items):
        return self.value[items].sum() + np.power(len(items), 1 + ADDITIVITY)


def auction_instance(rng, n_items=100, n_bids=500, add_item_prob=0.7):
    """The next instance of the stream `rng` (a numpy RandomState), as an instance dict."""
    common = VALUE_LO + (VALUE_HI - VALUE_LO) * rng.rand(n_items)                        # :190
    compat = np.triu(rng.rand(n_items, n_items), k=1)                                   # :193-195
    compat = compat + compat.transpose()
    compat = compat / compat.sum(1)
    bids = []                                                                           # (rows of E in this column, price)
    n_dummy = 0
    while len(bids) < n_bids:                                                           # one bidder per round (:201)
        who = _Bidder(rng, common, compat)
        taken = np.full(n_items, 0)
        taken[who.first_item()] = 1
        while rng.rand() < add_item_prob:                                               # :217-222
            if taken.sum() == n_items:
                break
            taken[who.next_item(taken)] = 1
        first = np.nonzero(taken)[0]
        first_price = who.price(first)
        if first_price < 0:                                                             # :232-235
            continue
        accepted = {frozenset(first): first_price}
        # one candidate substitute per item of the first bundle, same size, sharing that item (:241-259)
        cand = []
        for it in first:
            seed_mask = np.full(n_items, 0)
            seed_mask[it] = 1
            sub = who.grow(seed_mask, len(first))
            cand.append((sub, who.price(sub)))
        budget = BUDGET_FACTOR * first_price
        floor = RESALE_FACTOR * common[first].sum()
        for k in np.argsort([-p for _, p in cand]):                                     # dearest first (:264-265)
            sub, p = cand[k]
            if len(accepted) >= MAX_SUBSTITUTES + 1 or len(bids) + len(accepted) >= n_bids:
                break
            if p < 0 or p > budget or common[sub].sum() < floor or frozenset(sub) in accepted:
                continue
            accepted[frozenset(sub)] = p
        extra = []
        if len(accepted) > 2:                                                           # XOR constraint of the bidder: a dummy item (:293-297)
            extra = [n_items + n_dummy]
            n_dummy += 1
        for bundle, p in accepted.items():
            bids.append((sorted(int(i) for i in bundle) + extra, p))
    n = len(bids)
    colptr = np.zeros(n + 1, np.int32)
    colptr[1:] = np.cumsum([len(r) for r, _ in bids])
    rowidx = np.fromiter((i for r, _ in bids for i in r), np.int32, count=int(colptr[-1]))
    price = np.array([p for _, p in bids], np.float64)
    return dict(n=n, l=int(rowidx.max()) + 1, nnz=int(colptr[-1]), colptr=colptr, rowidx=rowidx, b=-1.0 * price)


def digest(inst):
    """sha1 over (colptr, rowidx, price) -- the fingerprint tests/golden/make_lp_stream_fixture.py stores for every instance of a stream."""
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(inst["colptr"], np.int32).tobytes())
    h.update(np.ascontiguousarray(inst["rowidx"], np.int32).tobytes())
    h.update(np.ascontiguousarray(-1.0 * np.asarray(inst["b"]), np.float64).tobytes())
    return np.frombuffer(h.digest(), np.uint8)


def default_stream_fixture(n_items, n_bids, seed=0):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    return os.path.join(root, "tests", "golden", f"lp_stream_{n_items}_{n_bids}_seed{seed}.npz")


def _block(args):
    key, pos, n_items, n_bids, count, skip = args
    rng = np.random.RandomState()
    rng.set_state(("MT19937", np.asarray(key, np.uint32), int(pos), 0, 0.0))
    out = [auction_instance(rng, n_items, n_bids) for _ in range(skip + count)]
    return out[skip:]


def stream_instances(n_items, n_bids, first, count, fixture=None, workers=1, check=True):
    """Instances first .. first + count - 1 (0-based) of the reference's seed-0 stream for this size, started from the nearest stored
    generator state (every 16 instances) and generated in `workers` processes.  check: compare every digest with the fixture's."""
    fx = np.load(fixture or default_stream_fixture(n_items, n_bids))
    every = int(fx["every"])
    if first < 0 or first + count > len(fx["n"]):
        raise ValueError("the stored stream holds instances 0 .. %d" % (len(fx["n"]) - 1))
    jobs = []
    i = first
    while i < first + count:
        blk = i // every
        skip = i - blk * every
        take = min(every - skip, first + count - i)
        jobs.append((fx["key"][blk], int(fx["pos"][blk]), n_items, n_bids, take, skip))
        i += take
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        # an executor, not mp.Pool: a worker that cannot start (e.g. a __main__ without a file under spawn) raises BrokenProcessPool
        # here; mp.Pool would respawn it for ever
        with ProcessPoolExecutor(min(workers, len(jobs)), mp_context=mp.get_context("spawn")) as pool:
            parts = list(pool.map(_block, jobs))
    else:
        parts = [_block(j) for j in jobs]
    out = [inst for p in parts for inst in p]
    if check:
        for k, inst in enumerate(out):
            if not np.array_equal(digest(inst), fx["digest"][first + k]):
                raise RuntimeError("instance %d of the %d/%d stream differs from the reference generator's" % (first + k, n_items, n_bids))
    return out
