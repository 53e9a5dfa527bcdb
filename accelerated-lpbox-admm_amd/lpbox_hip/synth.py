"""Synthetic combinatorial-auction-like LP instances of arbitrary size (BASELINE config 5): the reference's
generator (generate_instances.py) cannot reach n = 1e6 (dense n_items^2 compatibility matrix, per-bid Python loops), so this
build-side generator matches its statistics instead: every bid (column) holds 1 + Geometric items out of ~0.3 n distinct
items, about 60 % of the bids additionally share a dummy item with 2-5 neighbouring bids, l ~ 0.43 n, every stored value 1,
prices ~ sum of the item values with +-50 % bidder noise plus a super-additive term."""
import numpy as np


def make_auction_like(n, seed=0, item_frac=0.3, add_prob=0.78, max_items=14):
    rng = np.random.RandomState(seed)
    n_items = max(8, int(item_frac * n))
    values = 1.0 + 99.0 * rng.rand(n_items)
    size = 1 + np.minimum(rng.geometric(1.0 - add_prob, n) - 1, max_items - 1)
    # items of every bid: distinct random items (draw with replacement, then drop duplicates per bid)
    tot = int(size.sum())
    col = np.repeat(np.arange(n), size)
    item = rng.randint(0, n_items, tot)
    key = np.unique(col.astype(np.int64) * n_items + item)
    col, item = (key // n_items).astype(np.int64), (key % n_items).astype(np.int64)
    # dummy items: consecutive groups of 3-6 bids share one extra row (the XOR constraint of a bidder)
    rows_extra, cols_extra = [], []
    j, d = 0, n_items
    grp = rng.randint(3, 7, n // 3 + 1)
    has = rng.rand(n // 3 + 1) < 0.6
    g = 0
    while j < n:
        k = int(grp[g])
        if has[g] and j + k <= n:
            rows_extra.append(np.full(k, d)); cols_extra.append(np.arange(j, j + k)); d += 1
        j += k; g += 1
    if rows_extra:
        item = np.concatenate([item, np.concatenate(rows_extra)]); col = np.concatenate([col, np.concatenate(cols_extra)])
    # drop empty rows, renumber
    used = np.unique(item)
    remap = -np.ones(d, np.int64); remap[used] = np.arange(len(used))
    item = remap[item]
    l = len(used)
    order = np.lexsort((item, col))
    item, col = item[order], col[order]
    colptr = np.zeros(n + 1, np.int64)
    np.add.at(colptr, col + 1, 1)
    colptr = np.cumsum(colptr)
    real = item < (remap[:n_items] >= 0).sum()
    inv = np.full(l, -1, np.int64)
    inv[remap[used[used < n_items]]] = used[used < n_items]
    val_row = np.where(inv >= 0, values[np.maximum(inv, 0)], 0.0)
    noise = 0.5 + rng.rand(len(item))
    price = np.zeros(n)
    np.add.at(price, col, val_row[item] * noise * real)
    cnt = np.diff(colptr)
    price += np.power(cnt, 1.2)
    return dict(n=int(n), l=int(l), colptr=colptr.astype(np.int32), rowidx=item.astype(np.int32), b=-price)


def write_instance_files(inst, path_C, path_b):
    """An instance dict (n, l, colptr, rowidx, b) in the reference's on-disk format (generate_instances.py:339-359, read back by
    readSparseMat / readDenseVec, LPcpp:2407-2444): `<row>,<col>,1` per entry, 1-based, ROW-major with ascending columns; one price
    per line.  `b` holds the negated prices (LPcpp:2520), so the file gets -b, printed with repr() like the generator's f-string."""
    n, l = int(inst["n"]), int(inst["l"])
    colptr, rowidx = np.asarray(inst["colptr"]), np.asarray(inst["rowidx"])
    cols = np.repeat(np.arange(n), np.diff(colptr))
    order = np.lexsort((cols, rowidx))                    # by row, then by column
    with open(path_C, "w") as f:
        for r, c in zip(rowidx[order], cols[order]):
            f.write("%d,%d,1\n" % (r + 1, c + 1))
    with open(path_b, "w") as f:
        for v in np.asarray(inst["b"], np.float64):
            f.write("%r\n" % float(-v))
    return l
