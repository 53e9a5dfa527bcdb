#!/usr/bin/env python3
"""Checkpoints of the reference generator's random stream over ALL 2048 instances of a BASELINE batch (configs[3]: 2048 x j500/k2000, and the
same for j100/k500), so that rank r of a multi-GPU run can produce ITS 256 instances (numbers 256 r + 1 ... 256 r + 256 of the one
`RandomState(0)` stream, generate_instances.py:374, 393-396) without replaying the 256 r instances before them.

Runs ONLY in the authoring container (imports the reference's generator like make_lp_fixtures.py; about 2 h for j500/k2000 on one core).
Output lp_stream_<items>_<bids>_seed0.npz (data only):
  every      = 16                       a checkpoint before instance 0, 16, 32, ...
  key, pos   = MT19937 state at each checkpoint (numpy RandomState.get_state(): 624 words + position; no Gaussian is ever drawn)
  n, l, nnz  = sizes of every instance
  digest     = sha1 over (colptr int32, rowidx int32, price float64) of every instance: what lpbox_hip/synth.py's restatement must reproduce
usage: make_lp_stream_fixture.py items bids [count=2048]"""
import contextlib, hashlib, io, os, shutil, sys, tempfile, time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_lp_fixtures as M  # noqa: E402


def main():
    items, bids = int(sys.argv[1]), int(sys.argv[2])
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    every = 16
    gi = M.load_generator()
    rng = np.random.RandomState(0)
    tmp = tempfile.mkdtemp()
    keys, poss, ns, ls, nnzs, digests = [], [], [], [], [], []
    t0 = time.time()
    for i in range(count):
        if i % every == 0:
            st = rng.get_state()
            assert st[0] == "MT19937" and st[3] == 0
            keys.append(st[1].copy()); poss.append(st[2])
        prefix = os.path.join(tmp, "inst")
        with contextlib.redirect_stdout(io.StringIO()):
            gi.generate_cauctions(rng, prefix, n_items=items, n_bids=bids, add_item_prob=0.7)
        n, l, colptr, rowidx, price = M.read_instance(prefix)
        h = hashlib.sha1()
        h.update(np.ascontiguousarray(colptr, np.int32).tobytes()); h.update(np.ascontiguousarray(rowidx, np.int32).tobytes())
        h.update(np.ascontiguousarray(price, np.float64).tobytes())
        ns.append(n); ls.append(l); nnzs.append(len(rowidx)); digests.append(np.frombuffer(h.digest(), np.uint8))
        for suffix in ("_C.txt", "_b.txt", ".lp"):
            os.remove(prefix + suffix)
        if i % 64 == 63:
            print(f"{items}/{bids}: {i + 1} instances, {time.time() - t0:.0f} s", flush=True)
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, f"lp_stream_{items}_{bids}_seed0.npz"), every=np.int32(every), key=np.stack(keys).astype(np.uint32),
                        pos=np.array(poss, np.int32), n=np.array(ns, np.int32), l=np.array(ls, np.int32), nnz=np.array(nnzs, np.int32),
                        digest=np.stack(digests), n_items=np.int32(items), n_bids=np.int32(bids), seed=np.int32(0))


if __name__ == "__main__":
    main()
