"""usage: python tools/rank_shards.py [2|4]
What the weak-scaling run of config 2 (or, with argument 4, BASELINE configs[3]: 8 x 256 j=500/k=2000) will see: the eight rank shards of the reference generator's seed-0 stream (rank r = draws
256 r .. 256 r + 255, lpbox_hip/auction.py), solved one after the other on ONE GPU.  A step of the N-rank run ends with its slowest rank."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')]
import numpy as np
import bench
from lpbox_hip import auction
from lpbox_hip.lp import LpBatch


def main():
    c4 = len(sys.argv) > 1 and sys.argv[1] == "4"
    items, bids, fixture = (500, 2000, bench.FIXTURE_C4) if c4 else (100, 500, bench.FIXTURE)
    ms_all, it_all = [], []
    for rank in range(8):
        shard = bench.load_instances(fixture) if rank == 0 else auction.stream_instances(items, bids, 256 * rank, 256, workers=16)
        b = LpBatch(shard); b.solve_init(); b.kernel_time(reset=True); b.solve_iter(0, 20000)
        it = np.array([b.counters(i)[0] for i in range(256)]); ms, _ = b.kernel_time()
        ms_all.append(ms); it_all.append(it.sum())
        print('rank', rank, 'iterations mean %.0f' % it.mean(), 'max', it.max(), 'p99 %.0f' % np.percentile(it, 99), 'ms %.1f' % ms,
              'capped at 2e4:', int((it >= 20000).sum()), 'objective mean %.2f' % np.mean([-b.cal_obj(i) for i in range(256)]), flush=True)
    for n in (1, 2, 4, 8):
        t = max(ms_all[:n]); v = sum(it_all[:n]) / t * 1e3
        print('%d ranks: step %.1f ms, %.2f M instance-iterations/s, efficiency vs rank 0 alone %.2f' % (n, t, v / 1e6, v / n / (it_all[0] / ms_all[0] * 1e3)))


if __name__ == "__main__":      # (the generator's worker pool re-imports this file: nothing may touch the GPU at import)
    main()
