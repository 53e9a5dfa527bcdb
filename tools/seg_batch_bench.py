"""The reference's segmentation workload (image_segmentation.cpp:24-29: images 0..99 at 1e4 nodes, one after the other) as ONE batched
launch chain vs one solve at a time.  100 distinct problems are cut from the two committed sample images (random windows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'accelerated-lpbox-admm_amd')); sys.path.insert(0, ROOT)
import numpy as np
from lpbox_hip.seg import PyLPboxADMMsolver, load_gray, solve_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nodes = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000
src = [load_gray(os.path.join(ROOT, 'tests', 'golden', 'seg', f)) for f in ('0.jpg', '7.jpg')]
rs = np.random.RandomState(0)
def make(k):
    g = src[k % 2]
    h, w = g.shape
    hh, ww = rs.randint(h // 2, h + 1), rs.randint(w // 2, w + 1)
    y, x = rs.randint(0, h - hh + 1), rs.randint(0, w - ww + 1)
    s = PyLPboxADMMsolver(0, nodes, k); s.write_files = False; s.set_image(np.ascontiguousarray(g[y:y + hh, x:x + ww])); return s
ss = [make(k) for k in range(B)]
for s in ss[:2]: s.solve_init(); s.solve_iter()                     # warm up
t = time.perf_counter()
e1 = []
for s in ss: s.solve_init(); e1.append(s.solve_iter())
t1 = time.perf_counter() - t
it1 = sum(s.counters()[0] for s in ss)
solve_batch(ss[:2])
reps = []
for _ in range(int(os.environ.get("SEG_BATCH_REPS", "5"))):          # the batched chain is host-paced: several runs, the fastest and the median reported
    t = time.perf_counter(); e2 = solve_batch(ss); reps.append(time.perf_counter() - t)
t2 = min(reps)
it2 = sum(s.counters()[0] for s in ss)
assert e1 == e2 and it1 == it2
print(f"{B} problems at {nodes} nodes ({it1} outer iterations in all): one at a time {t1*1e3:.0f} ms ({B/t1:.1f} images/s, {t1/B*1e3:.1f} ms each); "
      f"batched {t2*1e3:.0f} ms ({B/t2:.1f} images/s; median of {len(reps)} runs {sorted(reps)[len(reps)//2]*1e3:.0f} ms) -> {t1/t2:.1f}x")
