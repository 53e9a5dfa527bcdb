import sys, os
sys.path[:0]=['/root/repo','/root/repo/accelerated-lpbox-admm_amd']
import numpy as np, bench
from lpbox_hip.lp import LpBatch
insts=bench.load_instances(bench.FIXTURE)
for rank in range(8):
    shard=insts if rank==0 else [bench.relabel(I,1000*rank+i) for i,I in enumerate(insts)]
    b=LpBatch(shard); b.solve_init(); b.kernel_time(reset=True); b.solve_iter(0,20000)
    it=np.array([b.counters(i)[0] for i in range(256)]); ms,_=b.kernel_time()
    print('rank',rank,'mean',it.mean(),'max',it.max(),'p99',np.percentile(it,99),'ms %.1f'%ms, 'objective mean %.2f'%np.mean([-b.cal_obj(i) for i in range(256)]))
print("--- which instance is slowest ---")
for rank in (1,2,3,5,0):
    shard=insts if rank==0 else [bench.relabel(I,1000*rank+i) for i,I in enumerate(insts)]
    b=LpBatch(shard); b.solve_init(); b.solve_iter(0,20000)
    it=np.array([b.counters(i)[0] for i in range(256)])
    top=np.argsort(it)[::-1][:3]
    print('rank',rank,'top3',[(int(i),int(it[i]),b.stop(int(i))[0]) for i in top])
