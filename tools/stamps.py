import sys, os, ctypes as C
os.environ["LPBOX_LIB_VARIANT"]="stamps"
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np
from bench import load_instances, FIXTURE, FIXTURE_C4
from lpbox_hip.lp import LpBatch
c4=len(sys.argv)>1 and sys.argv[1]=="4"
insts=load_instances(FIXTURE_C4 if c4 else FIXTURE)[:256]
N=1000 if c4 else 2000
b=LpBatch(insts); b.solve_init(); b.solve_iter(0,N)
L=b._L; L.lpbox_debug_get_stamps.argtypes=[C.c_void_p,C.c_int,C.c_void_p]
names=["A y1y2+red","B y3,rhs,Ey1","C pcg setup+red3","D1 gxwrite+bar","D2 rows gather","D3 glwrite+bar","D4 cols+Mp","D5 blocksum1","D6 upd","D7 blocksum2","D8 p upd","E post-pcg","F duals+Ex","G blocksum5","H tail","looptop"]
order=[15,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14]
tot=np.zeros(16)
for i in range(0,256,32):
    out=(C.c_ulonglong*16)(); L.lpbox_debug_get_stamps(b._h,i,out); a=np.array(list(out),float)
    o,p=b.counters(i)
    tot+=a/ o
    if i==0: print("outer",o,"pcg",p, "cycles/outer", a.sum()/o)
tot/=8
labels=["A","B","C","D1","D2","D3","D4","D5","D6","D7","D8","E","F","G","H","top"]
for k in range(16): print("%-18s %8.0f cyc/outer  %5.1f%%"%(names[k] , tot[k], 100*tot[k]/tot.sum()))
print("total cyc/outer", tot.sum(), " kernel ms", b.kernel_time()[0])
