import sys, os, time
sys.path[:0]=['.','accelerated-lpbox-admm_amd']
from lpbox_hip.big import BigLp
from lpbox_hip.synth import make_auction_like
n=int(float(sys.argv[1]))
P=make_auction_like(n,0)
g=BigLp(P); g.solve_init()
for w in range(2):
    l0=g.scalar("launches"); t=time.time(); g.solve_iter(20*w,20*w+20); dt=time.time()-t
    print("fold",g.scalar("folded_reductions"),"groups",g.scalar("groups"),"chunk",g.scalar("chunk"),"outer",g.scalar("outer_total"),"pcg",g.scalar("pcg_total"),"launches/iter",(g.scalar("launches")-l0)/20,"ms/iter",dt*50)
