"""Where the time of the product loop goes (held-out 128 instances, policy trained for 30 epochs by tools/train_policy.py recipe)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np, torch
from bench import load_instances, FIXTURE
from lpbox_hip.lp import LpBatch
from lpbox_hip.policy import FusedEarlyFixPolicy
from lpbox_hip.train import TrainablePolicy, collect_training_data, train
insts = load_instances(FIXTURE)
torch.manual_seed(0)
hist, labels, _ = collect_training_data(insts[:100])
net = TrainablePolicy(20).cuda(); train(net, hist, labels, epochs=30, lr=1e-4)
pol = FusedEarlyFixPolicy(net.state_dict(), tokens=20)
test = insts[128:256]; B = len(test)
for rep in range(2):
    b = LpBatch(test); b.solve_init(); b.kernel_time(reset=True)
    ws = 100; nmax = 500
    vecs = np.zeros((B, nmax)); nums = np.zeros(B, np.int32); done = np.zeros(B, bool)
    T = dict(set_active=0.0, l2f_call=0.0, xiters=0.0, getn=0.0, offsets=0.0, policy=0.0, d2h=0.0, host=0.0); wins = 0; act_hist = []
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for w in range(100):
        t0 = time.perf_counter(); b.set_active(~done); t1 = time.perf_counter()
        rets = b.solve_iter_l2f(ws * w, ws * (w + 1), vecs, nums); t2 = time.perf_counter()
        wins += 1; done |= rets != 0; act_hist.append(int((~done).sum()))
        if done.all(): T["set_active"] += t1 - t0; T["l2f_call"] += t2 - t1; break
        flat, stride = b.x_iters_torch(ws); t3 = time.perf_counter()
        act = np.flatnonzero(~done); rows = [b.get_n(int(i)) for i in act]; t4 = time.perf_counter()
        r = np.asarray(rows, np.int64); first = np.repeat(np.cumsum(r) - r, r)
        off = np.repeat(act.astype(np.int64) * stride, r) + (np.arange(int(r.sum()), dtype=np.int64) - first) * ws
        offd = torch.from_numpy(off).cuda(); t5 = time.perf_counter()
        sig = pol.scores_from_xiters(flat, offd, 5); torch.cuda.synchronize(); t6 = time.perf_counter()
        vec = torch.where(sig > 0.9, 1.0, torch.where(sig < 0.1, 0.0, -1.0)).to(torch.float64).cpu().numpy(); t7 = time.perf_counter()
        nums[:] = 0; o = 0
        for i, rr in zip(act.tolist(), rows):
            v = vec[o:o + rr]; o += rr; k = int(np.count_nonzero(v != -1))
            if k > 10: vecs[i, :rr] = v; nums[i] = k
        t8 = time.perf_counter()
        for k_, d_ in (("set_active", t1 - t0), ("l2f_call", t2 - t1), ("xiters", t3 - t2), ("getn", t4 - t3), ("offsets", t5 - t4), ("policy", t6 - t5), ("d2h", t7 - t6), ("host", t8 - t7)): T[k_] += d_
    torch.cuda.synchronize(); tot = time.perf_counter() - t_all
    kms, nl = b.kernel_time()
    print("rep %d: total %.1f ms, %d windows; kernel (HIP events) %.1f ms in %d launches; " % (rep, tot * 1e3, wins, kms, nl) + ", ".join("%s %.1f" % (k, v * 1e3) for k, v in T.items()))
    print("   active instances per window:", act_hist[:12], "...", act_hist[-5:], " rows scored in window 0..3:", )
