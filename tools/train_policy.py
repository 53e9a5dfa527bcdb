"""End to end on one MI355X: train the early-fixing policy on the iterates of plain solves (LP/trainer.py:254-299 recipe), then run
the product loop (LP/trainer.py:504-545) on held-out instances and compare with the plain solver: objective gap, iterations, time."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'accelerated-lpbox-admm_amd')); sys.path.insert(0,ROOT)
import numpy as np, torch
from bench import load_instances, FIXTURE
from lpbox_hip.lp import LpBatch
from lpbox_hip import l2f
from lpbox_hip.policy import EarlyFixPolicy, FusedEarlyFixPolicy
from lpbox_hip.train import TrainablePolicy, collect_training_data, train
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 100
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
insts = load_instances(FIXTURE)
train_set, test_set = insts[:n_train], insts[128:256]
torch.manual_seed(0)
t0 = time.perf_counter()
hist, labels, obj_train = collect_training_data(train_set)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("data: %d instances, histories %s, %.2f s; ones in labels %.3f" % (len(hist), tuple(hist[0].shape), t1 - t0, float(torch.cat(labels).mean())))
net = TrainablePolicy(20).cuda()
losses = train(net, hist, labels, epochs=epochs, lr=lr, log=lambda s: print(s) if int(s.split()[1][:-1]) % 5 == 0 else None)
torch.cuda.synchronize(); t2 = time.perf_counter()
print("training: %d epochs x %d steps in %.1f s, loss %.4f -> %.4f" % (epochs, len(hist), t2 - t1, losses[0], losses[-1]))
torch.save(net.state_dict(), os.path.join(ROOT, "gpurun_out", "policy_trained.pt"))
# held-out evaluation
B = len(test_set)
b = LpBatch(test_set); b.solve_init(); torch.cuda.synchronize(); t = time.perf_counter(); b.solve_iter(0, 20000); torch.cuda.synchronize(); t_plain = time.perf_counter() - t
obj_plain = np.array([-b.cal_obj(i) for i in range(B)]); it_plain = np.array([b.counters(i)[0] for i in range(B)]); inf_plain = np.array([b.check_infeasible_l2f(i) for i in range(B)])
print("plain: %.1f ms, mean objective %.2f, mean iterations %.0f, infeasible %d" % (t_plain * 1e3, obj_plain.mean(), it_plain.mean(), (inf_plain > 0).sum()))
for name, pol in (("fp32 torch", EarlyFixPolicy(net.state_dict(), tokens=20, device="cuda")), ("fused fp16", FusedEarlyFixPolicy(net.state_dict(), tokens=20))):
    for rep in range(2):
        b = LpBatch(test_set); b.solve_init(); tm = {}
        torch.cuda.synchronize(); t = time.perf_counter()
        res = l2f.run_l2f_batch(b, pol, ws=100, timing=tm)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    its = np.array([b.counters(i)[0] for i in range(B)])
    gap = (res["objective"] - obj_plain) / np.abs(obj_plain)
    fixed = sum(b.get_org_n(i) - b.get_n(i) for i in range(B))
    print("%s: %.1f ms (solve %.1f policy %.1f host %.1f), windows %d, mean iterations %.0f, fixed %.1f%% of variables, objective gap mean %+.4f (min %+.4f max %+.4f), infeasible %d" % (
        name, dt * 1e3, tm["solve"] * 1e3, tm["policy"] * 1e3, tm["host"] * 1e3, res["windows"], its.mean(), 100.0 * fixed / sum(I["n"] for I in test_set),
        gap.mean(), gap.min(), gap.max(), int((res["infeasible"] > 0).sum())))
    if hasattr(pol, "rescored"):
        print("   rows re-scored in fp32 (decision band) over both repetitions: %d" % pol.rescored)
