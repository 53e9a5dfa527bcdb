"""The early-fixing LOOP around the solver (LP/trainer.py:504-545): the harness in
lpbox_hip.l2f drives the HIP solver, the same harness drives the oracle, both with the same deterministic policy."""
import numpy as np
import pytest

from helpers import bits_equal, lp_instances, oracle_for
from lpbox_hip import l2f
from lpbox_hip.lp import LpBatch, PyLPboxADMMsolver

pytestmark = pytest.mark.gpu


def _last_iterate(x):                 # policy stand-in: the score of a variable is its latest iterate (exact in float32 on CPU and GPU)
    return x[:, -1, -1]


def test_fix_vector_rule():
    vec, f1, f0 = l2f.fix_vector_from_scores([0.95, 0.5, 0.05, 0.9, 0.1])
    assert vec.tolist() == [1.0, -1.0, 0.0, -1.0, -1.0] and (f1, f0) == (1, 1)      # strict > / < (LP/trainer.py:118-125)


def test_single_instance_loop_matches_oracle():
    I = lp_instances("lp_100_500_seed0.npz")[3]
    g = PyLPboxADMMsolver(0)
    g.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
    g.solve_init()
    o = oracle_for(g.batch, 0, I)
    rg = l2f.run_l2f(g, _last_iterate, ws=100)
    ro = l2f.run_l2f(o, _last_iterate, ws=100)
    assert rg == ro and rg["fixed"] > 0
    assert bits_equal(g.get_x_sol(I["n"]), o.get_x_sol(I["n"]))


def test_batched_device_loop_matches_per_instance_oracle():
    insts = lp_instances("lp_100_500_seed0.npz")[:6]
    b = LpBatch(insts)
    b.solve_init()
    res = l2f.run_l2f_batch(b, _last_iterate, ws=100)
    for i, I in enumerate(insts):
        o = oracle_for(b, i, I)
        ro = l2f.run_l2f(o, _last_iterate, ws=100)
        assert res["objective"][i] == ro["objective"] and res["infeasible"][i] == ro["infeasible"], i
        assert bits_equal(b.get_x_sol(i).ravel(), o.get_x_sol().ravel()), i


def test_batched_loop_with_fused_policy_runs_and_matches_unfused_decisions():
    """The fused encoder inside the loop: same windows / fixed sets as the fp32 torch evaluation of the same weights wherever the
    scores are not within 1e-3 of a threshold (fp16 operands); with reference-style initial weights nothing gets near one."""
    import torch
    from lpbox_hip.policy import EarlyFixPolicy, FusedEarlyFixPolicy, random_state
    sd = random_state(20, seed=0)
    insts = lp_instances("lp_100_500_seed0.npz")[:4]
    res = []
    for pol in (FusedEarlyFixPolicy(sd, tokens=20), EarlyFixPolicy(sd, tokens=20, device="cuda")):
        b = LpBatch(insts)
        b.solve_init()
        res.append(l2f.run_l2f_batch(b, pol, ws=100, max_iter=1000))
    assert res[0]["windows"] == res[1]["windows"] and np.array_equal(res[0]["objective"], res[1]["objective"])


def test_training_pipeline_end_to_end_small():
    """lpbox_hip.train on the device: histories recorded by the solver (zero-copy), labels from the full solve, a few optimiser steps,
    and the trained weights drive the fused policy inside the batched loop."""
    import torch
    from lpbox_hip.policy import FusedEarlyFixPolicy
    from lpbox_hip.train import TrainablePolicy, collect_training_data, train
    insts = lp_instances("lp_100_500_seed0.npz")[:4]
    hist, labels, obj = collect_training_data(insts)
    assert len(hist) == 4 and hist[0].shape == (500, 1000) and labels[0].shape == (500, 1) and hist[0].is_cuda
    # the recorded windows are the plain solve's iterates: column 0 is x after the first iteration, bit-identical to a fresh run
    b = LpBatch(insts[:1]); b.solve_init(); b.solve_iter(0, 1)
    assert bits_equal(hist[0][:, 0].cpu().numpy(), b.debug_vec("x", 0))
    torch.manual_seed(0)
    net = TrainablePolicy(20).cuda()
    losses = train(net, hist, labels, epochs=4, lr=1e-3)
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    b = LpBatch(insts); b.solve_init()
    res = l2f.run_l2f_batch(b, FusedEarlyFixPolicy(net.state_dict(), tokens=20), ws=100, max_iter=500)
    assert res["windows"] >= 1 and np.isfinite(res["objective"]).all()
