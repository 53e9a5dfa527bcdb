"""Segmentation: the legacy loop's on-disk formats (SEGcpp:1209-1216, 1270-1277, 1376) and the validation loop of
SEG/trainer.py:699-745 driven through lpbox_hip.l2f, HIP solver vs oracle with the same deterministic policy."""
import os

import numpy as np
import pytest

from lpbox_hip import l2f
from test_seg_gpu_parity import make_pair

pytestmark = pytest.mark.gpu


def _near_binary(x):                   # policy stand-in: confident where the newest iterate is (exact in float32 on both sides)
    return x[:, -1, -1]


def test_legacy_files(tmp_path):
    g, o = make_pair(10000)
    type(g).write_files = True
    type(g).xiter_root = str(tmp_path / "xiter")
    type(g).result_root = str(tmp_path / "result")
    try:
        g.print_info = 1
        e = g.solve_iter()
    finally:
        type(g).write_files = None
        type(g).xiter_root = type(g).result_root = None
    reason, p1 = g.stop()
    with open(tmp_path / "xiter" / "0.csv") as f:
        lines = f.read().splitlines()
    assert len(lines) == p1 and lines[0].startswith("Iter1,") and lines[-1].startswith("Iter%d," % p1)
    assert lines[-1] == "Iter%d," % p1 + ",".join("%f" % v for v in g.debug_vec("x"))
    eo = o.solve_iter()                 # recording must not disturb the solve
    assert e == eo
    with open(tmp_path / "result" / "xiter_all.csv") as f:
        rec = [ln.split(",") for ln in f.read().splitlines()]
    assert len(rec) == 1 and int(rec[0][0]) == 0 and int(rec[0][3]) == p1
    assert float(rec[0][2]) == float("%f" % (g.debug_scalar("cur_obj") + g.debug_scalar("c"))) and int(float(rec[0][2])) == e


def test_validation_loop_matches_oracle():
    g, o = make_pair(10000)
    rg = l2f.run_l2f_seg(g, _near_binary)
    ro = l2f.run_l2f_seg(o, _near_binary)
    assert rg == ro and rg["fixed"] > 0
    assert np.array_equal(g.get_x_sol(), o.get_x_sol())


def test_device_loop_with_fused_policy_equals_host_loop():
    """SEG variant of row f1: the fused encoder (5 tokens, sliding windows = token stride 1) reads the packed x_iters on the device;
    the host loop feeds the same policy the reference-shaped (n, 5, 5) windows.  Same decisions, same energy."""
    import torch
    from lpbox_hip.policy import FusedEarlyFixPolicy, random_state
    pol = FusedEarlyFixPolicy(random_state(5, seed=2), tokens=5)
    # a policy with random weights scores ~0.5 everywhere: sharpen it so that it actually fixes (same function on both paths)
    class Sharp:
        def scores_from_xiters(self, flat, off, stride):
            return torch.sigmoid(5000.0 * (pol.logits_from_xiters(flat, off, stride) - pol.logits_from_xiters(flat, off, stride).median()))
        def __call__(self, x):
            lg = pol.logits(torch.from_numpy(x).cuda())
            return torch.sigmoid(5000.0 * (lg - lg.median())).cpu().numpy()
    g1, _ = make_pair(10000)
    g2, _ = make_pair(10000)
    r_dev = l2f.run_l2f_seg_device(g1, Sharp())
    r_host = l2f.run_l2f_seg(g2, Sharp())
    assert r_dev == r_host and r_dev["fixed"] > 0
    assert np.array_equal(g1.get_x_sol(), g2.get_x_sol())
