"""north_star: "objective gap <= reference".  The reference binary cannot be run (no Eigen), so "reference" is the CPU oracle in
Eigen's summation order.  The iteration is chaotic in rounding: two summation orders end on different binary
solutions of statistically equal quality, so the claim is tested as a paired comparison over the WHOLE 256-instance benchmark batch:
mean relative objective difference not below zero by more than two standard errors, nobody infeasible.

tests/golden/objective_study_100_500.npz is written by tools/objective_study.py on the GPU box (Eigen-order oracle on all 256
instances + the HIP solver of that day); the -m gpu test recomputes the HIP side and uses only the fixture's oracle half.
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN, lp_instances

FX = np.load(os.path.join(GOLDEN, "objective_study_100_500.npz"))


def _paired(gpu_obj, eigen_obj):
    gap = (gpu_obj - eigen_obj) / eigen_obj                  # objective = accepted bid prices (maximisation): positive = GPU order better
    return gap.mean(), gap.std(ddof=1) / np.sqrt(len(gap))


def test_fixture_objective_gap_not_worse_than_eigen_order():
    assert FX["eigen_obj"].shape == (256,) and FX["gpu_obj"].shape == (256,)
    assert int((FX["eigen_infeasible"] > 0).sum()) == 0 and int((FX["gpu_infeasible"] > 0).sum()) == 0
    assert set(np.unique(FX["eigen_stop"])) <= {1, 2} and set(np.unique(FX["gpu_stop"])) <= {1, 2}     # every solve met a reference stop test
    mean, se = _paired(FX["gpu_obj"], FX["eigen_obj"])
    assert mean >= -2 * se, (mean, se)
    assert abs(FX["gpu_iters"].mean() / FX["eigen_iters"].mean() - 1) < 0.03                          # same work: mean iterations within 3 %


def test_fixture_eigen_half_is_reproducible_on_cpu():
    """The oracle half of the fixture is what this repository's CPU oracle computes (3 instances re-solved here)."""
    from oracle import oracle as O
    insts = lp_instances("lp_100_500_seed0.npz")
    for i in (0, 100, 255):
        I = insts[i]
        s = O.LpOracle(0, order=O.ORDER_EIGEN)
        s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
        s.solve_init()
        s.solve_iter(0, 20000)
        assert -s.cal_Obj() == FX["eigen_obj"][i] and s.total_outer_iters == FX["eigen_iters"][i]


@pytest.mark.gpu
def test_gpu_objective_gap_not_worse_than_eigen_order():
    from lpbox_hip.lp import LpBatch
    insts = lp_instances("lp_100_500_seed0.npz")
    B = LpBatch(insts)
    B.solve_init()
    B.solve_iter(0, 20000)
    obj = np.array([-B.cal_obj(i) for i in range(256)])
    assert sum(B.check_infeasible_l2f(i) > 0 for i in range(256)) == 0
    mean, se = _paired(obj, FX["eigen_obj"])
    assert mean >= -2 * se, (mean, se)
    # never better than the exact optimum of the LP relaxation's integer problem would allow is checked in test_oracle_lp (milp bound)


# ------------------------------------------------------------------------------------------------------------------------
# The opt-in direct x-update is NOT the reference's algorithm; what is claimed for it is statistical: over the
# whole benchmark batch its objectives are not worse than the reference algorithm's (Eigen-order PCG oracle, the fixture above) by more
# than two standard errors, in either summation order, with the same amount of ADMM work.  objective_study_direct_100_500.npz is
# written by `tools/objective_study.py ... direct` (HIP direct mode + the oracle's mirror in Eigen's order, all 256 instances).
# ------------------------------------------------------------------------------------------------------------------------
FXD = np.load(os.path.join(GOLDEN, "objective_study_direct_100_500.npz"))


def test_direct_mode_fixture_not_worse_than_the_reference_algorithm():
    ref = FX["eigen_obj"]
    for side in ("eigen", "gpu"):
        obj, iters, stop, inf = FXD[side + "_obj"], FXD[side + "_iters"], FXD[side + "_stop"], FXD[side + "_infeasible"]
        capped = stop == 0                                    # ran into the loop bound of 20 000 iterations (LPcpp:796)
        assert capped.sum() <= 1 and (inf > 0).sum() <= capped.sum(), side     # every solve that stopped is feasible
        mean, se = _paired(obj, ref)
        assert mean >= -2 * se, (side, mean, se)
        assert abs(iters[~capped].mean() / FX["eigen_iters"].mean() - 1) < 0.05, side
    assert (FXD["gpu_pcg"] == 0).all() and (FXD["eigen_pcg"] == 0).all()


def test_direct_mode_fixture_oracle_half_is_reproducible_on_cpu():
    from oracle import oracle as O
    from test_direct_x_update import greedy_split
    insts = lp_instances("lp_100_500_seed0.npz")
    for i in (1, 200):
        I = insts[i]
        s = O.LpOracle(0, order=O.ORDER_EIGEN, x_update="direct", direct_rows=greedy_split(I))   # = the library's split (ascending greedy)
        s.set_problem(I["n"], I["l"], I["colptr"], I["rowidx"], I["b"])
        s.solve_init()
        s.solve_iter(0, 20000)
        assert -s.cal_Obj() == FXD["eigen_obj"][i] and s.total_outer_iters == FXD["eigen_iters"][i]


@pytest.mark.gpu
def test_direct_mode_row_split_is_the_ascending_greedy_one():
    """(what the CPU test above assumes about lpbox_set_x_update's host-side choice)"""
    from lpbox_hip.lp import LpBatch
    from test_direct_x_update import greedy_split
    insts = lp_instances("lp_100_500_seed0.npz")[:16]
    b = LpBatch(insts)
    b.set_x_update("direct")
    for i, I in enumerate(insts):
        assert np.array_equal(b.direct_rows(i), greedy_split(I)), i
