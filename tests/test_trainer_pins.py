"""The host-side callers either side of the hot path, pinned to vectors the REFERENCE'S OWN CODE produced
(tests/golden/make_trainer_fixtures.py imports LP/trainer.py, SEG/trainer.py and LP/common/utils.py in the build container):

* pure helpers: position code, deter_fix_2 thresholds, getLabel, getSubset;
* on-disk formats: files written by lpbox_hip/files.py as parsed by the reference's readFile / get_lpbox_info;
* the validation loops: the reference's `_valid_2` (LP/trainer.py:483-597) and `_my_valid` (SEG/trainer.py:676-811) were run
  end to end on the CPU oracle with a scripted network, logging every solver call; lpbox_hip.l2f.run_l2f / run_l2f_seg on the
  same solver, network and instances must make the identical calls and read the identical results.

CPU only: the solver under the loops is the oracle (the loops are host logic; HIP-vs-oracle parity of the solver itself is the
business of the -m gpu tests).
"""
import json
import os

import numpy as np

import helpers as H
from helpers import GOLDEN

FX = np.load(os.path.join(GOLDEN, "trainer_reference.npz"))


def test_position_code_equals_reference_position_encoding():
    from lpbox_hip.policy import position_code
    assert np.array_equal(position_code(20, 5).numpy(), FX["posenc_20_5"])          # LP/common/utils.py:20-32, bit for bit
    assert np.array_equal(position_code(5, 5).numpy(), FX["posenc_5_5"])


def test_fix_vector_from_scores_equals_deter_fix_2():
    from lpbox_hip.l2f import fix_vector_from_scores
    vec, f1, f0 = fix_vector_from_scores(FX["fix2_sig"])                              # incl. scores exactly at 0.9 / 0.1 in float32
    assert np.array_equal(vec, FX["fix2_vec"]) and [f1, f0] == FX["fix2_f1f0"].tolist()


def test_label_and_subset_rules():
    from lpbox_hip import files
    assert np.array_equal(files.labels_from_iterates(FX["label_in"]), FX["label_out"])   # getLabel: >= 0.5, shape (n, 1)
    assert np.array_equal(files.window_subset(FX["subset_in"], 2, 4), FX["subset_out_2_4"])


def test_xiters_csv_roundtrip_through_reference_reader(tmp_path):
    from lpbox_hip import files
    p = tmp_path / "x.csv"
    files.write_xiters_csv(p, FX["readfile_in"])
    got = files.read_xiters_csv(p)
    assert np.array_equal(got, FX["readfile_out"])                    # what the reference's readFile returned for OUR file
    assert np.allclose(FX["readfile_out"].T, FX["readfile_in"], atol=5.1e-7, rtol=0)     # "%lf" keeps 6 decimals
    first = open(p).readline()
    assert first.startswith("Iter1,") and first.count(",") == FX["readfile_in"].shape[1]


def test_result_files_as_parsed_by_reference(tmp_path):
    from lpbox_hip import files
    p, q = tmp_path / "allres.csv", tmp_path / "xiter_all.csv"
    for i in range(1, 11):
        files.append_allres(p, i, 1234.5 + i / 3.0, 7000 + i, 0.25 * i)
    for it in range(10):
        files.append_xiter_all(q, it, 100.0 + it, 150.5 + it, 40 + it, 0.125 * (it + 1))
    assert np.array_equal(np.array(files.read_results_csv(p)), FX["allres_out"])
    assert np.array_equal(np.array(files.read_results_csv(q)), FX["seg_xiter_all_out"])
    assert FX["allres_out"].shape == (10, 4) and FX["seg_xiter_all_out"].shape == (10, 5)   # columns the trainers index (:547-549; SEG :745-746)


def _lp_instances():
    out, cp, ri, pr = [], 0, 0, 0
    for n, l, nnz in zip(FX["lpinst_n"], FX["lpinst_l"], FX["lpinst_nnz"]):
        n, l, nnz = int(n), int(l), int(nnz)
        out.append(dict(n=n, l=l, colptr=FX["lpinst_colptr"][cp:cp + n + 1].astype(np.int32),
                        rowidx=FX["lpinst_rowidx"][ri:ri + nnz].astype(np.int32), b=-1.0 * FX["lpinst_price"][pr:pr + n]))
        cp += n + 1; ri += nnz; pr += n
    return out


def _score(x):
    return H.scripted_scores(x).numpy()


def test_run_l2f_makes_the_calls_of_reference_valid_2():
    from lpbox_hip.l2f import run_l2f
    ref = json.loads(str(FX["lp_valid2_log"]))
    log = H.CallLog()
    H.LoggedLpSolver.instances, H.LoggedLpSolver.log = _lp_instances(), log
    for it in range(1, 11):                                            # LP/trainer.py:501-506
        s = H.LoggedLpSolver(0)
        s.read_File(it, 100, 500)
        s.solve_init()
        r = run_l2f(s, _score, ws=100, max_iter=1e4, col=80)
        assert r["objective"] == -log.solvers[-1]["final"]["cal_obj"] and r["windows"] == len(log.solvers[-1]["windows"])
    assert len(ref) == 10 and sum(len(s["windows"]) for s in ref) > 300      # the fixture exercises multi-window fixing
    assert log.solvers == ref


def test_run_l2f_seg_makes_the_calls_of_reference_my_valid():
    from lpbox_hip.l2f import run_l2f_seg
    ref = json.loads(str(FX["seg_myvalid_log"]))
    log = H.CallLog()
    H.LoggedSegSolver.log = log
    for it in range(10):                                               # SEG/trainer.py:696-700
        s = H.LoggedSegSolver(0, 1e4, it)
        s.solve_init()
        r = run_l2f_seg(s, _score, ws=10, max_iter=30)
        assert r["energy"] == log.solvers[-1]["final"]["obj"]
        s.get_x_sol()                                                  # :749
    assert log.solvers == ref
    assert any(w["num"] > 10 for s in ref for w in s["windows"])
