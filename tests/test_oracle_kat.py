"""The oracle against the known-answer vectors the survey captured from the unmodified
reference (tests/golden/survey_kat.json = SURVEY.md 8c).  With the reference's own Eigen QR
(oracle/_ref) the restatement must hit them bit for bit; with its own QR, to rounding."""
import json
import os

import numpy as np
import pytest

import oracle_api
from oracle_api import Oracle, StepInfo


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "survey_kat.json")) as f:
        return json.load(f)


def _scale_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))


def test_default_init_quantities_are_bit_exact(oracle, kat):
    k = kat["f3_default"]
    v = oracle.init_default(3)
    errs = [oracle.constraint(3, i, v)[0] for i in range(8)]
    assert errs == k["init_errors"]
    for i, row in k["init_grad_rows_v_t0_t1"].items():
        g = oracle.constraint(3, int(i), v)[1]
        assert list(g) == row
        assert list(oracle.constraint(3, int(i) + 1, v)[1]) == [-x for x in row]   # odd rows are the negation
    assert oracle.gap(3, v) == k["init_gap"]
    assert oracle.residual_norm(3, v, k["init_perturbation"]) == k["init_residual_norm_sq"]


@pytest.mark.skipif(not oracle_api.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_with_reference_eigen_qr_every_kat_is_bit_exact(kat):
    orc = Oracle(eigen=True)
    assert orc.ref.ref_eigen_version() == b"3.3.0"
    k = kat["f3_default"]
    v = orc.init_default(3)
    info = StepInfo()
    feas, res = [], []
    for it in range(1, 31):
        orc.step(3, v, info)
        feas.append(info.feas_halvings)
        res.append(info.resid_halvings)
        if it == 1:
            assert list(v[:11]) == k["after_step_1"]
        if it == 2:
            assert list(v[:11]) == k["after_step_2"]
        if it == 5:
            assert list(v[:3]) == k["after_step_5_v_t0_t1"]
        if it == 15:
            assert list(v[:3]) == k["after_step_15_v_t0_t1"]
            assert abs(orc.gap(3, v) - k["gap_after_step_15"]) < 1e-11
        if it == 25:
            assert list(v[:3]) == k["after_step_25_v_t0_t1"]
            assert [v[4], v[6], v[7], v[9]] == k["after_step_25_multipliers_1_3_4_6"]
            assert abs(v[3] / k["after_step_25_small_multiplier_magnitude"] - 1) < 1e-2
    assert feas[:8] == k["feasibility_halvings_steps_1_to_8"]
    assert feas[21] == k["feasibility_halvings_step_22"]
    assert max(res[:24]) == k["residual_halvings_before_step_25"]
    assert res[26:] == [k["residual_halvings_per_step_once_frozen"]] * 4

    m = kat["f3_monotone_sample"]
    v = orc.init_feasible(3, *m["pos"])
    assert [v[1], v[2]] == m["init_t0_t1"]
    assert orc.solve_gated(3, v) == m["gated_steps"]
    assert list(v[:3]) == m["gated_v_t0_t1"]

    f4 = kat["f4_default"]
    v = orc.init_default(4)
    assert abs(orc.gap(4, v) - f4["init_gap"]) < 1e-4
    orc.step(4, v)
    assert list(v[:3]) == f4["after_step_1_v_t0_t1"]
    assert abs(orc.gap(4, v) - f4["gap_after_step_1"]) < 1e-5
    for _ in range(49):
        orc.step(4, v)
    assert abs(orc.gap(4, v) - f4["gap_after_step_50_approx"]) < 1e-3
    assert 1e-10 < abs(v[1] - v[2]) < 1e-8


def test_own_qr_matches_kats_to_rounding(oracle, kat):
    k = kat["f3_default"]
    v = oracle.init_default(3)
    gate = None
    for it in range(1, 26):
        if gate is None and oracle.gap(3, v) < 1e-8:
            gate = it - 1
        oracle.step(3, v)
        if it == 1:
            assert _scale_err(v[:11], k["after_step_1"]) < 1e-12
        if it == 2:
            assert _scale_err(v[:11], k["after_step_2"]) < 1e-12
        if it == 5:
            assert _scale_err(v[:3], k["after_step_5_v_t0_t1"]) < 1e-12
        if it == 15:
            assert _scale_err(v[:3], k["after_step_15_v_t0_t1"]) < 1e-13
    assert gate == k["first_step_with_gap_below_1e-8"]
    assert _scale_err(v[:3], k["after_step_25_v_t0_t1"]) < 1e-14
    m = kat["f3_monotone_sample"]
    v = oracle.init_feasible(3, *m["pos"])
    assert oracle.solve_gated(3, v) == m["gated_steps"]
    assert _scale_err(v[:3], m["gated_v_t0_t1"]) < 1e-13


def test_stuck_and_infeasible_behaviour(oracle, kat):
    v = oracle.init_stuck()
    info = StepInfo()
    feas = []
    for _ in range(30):
        oracle.step(3, v, info)
        feas.append(info.feas_halvings)
    assert abs(oracle.gap(3, v) - kat["f3_stuck"]["gap_after_30_steps_approx"]) < 5e-3
    assert set(feas) <= set(kat["f3_stuck"]["feasibility_halvings_alternate"])
    assert feas.count(0) >= 12 and sum(1 for f in feas if f >= 10) >= 12

    v = oracle.init_default(3)
    v[13] = kat["f3_infeasible_start"]["pos1"]
    before = v.copy()
    for _ in range(3):
        oracle.step(3, v, info)
        assert info.feas_halvings == kat["f3_infeasible_start"]["feasibility_halvings_every_step"]
    assert not oracle.satisfied(3, v)
    assert np.max(np.abs(v - before)) < 1e-25     # 0.99 * 2^-100 of a finite step: frozen
