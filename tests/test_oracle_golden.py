"""The oracle (own Householder QR) against the committed golden fixtures, which were generated
with the reference's vendored Eigen QR doing the linear solve (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

import oracle_api
from oracle_api import Oracle


def _scale_err(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))


@pytest.fixture(scope="module")
def batch(golden_dir):
    return np.load(os.path.join(golden_dir, "f3_batch.npz"))


def test_fixture_inputs_are_the_seeded_generator(batch, oracle):
    from rocket_path_amd import problems
    pos, dist = batch["pos"], batch["dist"]
    first = 0
    for d in (0, 2, 1):
        n = int((dist == d).sum())
        p = np.stack(problems.generate(int(batch["seed"]), first, n, d), axis=1)
        assert np.array_equal(p, pos[first:first + n])
        q = np.stack(oracle.gen_problems(int(batch["seed"]), first, n, d), axis=1)
        assert np.array_equal(q, p)              # C and numpy generators agree bit for bit
        first += n
    init = oracle.batch_init_feasible(3, pos[:, 0], pos[:, 1], pos[:, 2])
    assert np.array_equal(init, batch["init"])


def test_fixed_steps_match_golden(batch, oracle):
    aos = batch["init"].copy()
    oracle.batch_steps(3, aos, 1)
    assert _scale_err(aos[:, :3], batch["after1"][:, :3]) < 1e-11
    oracle.batch_steps(3, aos, 4)
    assert _scale_err(aos[:, :3], batch["after5"][:, :3]) < 1e-10
    oracle.batch_steps(3, aos, 45)
    assert _scale_err(aos[:, :3], batch["after50"][:, :3]) < 1e-10


def test_multipliers_after_fifty_steps_are_pinned_only_where_they_are_determined(batch, oracle):
    """After 50 fixed steps (35 of them in the reference's post-convergence regime) the oracle's own QR and the reference's
    Eigen QR -- the fixture -- agree on the multipliers to rounding on the monotone and reference-like sets.  On the
    non-monotone stress set the optimum has four active constraints for three variables, the multipliers have a null
    direction, and the two CPU evaluations of the same algorithm drift apart along it (measured: 0.42 of the largest
    multiplier at worst, median 8e-7): which is why the GPU test asserts no tolerance there."""
    aos = batch["init"].copy()
    oracle.batch_steps(3, aos, 50)
    ref = batch["after50"][:, 3:11]
    err = np.max(np.abs(aos[:, 3:11] - ref) / np.max(np.abs(ref), axis=1, keepdims=True), axis=1)
    regular = batch["dist"] != 2
    assert err[regular].max() < 1e-12
    assert np.median(err[~regular]) < 1e-4 and err[~regular].max() > 1e-6      # drifted, but the same optimum
    assert _scale_err(aos[:, :3], batch["after50"][:, :3]) < 1e-10


def test_gated_solve_matches_golden(batch, oracle):
    aos = batch["init"].copy()
    iters, total = oracle.batch_solve_gated(3, aos, 1e-8, 200)
    assert np.array_equal(iters, batch["iters"])
    assert total == int(batch["iters"].sum())
    assert _scale_err(aos[:, :3], batch["gated"][:, :3]) < 1e-10
    mono = batch["dist"] == 0
    assert 12 <= iters[mono].min() and iters[mono].max() <= 40        # 12-37 typical
    assert (iters < 200).all()


def test_threaded_batch_equals_serial(batch, oracle):
    a = batch["init"][:512].copy()
    b = batch["init"][:512].copy()
    ia, _ = oracle.batch_solve_gated(3, a, 1e-8, 200, threads=1)
    ib, _ = oracle.batch_solve_gated(3, b, 1e-8, 200, threads=4)
    assert np.array_equal(ia, ib) and np.array_equal(a, b)


def test_named_trajectories(golden_dir, oracle):
    t = np.load(os.path.join(golden_dir, "f3_trajectories.npz"))
    v = oracle.init_default(3)
    assert np.array_equal(v[11:], t["default_const"])
    for s in range(1, 51):
        oracle.step(3, v)
        assert _scale_err(v[:3], t["default_states"][s, :3]) < 1e-12, s
    v = oracle.init_stuck()
    for s in range(1, 31):
        oracle.step(3, v)
    assert _scale_err(v[:3], t["stuck_states"][30, :3]) < 1e-9
    v = oracle.init_default(4)
    for s in range(1, 11):                    # F4 is chaotic beyond a few steps
        oracle.step(4, v)
        assert _scale_err(v[:3], t["f4_default_states"][s, :3]) < 1e-9, s


def test_f4_single_steps(golden_dir, oracle):
    t = np.load(os.path.join(golden_dir, "f4_steps.npz"))
    worst = 0.0
    for i in range(0, 4096, 8):
        v = t["state_in"][i].copy()
        oracle.step(4, v)
        worst = max(worst, _scale_err(v[:3], t["state_out"][i, :3]))
    assert worst < 1e-9


@pytest.mark.skipif(not oracle_api.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_own_qr_against_reference_eigen_qr(batch):
    orc = Oracle(eigen=True)
    rng = np.random.RandomState(3)
    # KKT systems along real trajectories
    for i in rng.choice(len(batch["init"]), 40, replace=False):
        v = batch["init"][i].copy()
        for s in range(12):
            M, r, _ = orc.kkt(3, v)
            x_own, nz_own = orc.qr_solve(M, -r)
            x_ref, nz_ref = orc.ref_qr_solve(M, -r)
            assert nz_own == nz_ref == 11
            assert np.max(np.abs(x_own - x_ref) / np.maximum(np.abs(x_ref), 1e-3 * np.max(np.abs(x_ref)))) < 1e-6
            # both are backward stable: compare residuals of the linear system
            assert np.linalg.norm(M @ x_own + r) <= 1e-9 * (np.linalg.norm(M) * np.linalg.norm(x_own) + np.linalg.norm(r))
            orc.step(3, v)
    # generic random systems, fixed- and dynamic-size code paths
    for n in (3, 4, 7, 8, 11):
        A = rng.randn(n, n)
        b = rng.randn(n)
        x_own, _ = orc.qr_solve(A, b)
        x_ref, _ = orc.ref_qr_solve(A, b)
        x_dyn, _ = orc.ref_qr_solve(A, b, force_dynamic=True)
        assert np.allclose(x_own, x_ref, rtol=1e-9, atol=1e-11)
        assert np.allclose(x_dyn, x_ref, rtol=1e-9, atol=1e-11)
    # the squared-norm emulation used by the Armijo test is Eigen's, bit for bit: compare
    # residualNorm (emulated reduction order) with Eigen's squaredNorm of the same residual
    import ctypes
    res = orc.lib.orc_residual
    res.argtypes = [ctypes.c_int, oracle_api._dp, ctypes.c_double, oracle_api._dp]
    for i in rng.choice(len(batch["init"]), 60, replace=False):
        v = batch["init"][i].copy()
        for s in range(10):
            p = orc.gap(3, v) / 80.0
            r = np.zeros(11)
            res(3, v.ctypes.data_as(oracle_api._dp), p, r.ctypes.data_as(oracle_api._dp))
            assert orc.ref.ref_squared_norm(11, r.ctypes.data_as(oracle_api._dp)) == orc.residual_norm(3, v, p)
            orc.step(3, v)


@pytest.mark.skipif(not oracle_api.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_own_qr_against_reference_eigen_on_feasibility_gram_systems():
    """moveTowardFeasibility solves a = G G^T by the DYNAMIC-size Eigen QR (onedpath_ip.cpp:676-693).  The oracle's QR in
    its dynamic form (orc_colpiv_qr_solve_dynamic: Eigen's run-time-sized reduction orders) reproduces the reference's
    Eigen bit for bit on Gram systems of 1, 2, 3 AND 4 violated rows.  Four rows of three-variable gradients give a
    matrix that is singular in exact arithmetic: its last pivot is rounding noise, and only the same sequence of
    roundings gives the same answer -- round 2's fixed-size ordering agreed on half of such systems, which is what
    this test used to record.  (The whole move against the reference's own function: tests/test_oracle_reference.py.)"""
    import rocket_path_amd as rp
    orc = Oracle(eigen=True)
    rng = np.random.RandomState(5)
    for variant in (3, 4):
        m = orc.num_constraints(variant)
        n_prob = 3600
        p0, p1, p2 = rp.problems.generate(36, 0, n_prob, 0)
        aos = orc.batch_init_feasible(variant, p0, p1, p2)
        short = rng.uniform(0.3, 1.2, (n_prob, 2))
        aos[:, 1] *= short[:, 0]
        aos[:, 2] *= short[:, 1]
        aos[:, 0] = rng.uniform(-50, 250, n_prob)
        aos[::2, 0] = rng.uniform(-5, 5, len(aos[::2]))
        aos[::2, 1:3] *= 0.6
        equal = {1: [], 2: [], 3: [], 4: []}
        fixed_form_equal = []
        for row in aos:
            rows = [orc.constraint(variant, i, row) for i in range(m)]
            viol = [(e, g) for e, g in rows if e > 0]
            if not viol:
                continue
            G = np.array([g for _, g in viol])
            e = np.array([x for x, _ in viol])
            n = len(e)
            A = np.zeros((n, n))
            for i in range(n):
                for j in range(n):
                    acc = 0.0
                    for k in range(3):
                        acc += G[i, k] * G[j, k]
                    A[i, j] = acc
            x_own, nz_own = orc.qr_solve(A, e, dynamic=True)
            x_ref, nz_ref = orc.ref_qr_solve(A, e, force_dynamic=True)
            assert np.all(np.isfinite(x_own)) and np.all(np.isfinite(x_ref))
            equal[n].append(np.array_equal(x_own, x_ref) and nz_own == nz_ref)
            if n == 4:
                x_fix, nz_fix = orc.qr_solve(A, e)
                fixed_form_equal.append(np.array_equal(x_fix, x_ref) and nz_fix == nz_ref)
        for n in (1, 2, 3):
            assert len(equal[n]) > 20 and all(equal[n]), (variant, n)
        assert len(equal[4]) >= 2000 and all(equal[4]), (variant, len(equal[4]), np.mean(equal[4]))
        assert np.mean(fixed_form_equal) < 0.9      # the order matters: the fixed-size form is NOT the reference here


def test_multipliers_at_the_gate_are_pinned_where_they_are_well_conditioned(batch, oracle):
    """The measured reason for the two multiplier tolerances of tests/test_gpu_parity.py: the golden gated states were
    produced with the reference's Eigen QR, this solve uses the oracle's own QR -- same algebra, different rounding
    order.  (v, t0, t1) and the iteration counts agree everywhere; the multipliers agree to rounding on the monotone
    and reference-like sets and only to ~5e-8 (of the problem's largest multiplier) on the non-monotone stress set."""
    aos = batch["init"].copy()
    it, _ = oracle.batch_solve_gated(3, aos, 1e-8, 200)
    assert np.array_equal(it, batch["iters"])
    ref = batch["gated"][:, 3:11]
    err = np.max(np.abs(aos[:, 3:11] - ref) / np.max(np.abs(ref), axis=1, keepdims=True), axis=1)
    regular = batch["dist"] != 2
    assert err[regular].max() < 1e-13
    assert 1e-9 < err[~regular].max() < 1e-7
