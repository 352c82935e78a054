"""BASELINE.json's full sizes.  The oracle checks a slice it can finish in seconds; the whole
batch is checked through size-independent properties of the solve."""
import numpy as np
import pytest

import rocket_path_amd as rp
from parity_util import certify_iteration_counts, keep_mask

pytestmark = pytest.mark.gpu


def serr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def test_config3_one_million_gated(oracle):
    n = 1 << 20
    p0, p1, p2 = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
    with rp.Batch(n) as b:
        b.set_problems(p0, p1, p2)
        b.solve(1e-8, 200, 0)
        it, status = b.get_iters()
        st = b.get_state()
        r = b.reduce()
        pos, acc = b.sample()
    # properties over the whole batch
    assert np.all(status == rp.ST_CONVERGED)
    assert r["n_converged"] == n and r["total_steps"] == float(it.sum()) and r["max_gap"] < 1e-8
    assert 12 <= it.min() and it.max() <= 60 and 14.5 < it.mean() < 16.5
    assert np.all(np.isfinite(st))
    assert np.max(np.abs(acc)) <= 100.0 * (1 + 1e-12)                             # feasible: |a| <= L at all 4 ends
    assert np.all(np.max(np.abs(acc), axis=1) > 100.0 * (1 - 1e-6))               # and the limit is active at the optimum
    assert np.all(st[:, 1] > 0) and np.all(st[:, 2] > 0) and np.all(st[:, 3:11] >= 0)
    assert np.array_equal(st[:, 11:], np.stack([p0, np.zeros(n), p1, p2, np.zeros(n)], axis=1))   # constants untouched
    # translation invariance: shifting all positions by a constant leaves (v, t0, t1) and the counts unchanged
    m = 1 << 16
    with rp.Batch(m) as c:
        c.set_problems(p0[:m] + 128.0, p1[:m] + 128.0, p2[:m] + 128.0)
        c.solve(1e-8, 200, 0)
        it2, _ = c.get_iters()
        st2 = c.get_state()
    same = it2 == it[:m]                      # a shifted problem rounds differently: a gate tie (parity_util) may move one step
    assert (~same).sum() <= 2 and np.all(np.abs(it2 - it[:m]) <= 1) and serr(st2[same, :3], st[:m, :3][same]) < 1e-10
    # the oracle on ALL 1,048,576 problems (every host thread: 1.5 s on the GPU box's 16): iteration counts -- identical but for
    # certified gate ties, whose number is printed -- and (v, t0, t1) at 1e-10, multipliers at 1e-9
    _whole_batch_against_the_oracle(oracle, p0, p1, p2, it, st, "configs[2], 1,048,576 monotone problems", lam_tol=1e-9)


def _whole_batch_against_the_oracle(oracle, p0, p1, p2, it, st, what, lam_tol=None, max_ties=8):
    n = len(p0)
    init = oracle.batch_init_feasible(3, p0, p1, p2)
    aos = init.copy()
    it_o, total = oracle.batch_solve_gated(3, aos, 1e-8, 200, threads=0)
    ties = certify_iteration_counts(oracle, 3, init, it, it_o, 1e-8, max_ties=max_ties)
    ok = keep_mask(n, ties)
    err = serr(st[ok, :3], aos[ok, :3])
    msg = "%s against the oracle in full: %d problems, %d oracle steps, %d certified gate tie(s), 0 other iteration mismatches, worst (v, t0, t1) error %.2e" % (
        what, n, total, len(ties), err)
    if lam_tol is not None:
        lerr = float(np.max(np.abs(st[ok, 3:11] - aos[ok, 3:11]) / np.max(np.abs(aos[ok, 3:11]), axis=1, keepdims=True)))
        msg += ", worst multiplier error %.2e" % lerr
        assert lerr < lam_tol, lerr
    print(msg)
    assert err < 1e-10, err
    return len(ties)


def test_config2_65536_fixed_50_steps(oracle):
    n = 65536
    p0, p1, p2 = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
    aos = oracle.batch_init_feasible(3, p0, p1, p2)
    oracle.batch_steps(3, aos, 50)
    with rp.Batch(n) as b:
        b.set_problems(p0, p1, p2)
        b.step(50)
        st = b.get_state()
        it, _ = b.get_iters()
    assert np.all(it == 50) and np.all(np.isfinite(st))
    assert serr(st[:, :3], aos[:, :3]) < 1e-10


def test_config5_f4_fp32_one_million_runs_and_stays_finite():
    n = 1 << 20
    p0, p1, p2 = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
    with rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32) as b:
        b.set_problems(p0, p1, p2)
        b.step(50)
        st = b.get_state()
        pos, acc = b.sample()
    assert np.all(np.isfinite(st))
    assert np.all(st[:, 1] > 0) and np.all(st[:, 2] > 0)
    assert np.max(np.abs(acc)) <= 100.0 * (1 + 1e-4)      # the line search never accepts an infeasible point


def test_config5_f4_fp32_state_one_million_one_step_against_the_fp64_oracle(oracle):
    # config 5 at full size, per-problem bound over the WHOLE batch (rounds 2-5 compared a 65,536 slice): after
    # 10 steps of the 1 Mi batch (fp32 state, fp64 arithmetic) take one more step and compare all 1,048,576 problems with the fp64
    # oracle stepping from the identical (fp32-representable) states -- and again from the state 30 more steps leave, where F4's
    # long halving sequences and its stuck problems are there (the oracle steps 1 Mi F4 problems in a few seconds on the host's threads)
    n = 1 << 20
    p0, p1, p2 = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
    worst = []
    with rp.Batch(n, rp.VARIANT_F4, rp.DTYPE_F32_STATE) as b:
        b.set_problems(p0, p1, p2)
        done = 0
        for upto in (10, 41):
            b.step(upto - done)
            before = b.get_state()
            b.step(1)
            after = b.get_state()
            done = upto + 1
            assert np.array_equal(before, before.astype(np.float32).astype(np.float64))
            exp = before.copy()
            oracle.batch_steps(4, exp, 1, threads=0)
            err = np.abs(after[:, :3] - exp[:, :3]) / np.maximum(np.abs(exp[:, :3]), 1.0)
            scale = np.max(np.abs(exp[:, 3:7]), axis=1, keepdims=True)
            lerr = np.abs(after[:, 3:7] - exp[:, 3:7]) / scale
            worst.append((upto + 1, float(err.max()), float(lerr.max()), int((err.max(axis=1) > 1.0e-7).sum()), int(np.all(after == before, axis=1).sum())))
            # SURVEY C5 asks for ~1e-5 per problem.  Ten steps in, every one of the 1,048,576 problems is within fp32 rounding (1e-7, the bound
            # the 4,096 golden states meet); 41 steps in, F4's problems sit in 12-21-halving sequences whose last trial is a tie for a few of a
            # million of them (the device's reciprocal against the reference's divisions: one more halving of a step of ~1e-6): 1e-6 there,
            # with the number of problems beyond 1e-7 printed
            tol = 1.0e-7 if upto == 10 else 1.0e-6
            assert err.max() < tol, (upto, err.max())
            assert lerr.max() < 10 * tol, (upto, lerr.max())
        b.step(50 - done)
        st = b.get_state()
        pos, acc = b.sample_range(0, 1 << 16)
    print("configs[4], fp32 state + fp64 arithmetic, all 1,048,576 problems against the fp64 oracle, one step from identical states: " +
          "; ".join("step %d: worst (v, t0, t1) %.2e, multipliers %.2e, %d problems beyond 1e-7, %d left bit for bit unchanged" % w for w in worst))
    assert np.all(np.isfinite(st)) and np.all(st[:, 1] > 0) and np.all(st[:, 2] > 0)
    assert np.max(np.abs(acc)) <= 100.0 * (1 + 1e-6)      # feasible up to the rounding of the state to fp32


def test_mirror_symmetry_property():
    # Reflecting a problem (p0,p1,p2) -> (-p2,-p1,-p0) swaps the two segments: the optimum has the same
    # midpoint velocity and swapped durations.  The arithmetic is not symmetric (seg 0 and seg 1 are
    # evaluated by different expressions), so this is a property of the converged answer, checked at the
    # level the gate guarantees rather than bit for bit.
    n = 1 << 18
    p0, p1, p2 = rp.problems.generate(2718, 0, n, rp.problems.DIST_MONOTONE)
    with rp.Batch(n) as a, rp.Batch(n) as b:
        a.set_problems(p0, p1, p2)
        b.set_problems(-p2, -p1, -p0)
        a.solve(1e-8, 200, 0)
        b.solve(1e-8, 200, 0)
        sa, sb = a.get_state(), b.get_state()
        ia, ib = a.get_iters()[0], b.get_iters()[0]
    assert np.mean(ia != ib) < 1e-3
    same = ia == ib
    assert serr(sb[same][:, [0, 2, 1]], sa[same][:, :3]) < 1e-8
    # The multipliers swap too.  x -> -x with the order of the nodes reversed is a time reversal: velocities keep their
    # sign, accelerations change it, initial and final ends and the two segments trade places -- so constraint i of the
    # mirrored problem (enum V: seg 0 init min/max, seg 0 final min/max, seg 1 init min/max, seg 1 final min/max) is
    # constraint 7 - i of the original, and lambda'_i = lambda_{7-i}.  Compared relative to the problem's largest
    # multiplier, at the level the gate leaves them (the inactive ones are ~ gap / |c|).
    la, lb = sa[same][:, 3:11], sb[same][:, 3:11][:, ::-1]
    lerr = np.max(np.abs(la - lb), axis=1) / np.max(np.abs(la), axis=1)
    print("mirror symmetry: multipliers swapped to %.2e (99.9 %% quantile %.2e)" % (lerr.max(), np.quantile(lerr, 0.999)))
    assert lerr.max() < 1e-6
    assert np.max(np.abs(la - sb[same][:, 3:11]), axis=1).max() / np.max(np.abs(la)) > 1e-3      # ... and only in that pattern


def test_non_monotone_stress_one_million(oracle):
    # mid position outside [start, end] for most problems: the optimum has vel1 ~ 0 and the scale-aware tolerance matters
    n = 1 << 20
    p0, p1, p2 = rp.problems.generate(4321, 0, n, rp.problems.DIST_NON_MONOTONE)
    with rp.Batch(n) as b:
        b.set_problems(p0, p1, p2)
        b.solve(1e-8, 200, 0)
        it, status = b.get_iters()
        st = b.get_state()
    assert np.all(np.isfinite(st)) and np.all(status == rp.ST_CONVERGED) and it.max() < 200
    _whole_batch_against_the_oracle(oracle, p0, p1, p2, it, st, "the non-monotone stress set, 1,048,576 problems")


def test_five_million_problems_ragged(oracle):
    # 64-bit indexing and a ragged last tile far from the start of the arrays (field offsets beyond 2^31 bytes... at f64
    # 16 fields x 5,000,003 x 8 B = 640 MB; per-field stride 40 MB)
    n = 5_000_003
    p0, p1, p2 = rp.problems.generate(777, 0, n, rp.problems.DIST_MONOTONE)
    with rp.Batch(n) as b:
        b.set_problems(p0, p1, p2)
        b.solve(1e-8, 200, 0)
        it, status = b.get_iters()
        r = b.reduce()
        st = b.get_state()
    assert np.all(status == rp.ST_CONVERGED) and r["n_converged"] == n and r["total_steps"] == float(it.sum())
    for sl in (slice(0, 8192), slice(n - 8192, n)):
        init = oracle.batch_init_feasible(3, p0[sl], p1[sl], p2[sl])
        aos = init.copy()
        it_o, _ = oracle.batch_solve_gated(3, aos, 1e-8, 200)
        ok = keep_mask(len(it_o), certify_iteration_counts(oracle, 3, init, it[sl], it_o, 1e-8))
        assert serr(st[sl, :3][ok], aos[ok, :3]) < 1e-10


def test_config4_eight_million_problems_as_eight_shards_on_one_device(oracle):
    # BASELINE configs[3]: 8,388,608 F3 problems in contiguous shards of 1,048,576 -- one per GPU of the node there;
    # here the eight shards run one after the other on the one device.  Each shard generates its own slice of the job's problems,
    # solves it (gated, fused) and reduces it; the eight summaries reduce (MAX, MAX, SUM, SUM -- what the one RCCL all-reduce of
    # the path does, sharding.py) to the summary of the WHOLE job solved as one batch; two shards meet the oracle in full, the
    # others on their device-side solution records and a range read-back.
    from hip_util import DeviceBuffer
    shards, per = 8, 1 << 20
    n = shards * per
    g = [0.0, -np.inf, 0.0, 0.0]
    recs = []
    with rp.Batch(per) as b, DeviceBuffer(32 * per) as out:
        b.bind_solution(out.ptr)
        for s in range(shards):
            first, count = rp.problems.shard_range(n, s, shards)
            assert (first, count) == (s * per, per)
            q0, q1, q2 = rp.problems.generate(12345, first, count, rp.problems.DIST_MONOTONE)
            b.set_problems(q0, q1, q2)
            b.solve(1e-8, 200, 0)
            rs = b.reduce()
            rec = out.read(np.dtype(rp.capi.SOLUTION_FIELDS))
            assert np.all(rec["status"] == rp.ST_CONVERGED) and rs["n_converged"] == per and rs["total_steps"] == float(rec["iters"].sum())
            assert np.array_equal(b.get_state_range(777, 64)[:, :3], np.stack([rec["vel1"], rec["duration0"], rec["duration1"]], axis=1)[777:841])
            if s in (0, 5):
                st = b.get_state()
                assert np.array_equal(st[:, 0], rec["vel1"]) and np.array_equal(st[:, 2], rec["duration1"])
                _whole_batch_against_the_oracle(oracle, q0, q1, q2, rec["iters"], st, "configs[3] shard %d of 8" % s)
            g = [max(g[0], rs["max_residual_sq"]), max(g[1], rs["max_gap"]), g[2] + rs["n_converged"], g[3] + rs["total_steps"]]
            recs.append((rec["iters"].astype(np.int64).sum(), float(rec["duration0"].sum()), float(rec["vel1"][12345])))
    # the whole job as ONE batch of 8,388,608 problems: same summary, same per-shard sums
    p0, p1, p2 = rp.problems.generate(12345, 0, n, rp.problems.DIST_MONOTONE)
    with rp.Batch(n) as whole, DeviceBuffer(32 * n) as out:
        whole.bind_solution(out.ptr)
        whole.set_problems(p0, p1, p2)
        whole.solve(1e-8, 200, 0)
        r = whole.reduce()
        rec = out.read(np.dtype(rp.capi.SOLUTION_FIELDS))
    assert g == [r["max_residual_sq"], r["max_gap"], r["n_converged"], r["total_steps"]] and r["n_converged"] == n
    for s in range(shards):
        sl = slice(s * per, (s + 1) * per)
        assert recs[s] == (rec["iters"][sl].astype(np.int64).sum(), float(rec["duration0"][sl].sum()), float(rec["vel1"][s * per + 12345]))
    print("configs[3]: 8 shards of 1,048,576 on one device: %.0f steps, all %d converged, max gap %.3e; summaries reduce to the whole batch's" % (g[3], n, g[1]))
